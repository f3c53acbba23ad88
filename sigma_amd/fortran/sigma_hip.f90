!==========================================================================!
! module sigma_hip                                                         !
!                                                                          !
! Thin ISO_C_BINDING host layer over libsigma_hip.so (include/sigma_hip.h).!
! It keeps SiGMA's type-bound interface for the hot path, so host code     !
! written against the reference reads the same:                            !
!                                                                          !
!   reference (src/...)                       this module                  !
!   type(csr_matrix)  cs_matrices.f90:112     type(hip_csr_matrix)         !
!   type(ellpack_matrix) ellpack_matrices:28  type(hip_ellpack_matrix)     !
!   A%matvec / A%matvec_add                   same names                   !
!     linear_operator_interface.f90:185-194                                !
!   cg(tol), bicgstab(tol)  cg_solvers.f90:36 hip_cg(tol), hip_bicgstab()  !
!   jacobi(), ldu()     jacobi_solvers.f90:23 hip_jacobi(), hip_ldu()      !
!   solver%setup(A), solver%solve(A,x,b[,pc]), solver%destroy()            !
!   solver%iterations, %tolerance, %nn, %initialized                       !
!                                                                          !
! The matrix types hold the SAME host arrays the reference types hold      !
! (g%ptr, g%node, val -- 1-based, cs_graphs.f90:16, cs_matrices.f90:35)    !
! plus one opaque device handle; `upload` (or the first matvec/solve)      !
! pushes them to HBM.  Nonzero status from the C side is turned into the   !
! reference's error behaviour: print + call exit(1) (cg_solvers.f90:61-65).!
!                                                                          !
!   type(sparse_matrix) composite             type(hip_sparse_matrix)      !
!     sparse_matrix_composites.f90:41-162       set_submatrix, matvec(_t)  !
!   lanczos(A,T,Q), generalized_lanczos       hip_lanczos,                 !
!     eigensolver.f90:27-38,95-108              hip_generalized_lanczos    !
!   add_edge / convert / set_value assembly   hip_csr_from_edges           !
!   (nothing: "This loop can be parallelized" hip_comm, hip_dist_csr_matrix!
!     sparse_matrix_composites.f90:1086)        one process per GPU, RCCL  !
!                                                                          !
! Every matrix type extends the abstract hip_matrix (a device handle that  !
! `upload` brings up to date); the solvers take class(hip_matrix).         !
!                                                                          !
! This file does NOT depend on the reference's modules, so that it builds  !
! on the GPU box; oracle/hip_binding.f90 (quoted in INTEGRATION.md) is the !
! variant that extends the reference's own csr_matrix / linear_solver      !
! types instead.  Every entry point of include/sigma_hip.h has its         !
! interface block here (tests/test_cabi_cpu.py checks the list).           !
!==========================================================================!
module sigma_hip

use iso_c_binding

implicit none

integer, parameter :: dp = kind(0.d0)       ! src/types.f90:5

integer(c_int), parameter :: SGM_HOST = 0, SGM_DEVICE = 1


!--------------------------------------------------------------------------!
! C ABI (include/sigma_hip.h)                                              !
!--------------------------------------------------------------------------!
interface
    function sgm_init(device) bind(c, name='sgm_init') result(rc)
        import :: c_int
        integer(c_int), value :: device
        integer(c_int) :: rc
    end function
    function sgm_set_option(name, value) bind(c, name='sgm_set_option') result(rc)
        import :: c_int, c_char
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), value :: value
        integer(c_int) :: rc
    end function
    function sgm_last_error() bind(c, name='sgm_last_error') result(msg)
        import :: c_ptr
        type(c_ptr) :: msg
    end function
    function sgm_csr_create(A, nrow, ncol, nnz, ptr, node, val, where) &
            & bind(c, name='sgm_csr_create') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nrow, ncol
        integer(c_int64_t), value :: nnz
        integer(c_int32_t), intent(in) :: ptr(*), node(*)
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_csr_set_values(A, val, where) &
            & bind(c, name='sgm_csr_set_values') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_ell_create(A, nrow, ncol, max_d, node, val, where) &
            & bind(c, name='sgm_ell_create') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_double
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nrow, ncol, max_d
        integer(c_int32_t), intent(in) :: node(*)
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_ell_set_values(A, val, where) &
            & bind(c, name='sgm_ell_set_values') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec(A, x, y, where) &
            & bind(c, name='sgm_mat_matvec') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec_add(A, x, y, where) &
            & bind(c, name='sgm_mat_matvec_add') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec_t(A, x, y, where) &
            & bind(c, name='sgm_mat_matvec_t') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec_t_add(A, x, y, where) &
            & bind(c, name='sgm_mat_matvec_t_add') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_left_permute(A, p, where) &
            & bind(c, name='sgm_mat_left_permute') result(rc)
        import
        type(c_ptr), value :: A
        integer(c_int32_t), intent(in) :: p(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_right_permute(A, p, where) &
            & bind(c, name='sgm_mat_right_permute') result(rc)
        import
        type(c_ptr), value :: A
        integer(c_int32_t), intent(in) :: p(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_graph_bfs_order(A, p) bind(c, name='sgm_graph_bfs_order') result(rc)
        import
        type(c_ptr), value :: A
        integer(c_int32_t), intent(out) :: p(*)
        integer(c_int) :: rc
    end function
    function sgm_graph_greedy_coloring(A, colors, num_colors) &
            & bind(c, name='sgm_graph_greedy_coloring') result(rc)
        import
        type(c_ptr), value :: A
        integer(c_int32_t), intent(out) :: colors(*)
        integer(c_int32_t), intent(out) :: num_colors
        integer(c_int) :: rc
    end function
    function sgm_graph_greedy_color_order(A, p, ptrs, ptrs_len, num_colors) &
            & bind(c, name='sgm_graph_greedy_color_order') result(rc)
        import
        type(c_ptr), value :: A
        integer(c_int32_t), intent(out) :: p(*), ptrs(*)
        integer(c_int32_t), value :: ptrs_len
        integer(c_int32_t), intent(out) :: num_colors
        integer(c_int) :: rc
    end function
    function sgm_mat_get(A, name, out, bytes, needed) bind(c, name='sgm_mat_get') result(rc)
        import
        type(c_ptr), value :: A
        character(kind=c_char), intent(in) :: name(*)
        type(c_ptr), value :: out
        integer(c_size_t), value :: bytes
        type(c_ptr), value :: needed
        integer(c_int) :: rc
    end function
    function sgm_mat_destroy(A) bind(c, name='sgm_mat_destroy') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: A
        integer(c_int) :: rc
    end function
    function sgm_jacobi_create(pc, A) bind(c, name='sgm_jacobi_create') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), intent(out) :: pc
        type(c_ptr), value :: A
        integer(c_int) :: rc
    end function
    function sgm_ildu0_create(pc, A) bind(c, name='sgm_ildu0_create') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), intent(out) :: pc
        type(c_ptr), value :: A
        integer(c_int) :: rc
    end function
    function sgm_pc_setup(pc, A) bind(c, name='sgm_pc_setup') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: pc, A
        integer(c_int) :: rc
    end function
    function sgm_pc_apply(pc, r, z, where) bind(c, name='sgm_pc_apply') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: pc
        real(c_double), intent(in) :: r(*)
        real(c_double), intent(inout) :: z(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_pc_destroy(pc) bind(c, name='sgm_pc_destroy') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: pc
        integer(c_int) :: rc
    end function
    function sgm_cg_create(s, tolerance) bind(c, name='sgm_cg_create') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), intent(out) :: s
        real(c_double), value :: tolerance
        integer(c_int) :: rc
    end function
    function sgm_bicgstab_create(s, tolerance) &
            & bind(c, name='sgm_bicgstab_create') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), intent(out) :: s
        real(c_double), value :: tolerance
        integer(c_int) :: rc
    end function
    function sgm_gmres_create(s, tolerance, restart) &
            & bind(c, name='sgm_gmres_create') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_double
        type(c_ptr), intent(out) :: s
        real(c_double), value :: tolerance
        integer(c_int32_t), value :: restart
        integer(c_int) :: rc
    end function
    function sgm_solver_setup(s, A) bind(c, name='sgm_solver_setup') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: s, A
        integer(c_int) :: rc
    end function
    function sgm_solver_solve(s, A, x, b, pc, where) &
            & bind(c, name='sgm_solver_solve') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: s, A, pc
        real(c_double), intent(inout) :: x(*)
        real(c_double), intent(in) :: b(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_solver_info(s, iterations, res2, converged, last) &
            & bind(c, name='sgm_solver_info') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), value :: s
        integer(c_int64_t), intent(out) :: iterations, last
        real(c_double), intent(out) :: res2
        integer(c_int32_t), intent(out) :: converged
        integer(c_int) :: rc
    end function
    function sgm_solver_destroy(s) bind(c, name='sgm_solver_destroy') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: s
        integer(c_int) :: rc
    end function
    ! ---- the rest of include/sigma_hip.h ------------------------------------
    function sgm_finalize() bind(c, name='sgm_finalize') result(rc)
        import :: c_int
        integer(c_int) :: rc
    end function
    function sgm_set_stream(stream) bind(c, name='sgm_set_stream') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: stream
        integer(c_int) :: rc
    end function
    function sgm_set_async(on) bind(c, name='sgm_set_async') result(rc)
        import :: c_int
        integer(c_int), value :: on
        integer(c_int) :: rc
    end function
    function sgm_synchronize() bind(c, name='sgm_synchronize') result(rc)
        import :: c_int
        integer(c_int) :: rc
    end function
    function sgm_heartbeat(out6, phase_name, len) bind(c, name='sgm_heartbeat') result(rc)
        import :: c_int, c_int64_t, c_char
        integer(c_int64_t), intent(out) :: out6(6)
        character(kind=c_char), intent(out) :: phase_name(*)
        integer(c_int), value :: len
        integer(c_int) :: rc
    end function
    function sgm_malloc(p, bytes) bind(c, name='sgm_malloc') result(rc)
        import :: c_ptr, c_int, c_size_t
        type(c_ptr), intent(out) :: p
        integer(c_size_t), value :: bytes
        integer(c_int) :: rc
    end function
    function sgm_free(p) bind(c, name='sgm_free') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: p
        integer(c_int) :: rc
    end function
    function sgm_memcpy(dst, src, bytes, kind) bind(c, name='sgm_memcpy') result(rc)
        import :: c_ptr, c_int, c_size_t
        type(c_ptr), value :: dst, src
        integer(c_size_t), value :: bytes
        integer(c_int), value :: kind          ! 0 host -> device, 1 device -> host, 2 device -> device
        integer(c_int) :: rc
    end function
    function sgm_dot(n, a, b, res, where) bind(c, name='sgm_dot') result(rc)
        import :: c_int, c_int64_t, c_double
        integer(c_int64_t), value :: n
        real(c_double), intent(in) :: a(*), b(*)
        real(c_double), intent(out) :: res
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_axpy(n, alpha, x, y, where) bind(c, name='sgm_axpy') result(rc)
        import :: c_int, c_int64_t, c_double
        integer(c_int64_t), value :: n
        real(c_double), value :: alpha
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_csr_from_edges(A, nrow, ncol, ne, ei, ej, ev, where) bind(c, name='sgm_csr_from_edges') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nrow, ncol
        integer(c_int64_t), value :: ne
        integer(c_int32_t), intent(in) :: ei(*), ej(*)
        real(c_double), intent(in) :: ev(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_ell_from_edges(A, nrow, ncol, ne, ei, ej, ev, where) bind(c, name='sgm_ell_from_edges') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nrow, ncol
        integer(c_int64_t), value :: ne
        integer(c_int32_t), intent(in) :: ei(*), ej(*)
        real(c_double), intent(in) :: ev(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_composite_create(A, nrb, ncb, row_ptr, col_ptr, blocks) bind(c, name='sgm_composite_create') result(rc)
        import :: c_ptr, c_int, c_int32_t
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nrb, ncb
        integer(c_int32_t), intent(in) :: row_ptr(*), col_ptr(*)
        type(c_ptr), intent(in) :: blocks(*)
        integer(c_int) :: rc
    end function
    function sgm_mat_info(A, nrow, ncol, nnz, fmt, x_len) bind(c, name='sgm_mat_info') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t
        type(c_ptr), value :: A
        integer(c_int32_t), intent(out) :: nrow, ncol, fmt
        integer(c_int64_t), intent(out) :: nnz, x_len
        integer(c_int) :: rc
    end function
    function sgm_mat_kernel(A, buf, len) bind(c, name='sgm_mat_kernel') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: A
        character(kind=c_char), intent(out) :: buf(*)
        integer(c_int), value :: len
        integer(c_int) :: rc
    end function
    function sgm_mat_footprint(A, resident_bytes, matvec_bytes) bind(c, name='sgm_mat_footprint') result(rc)
        import :: c_ptr, c_int, c_int64_t
        type(c_ptr), value :: A
        integer(c_int64_t), intent(out) :: resident_bytes, matvec_bytes
        integer(c_int) :: rc
    end function
    function sgm_pc_info(pc, part, out4, est_us, path_name, len) bind(c, name='sgm_pc_info') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_double, c_char
        type(c_ptr), value :: pc
        integer(c_int32_t), value :: part
        integer(c_int32_t), intent(out) :: out4(4)
        real(c_double), intent(out) :: est_us
        character(kind=c_char), intent(out) :: path_name(*)
        integer(c_int), value :: len
        integer(c_int) :: rc
    end function
    function sgm_pc_get(pc, name, out, bytes, needed) bind(c, name='sgm_pc_get') result(rc)
        import
        type(c_ptr), value :: pc
        character(kind=c_char), intent(in) :: name(*)
        type(c_ptr), value :: out
        integer(c_size_t), value :: bytes
        type(c_ptr), value :: needed
        integer(c_int) :: rc
    end function
    function sgm_solver_set_tolerance(s, tolerance) bind(c, name='sgm_solver_set_tolerance') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: s
        real(c_double), value :: tolerance
        integer(c_int) :: rc
    end function
    function sgm_solver_set_max_iter(s, max_iter) bind(c, name='sgm_solver_set_max_iter') result(rc)
        import :: c_ptr, c_int, c_int64_t
        type(c_ptr), value :: s
        integer(c_int64_t), value :: max_iter
        integer(c_int) :: rc
    end function
    function sgm_solver_set_history(s, capacity) bind(c, name='sgm_solver_set_history') result(rc)
        import :: c_ptr, c_int, c_int64_t
        type(c_ptr), value :: s
        integer(c_int64_t), value :: capacity
        integer(c_int) :: rc
    end function
    function sgm_solver_get_history(s, out, capacity, count) bind(c, name='sgm_solver_get_history') result(rc)
        import :: c_ptr, c_int, c_int64_t, c_double
        type(c_ptr), value :: s
        real(c_double), intent(out) :: out(*)
        integer(c_int64_t), value :: capacity
        integer(c_int64_t), intent(out) :: count
        integer(c_int) :: rc
    end function
    function sgm_lanczos(A, nsteps, q1, T, Q, where) bind(c, name='sgm_lanczos') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_double
        type(c_ptr), value :: A
        integer(c_int32_t), value :: nsteps
        real(c_double), intent(in) :: q1(*)
        real(c_double), intent(out) :: T(3, *), Q(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_generalized_lanczos(A, B, solver, pc, nsteps, q1, T, Q, where) &
            & bind(c, name='sgm_generalized_lanczos') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_double
        type(c_ptr), value :: A, B, solver, pc
        integer(c_int32_t), value :: nsteps
        real(c_double), intent(in) :: q1(*)
        real(c_double), intent(out) :: T(3, *), Q(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    ! ---- row-partitioned multi-GPU -----------------------------------------------
    function sgm_comm_unique_id(id) bind(c, name='sgm_comm_unique_id') result(rc)
        import :: c_int, c_char
        character(kind=c_char), intent(out) :: id(128)
        integer(c_int) :: rc
    end function
    function sgm_comm_init(comm, rank, nranks, id) bind(c, name='sgm_comm_init') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), intent(out) :: comm
        integer(c_int), value :: rank, nranks
        character(kind=c_char), intent(in) :: id(128)
        integer(c_int) :: rc
    end function
    function sgm_comm_attach_halo_comm(comm, id) bind(c, name='sgm_comm_attach_halo_comm') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: comm
        character(kind=c_char), intent(in) :: id(128)
        integer(c_int) :: rc
    end function
    function sgm_comm_destroy(comm) bind(c, name='sgm_comm_destroy') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: comm
        integer(c_int) :: rc
    end function
    function sgm_dist_profile(on) bind(c, name='sgm_dist_profile') result(rc)
        import :: c_int
        integer(c_int), value :: on
        integer(c_int) :: rc
    end function
    function sgm_dist_profile_read(ms, count) bind(c, name='sgm_dist_profile_read') result(rc)
        import :: c_int, c_int64_t, c_double
        real(c_double), intent(out) :: ms(6)
        integer(c_int64_t), intent(out) :: count(6)
        integer(c_int) :: rc
    end function
    function sgm_partition_rows_by_nnz(nrow, ptr, nparts, align, row_starts) &
            & bind(c, name='sgm_partition_rows_by_nnz') result(rc)
        import :: c_int, c_int32_t, c_int64_t
        integer(c_int32_t), value :: nrow, nparts, align
        integer(c_int32_t), intent(in) :: ptr(*)
        integer(c_int64_t), intent(out) :: row_starts(*)
        integer(c_int) :: rc
    end function
    function sgm_csr_create_dist(A, comm, row_starts, nnz, ptr, node, val, where) &
            & bind(c, name='sgm_csr_create_dist') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        type(c_ptr), value :: comm
        integer(c_int64_t), intent(in) :: row_starts(*)
        integer(c_int64_t), value :: nnz
        integer(c_int32_t), intent(in) :: ptr(*), node(*)
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_csr_create_dist_rect(A, comm, row_starts, col_starts, nnz, ptr, node, val, where) &
            & bind(c, name='sgm_csr_create_dist_rect') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        type(c_ptr), value :: comm
        integer(c_int64_t), intent(in) :: row_starts(*), col_starts(*)
        integer(c_int64_t), value :: nnz
        integer(c_int32_t), intent(in) :: ptr(*), node(*)
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_ell_create_dist(A, comm, row_starts, max_d, node, val, where) &
            & bind(c, name='sgm_ell_create_dist') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        type(c_ptr), value :: comm
        integer(c_int64_t), intent(in) :: row_starts(*)
        integer(c_int32_t), value :: max_d
        integer(c_int32_t), intent(in) :: node(*)
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_csr_create_partitioned(A, nparts, row_starts, nrow, ncol, nnz, ptr, node, val) &
            & bind(c, name='sgm_csr_create_partitioned') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nparts, nrow, ncol
        integer(c_int64_t), intent(in) :: row_starts(*)
        integer(c_int64_t), value :: nnz
        integer(c_int32_t), intent(in) :: ptr(*), node(*)
        real(c_double), intent(in) :: val(*)
        integer(c_int) :: rc
    end function
    function sgm_csr_create_partitioned_parts(A, nparts, row_starts, nnz_of_part, ptr_of_part, node_of_part, val_of_part, where) &
            & bind(c, name='sgm_csr_create_partitioned_parts') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nparts
        integer(c_int64_t), intent(in) :: row_starts(*), nnz_of_part(*)
        type(c_ptr), intent(in) :: ptr_of_part(*), node_of_part(*), val_of_part(*)      ! c_loc of every part's arrays
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_set_option(A, name, value) bind(c, name='sgm_mat_set_option') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: A
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), value :: value
        integer(c_int) :: rc
    end function
    function sgm_solver_set_option(s, name, value) bind(c, name='sgm_solver_set_option') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: s
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), value :: value
        integer(c_int) :: rc
    end function
    function sgm_pc_set_option(pc, name, value) bind(c, name='sgm_pc_set_option') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: pc
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), value :: value
        integer(c_int) :: rc
    end function
    function sgm_pc_create(pc, kind) bind(c, name='sgm_pc_create') result(rc)
        import :: c_ptr, c_int, c_int32_t
        type(c_ptr), intent(out) :: pc
        integer(c_int32_t), value :: kind          ! 1 jacobi, 2 ildu(0)
        integer(c_int) :: rc
    end function
    function c_usleep(us) bind(c, name='usleep') result(rc)
        import :: c_int
        integer(c_int), value :: us
        integer(c_int) :: rc
    end function
end interface


!--------------------------------------------------------------------------!
type, abstract :: hip_matrix                                               !
!--------------------------------------------------------------------------!
! linear_operator (linear_operator_interface.f90:18-45) as this layer sees !
! it: dimensions + one device handle that `upload` brings up to date.      !
! matvec / matvec_add / matvec_t / matvec_t_add are the same for every     !
! matrix type: upload, then the C entry point.                             !
!--------------------------------------------------------------------------!
    integer :: nrow = 0, ncol = 0
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure(hip_matrix_upload_ifc), deferred :: upload
    procedure :: matvec => hip_matrix_matvec
    procedure :: matvec_add => hip_matrix_matvec_add
    procedure :: matvec_t => hip_matrix_matvec_t
    procedure :: matvec_t_add => hip_matrix_matvec_t_add
    procedure :: kernel_name => hip_matrix_kernel_name
    procedure :: set_option => hip_matrix_set_option
end type hip_matrix

abstract interface
    subroutine hip_matrix_upload_ifc(A)
        import :: hip_matrix
        class(hip_matrix), intent(inout) :: A
    end subroutine
end interface


!--------------------------------------------------------------------------!
type, extends(hip_matrix) :: hip_csr_matrix                                !
!--------------------------------------------------------------------------!
! Same host data as csr_matrix (cs_matrices.f90:32-38,112): ptr/node of    !
! the cs_graph, val of the matrix.                                         !
!--------------------------------------------------------------------------!
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:)
    logical :: values_dirty = .true.
contains
    procedure :: init => hip_csr_init
    procedure :: get_value => hip_csr_get_value
    procedure :: set_value => hip_csr_set_value
    procedure :: add_value => hip_csr_add_value
    procedure :: zero => hip_csr_zero
    procedure :: upload => hip_csr_upload
    procedure :: left_permute => hip_csr_left_permute
    procedure :: right_permute => hip_csr_right_permute
    procedure :: destroy => hip_csr_destroy
end type hip_csr_matrix


!--------------------------------------------------------------------------!
type, extends(hip_matrix) :: hip_ellpack_matrix                            !
!--------------------------------------------------------------------------!
! Same host data as ellpack_matrix (ellpack_matrices.f90:28-33):           !
! node(max_d,n), degrees(n), val(max_d,n); padding slots repeat the last   !
! neighbour (ellpack_graphs.f90:164).                                      !
!--------------------------------------------------------------------------!
    integer :: max_d = 0
    integer, allocatable :: node(:,:), degrees(:)
    real(dp), allocatable :: val(:,:)
    logical :: values_dirty = .true.
contains
    procedure :: init => hip_ell_init
    procedure :: add_edge => hip_ell_add_edge
    procedure :: set_value => hip_ell_set_value
    procedure :: zero => hip_ell_zero
    procedure :: upload => hip_ell_upload
    procedure :: destroy => hip_ell_destroy
end type hip_ellpack_matrix


!--------------------------------------------------------------------------!
type :: hip_matrix_pointer                                                 !
!--------------------------------------------------------------------------!
    class(hip_matrix), pointer :: mat => null()
end type hip_matrix_pointer


!--------------------------------------------------------------------------!
type, extends(hip_matrix) :: hip_sparse_matrix                             !
!--------------------------------------------------------------------------!
! type(sparse_matrix), the block "matrix of matrices"                      !
! (sparse_matrix_composites.f90:41-162): set_num_blocks, set_block_sizes,  !
! set_submatrix(it, jt, B) with hip_csr_matrix / hip_ellpack_matrix leaves !
! (which stay the caller's); matvec_add is the reference's block loop      !
! (:1076-1099) run by the device over ONE handle (sgm_composite_create);   !
! the unpreconditioned and Jacobi-preconditioned solvers take it like any  !
! other matrix.                                                            !
!--------------------------------------------------------------------------!
    integer :: num_row_mats = 0, num_col_mats = 0
    integer, allocatable :: row_ptr(:), col_ptr(:)
    type(hip_matrix_pointer), allocatable :: sub_mats(:,:)
    type(c_ptr), allocatable :: built_from(:)
contains
    procedure :: set_num_blocks => hip_comp_set_num_blocks
    procedure :: set_block_sizes => hip_comp_set_block_sizes
    procedure :: set_submatrix => hip_comp_set_submatrix
    procedure :: upload => hip_comp_upload
    procedure :: destroy => hip_comp_destroy
end type hip_sparse_matrix


!--------------------------------------------------------------------------!
type :: hip_comm                                                           !
!--------------------------------------------------------------------------!
! One process per GPU, rank r of nranks; RCCL's 128-byte unique id goes    !
! from rank 0 to the others through a file (no MPI needed).                !
!--------------------------------------------------------------------------!
    integer :: rank = 0, nranks = 1
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure :: init => hip_comm_init
    procedure :: destroy => hip_comm_destroy
end type hip_comm


!--------------------------------------------------------------------------!
type, extends(hip_matrix) :: hip_dist_csr_matrix                           !
!--------------------------------------------------------------------------!
! This rank's contiguous row block of a square CSR matrix partitioned over !
! the ranks of a hip_comm.  nrow = ncol = the owned rows: products and     !
! solves take the owned slices of the vectors; halo entries of x travel    !
! between neighbour ranks inside the library, dot products inside the      !
! solvers are all-reduced.                                                 !
!--------------------------------------------------------------------------!
    integer :: nrow_global = 0, row_first = 0, row_last = 0
    integer(c_int64_t) :: x_len = 0
    integer(c_int64_t), allocatable :: row_starts(:)
contains
    procedure :: distribute => hip_dist_distribute
    procedure :: upload => hip_dist_upload
    procedure :: matvec => hip_dist_matvec
    procedure :: matvec_add => hip_dist_matvec_add
    procedure :: destroy => hip_dist_destroy
end type hip_dist_csr_matrix


!--------------------------------------------------------------------------!
type :: hip_device_vector                                                  !
!--------------------------------------------------------------------------!
! A vector that lives in HBM (sgm_malloc / sgm_memcpy / sgm_free): solves  !
! and products on device vectors cross PCIe only when the host asks        !
! (upload / download).  `view` is a Fortran array pointer onto the device  !
! memory -- an ADDRESS to hand to the library with SGM_DEVICE, never to be !
! dereferenced on the host.                                                !
!--------------------------------------------------------------------------!
    integer :: n = 0
    type(c_ptr) :: p = c_null_ptr
    real(dp), pointer :: view(:) => null()
contains
    procedure :: alloc => hip_dvec_alloc
    procedure :: upload => hip_dvec_upload
    procedure :: download => hip_dvec_download
    procedure :: free => hip_dvec_free
end type hip_device_vector


!--------------------------------------------------------------------------!
type :: hip_linear_solver                                                  !
!--------------------------------------------------------------------------!
! linear_solver (linear_operator_interface.f90:61-73) + the public fields  !
! of cg_solver / bicgstab_solver (cg_solvers.f90:13-18).                   !
!--------------------------------------------------------------------------!
    integer :: nn = 0
    logical :: initialized = .false.
    integer :: iterations = 0
    real(dp) :: tolerance = 1.0d-16
    integer :: kind = 0          ! 1 cg, 2 bicgstab, 3 gmres, 11 jacobi, 12 ldu
    integer :: restart = 30
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure :: setup => hip_solver_setup
    procedure :: solve_plain => hip_solver_solve
    procedure :: solve_pc => hip_solver_solve_pc
    generic :: solve => solve_plain, solve_pc
    procedure :: solve_device => hip_solver_solve_device
    procedure :: set_option => hip_solver_set_option
    procedure :: info => hip_solver_pc_info
    procedure :: set_max_iter => hip_solver_set_max_iter
    procedure :: set_params => hip_solver_set_params
    procedure :: destroy => hip_solver_destroy
end type hip_linear_solver


contains


!==========================================================================!
!==== error handling: print + exit(1), like the reference               ====!
!==========================================================================!
subroutine hip_check(rc)
    integer(c_int), intent(in) :: rc
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: cmsg
    integer :: k

    if (rc == 0) return
    cmsg = sgm_last_error()
    call c_f_pointer(cmsg, msg, [1024])
    k = 1
    do while (k < 1024 .and. msg(k) /= c_null_char)
        k = k + 1
    enddo
    print *, msg(1:k-1)
    print *, 'Terminating.'
    call exit(1)
end subroutine hip_check


!==========================================================================!
!==== options of the library (include/sigma_hip.h, sgm_set_option)      ====!
!==========================================================================!
subroutine hip_set_option(name, value)
    ! e.g. call hip_set_option("cg_small", 0): results do not depend on any of them beyond the
    ! summation order of dot products
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    call hip_check(sgm_set_option(trim(name) // c_null_char, int(value, c_int)))
end subroutine hip_set_option


subroutine hip_matrix_set_option(A, name, value)
    ! this matrix's own kernel-selection option (sgm_mat_set_option); hip_set_option only changes what matrices
    ! created later start with
    class(hip_matrix), intent(inout) :: A
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    call A%upload()
    call hip_check(sgm_mat_set_option(A%handle, trim(name) // c_null_char, int(value, c_int)))
end subroutine hip_matrix_set_option


!==========================================================================!
!==== CSR                                                               ====!
!==========================================================================!
subroutine hip_csr_init(A, nrow, ncol, ptr, node)
    ! A%init(nrow,ncol) + A%set_graph(g): the graph arrives as its arrays
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: nrow, ncol, ptr(:), node(:)
    A%nrow = nrow
    A%ncol = ncol
    A%ptr = ptr
    A%node = node
    allocate(A%val(size(node)))
    A%val = 0.0_dp
    A%values_dirty = .true.
end subroutine

function hip_csr_get_value(A, i, j) result(z)      ! cs_matrices.f90:709-724
    class(hip_csr_matrix), intent(in) :: A
    integer, intent(in) :: i, j
    real(dp) :: z
    integer :: k
    z = 0.0_dp
    do k = A%ptr(i), A%ptr(i + 1) - 1
        if (A%node(k) == j) z = A%val(k)
    enddo
end function

subroutine hip_csr_set_value(A, i, j, z)           ! cs_matrices.f90:840-863
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    integer :: k
    logical :: found
    found = .false.
    do k = A%ptr(i), A%ptr(i + 1) - 1
        if (A%node(k) == j) then
            A%val(k) = z
            found = .true.
        endif
    enddo
    if (.not. found) then
        print *, 'hip_csr_matrix: entry', i, j, 'is not in the sparsity pattern'
        call exit(1)
    endif
    A%values_dirty = .true.
end subroutine

subroutine hip_csr_add_value(A, i, j, z)           ! cs_matrices.f90:868-895
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    integer :: k
    do k = A%ptr(i), A%ptr(i + 1) - 1
        if (A%node(k) == j) A%val(k) = A%val(k) + z
    enddo
    A%values_dirty = .true.
end subroutine

subroutine hip_csr_zero(A)
    class(hip_csr_matrix), intent(inout) :: A
    A%val = 0.0_dp
    A%values_dirty = .true.
end subroutine

subroutine hip_csr_upload(A)
    class(hip_csr_matrix), intent(inout) :: A
    if (.not. c_associated(A%handle)) then
        call hip_check(sgm_csr_create(A%handle, int(A%nrow, c_int32_t), &
            & int(A%ncol, c_int32_t), int(size(A%node), c_int64_t), &
            & A%ptr, A%node, A%val, SGM_HOST))
    elseif (A%values_dirty) then
        call hip_check(sgm_csr_set_values(A%handle, A%val, SGM_HOST))
    endif
    A%values_dirty = .false.
end subroutine

!==========================================================================!
!==== products: the same for every matrix type                          ====!
!==========================================================================!
subroutine hip_matrix_matvec(A, x, y)      ! linear_operator_interface.f90:185-194
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_matrix_matvec_add(A, x, y)  ! cs_matrices.f90:600-622, ellpack_matrices.f90:640-665
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_add(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_matrix_matvec_t(A, x, y)    ! linear_operator_interface.f90:199-208
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_t(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_matrix_matvec_t_add(A, x, y)  ! csc_matvec_add, cs_matrices.f90:627-647
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_t_add(A%handle, x, y, SGM_HOST))
end subroutine

function hip_matrix_kernel_name(A) result(name)
    ! the SpMV kernel variant the matrix runs with (diagnostics: sgm_mat_kernel)
    class(hip_matrix), intent(inout) :: A
    character(len=:), allocatable :: name
    character(kind=c_char) :: buf(64)
    integer :: k
    call A%upload()
    call hip_check(sgm_mat_kernel(A%handle, buf, 64_c_int))
    name = ''
    do k = 1, 64
        if (buf(k) == c_null_char) exit
        name = name // buf(k)
    enddo
end function

! cs_matrices.f90:471-490: the permutation runs on the device; the host copies of the arrays
! (kept for get_value / set_value) are refreshed from it
subroutine hip_csr_left_permute(A, p)
    class(hip_csr_matrix), intent(inout), target :: A
    integer, intent(in) :: p(:)
    call A%upload()
    call hip_check(sgm_mat_left_permute(A%handle, int(p, c_int32_t), SGM_HOST))
    call hip_csr_download(A)
end subroutine

subroutine hip_csr_right_permute(A, p)
    class(hip_csr_matrix), intent(inout), target :: A
    integer, intent(in) :: p(:)
    call A%upload()
    call hip_check(sgm_mat_right_permute(A%handle, int(p, c_int32_t), SGM_HOST))
    call hip_csr_download(A)
end subroutine

subroutine hip_csr_download(A)
    class(hip_csr_matrix), intent(inout), target :: A
    call hip_check(sgm_mat_get(A%handle, 'ptr'//c_null_char, c_loc(A%ptr), &
        & int(4 * size(A%ptr), c_size_t), c_null_ptr))
    call hip_check(sgm_mat_get(A%handle, 'node'//c_null_char, c_loc(A%node), &
        & int(4 * size(A%node), c_size_t), c_null_ptr))
    call hip_check(sgm_mat_get(A%handle, 'val'//c_null_char, c_loc(A%val), &
        & int(8 * size(A%val), c_size_t), c_null_ptr))
    A%values_dirty = .false.
end subroutine

! src/graph/permutations.f90 on the matrix graph (the routines take the matrix, which owns it)
subroutine hip_breadth_first_search(p, A)              ! permutations.f90:22-78
    integer, intent(out) :: p(:)
    class(hip_csr_matrix), intent(inout) :: A
    integer(c_int32_t) :: p32(size(p))
    call A%upload()
    call hip_check(sgm_graph_bfs_order(A%handle, p32))
    p = p32
end subroutine

subroutine hip_greedy_coloring(colors, A)              ! permutations.f90:83-157
    integer, intent(out) :: colors(:)
    class(hip_csr_matrix), intent(inout) :: A
    integer(c_int32_t) :: c32(size(colors)), nc
    call A%upload()
    call hip_check(sgm_graph_greedy_coloring(A%handle, c32, nc))
    colors = c32
end subroutine

subroutine hip_greedy_color_ordering(p, ptrs, num_colors, A)   ! permutations.f90:162-205
    integer, intent(out) :: p(:), ptrs(:), num_colors
    class(hip_csr_matrix), intent(inout) :: A
    integer(c_int32_t) :: p32(size(p)), t32(size(ptrs)), nc
    call A%upload()
    t32 = 0
    call hip_check(sgm_graph_greedy_color_order(A%handle, p32, t32, int(size(ptrs), c_int32_t), nc))
    p = p32
    ptrs = t32
    num_colors = nc
end subroutine

subroutine hip_csr_destroy(A)
    class(hip_csr_matrix), intent(inout) :: A
    if (c_associated(A%handle)) call hip_check(sgm_mat_destroy(A%handle))
    A%handle = c_null_ptr
    if (allocated(A%ptr)) deallocate(A%ptr, A%node, A%val)
    A%nrow = 0
    A%ncol = 0
end subroutine


!==========================================================================!
!==== ELLPACK                                                           ====!
!==========================================================================!
subroutine hip_ell_init(A, nrow, ncol, max_d)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: nrow, ncol, max_d
    A%nrow = nrow
    A%ncol = ncol
    A%max_d = max_d
    allocate(A%node(max_d, nrow), A%val(max_d, nrow), A%degrees(nrow))
    A%node = 0
    A%degrees = 0
    A%val = 0.0_dp
end subroutine

subroutine hip_ell_add_edge(A, i, j)               ! ellpack_graphs.f90:380-400
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    integer :: k
    do k = 1, A%degrees(i)
        if (A%node(k, i) == j) return
    enddo
    k = A%degrees(i)
    if (k < A%max_d) then
        A%node(k + 1 :, i) = j      ! the rest of the row repeats the newest neighbour
        A%degrees(i) = k + 1
    else
        print *, 'hip_ellpack_matrix: row', i, 'is full'
        call exit(1)
    endif
end subroutine

subroutine hip_ell_set_value(A, i, j, z)           ! ellpack_matrices.f90:444-466
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    integer :: k
    do k = 1, A%degrees(i)
        if (A%node(k, i) == j) A%val(k, i) = z
    enddo
    A%values_dirty = .true.
end subroutine

subroutine hip_ell_zero(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    A%val = 0.0_dp
    A%values_dirty = .true.
end subroutine

subroutine hip_ell_upload(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    if (.not. c_associated(A%handle)) then
        call hip_check(sgm_ell_create(A%handle, int(A%nrow, c_int32_t), &
            & int(A%ncol, c_int32_t), int(A%max_d, c_int32_t), &
            & A%node, A%val, SGM_HOST))
    elseif (A%values_dirty) then
        call hip_check(sgm_ell_set_values(A%handle, A%val, SGM_HOST))
    endif
    A%values_dirty = .false.
end subroutine

subroutine hip_ell_destroy(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    if (c_associated(A%handle)) call hip_check(sgm_mat_destroy(A%handle))
    A%handle = c_null_ptr
    if (allocated(A%node)) deallocate(A%node, A%val, A%degrees)
end subroutine


!==========================================================================!
!==== solver / preconditioner factories and methods                     ====!
!==========================================================================!
function hip_cg(tolerance) result(s)               ! cg_solvers.f90:36-47
    real(dp), intent(in), optional :: tolerance
    type(hip_linear_solver), pointer :: s
    allocate(s)
    s%kind = 1
    if (present(tolerance)) s%tolerance = tolerance
end function

function hip_bicgstab(tolerance) result(s)         ! bicgstab_solvers.f90:37-48
    real(dp), intent(in), optional :: tolerance
    type(hip_linear_solver), pointer :: s
    allocate(s)
    s%kind = 2
    if (present(tolerance)) s%tolerance = tolerance
end function

function hip_gmres(tolerance, restart) result(s)   ! no reference counterpart
    real(dp), intent(in), optional :: tolerance
    integer, intent(in), optional :: restart
    type(hip_linear_solver), pointer :: s
    allocate(s)
    s%kind = 3
    if (present(tolerance)) s%tolerance = tolerance
    if (present(restart)) s%restart = restart
end function

function hip_jacobi() result(s)                    ! jacobi_solvers.f90:23-31
    type(hip_linear_solver), pointer :: s
    allocate(s)
    s%kind = 11
end function

function hip_ldu(incomplete, level, reorder) result(s)      ! ldu_solvers.f90:73-86
    ! reorder = "colour" (extension, off by default): ILDU(0) of the colour-ordered matrix P A P^T (P = greedy_color_ordering of
    ! the graph of A) applied as z = P^T M^-1 P r -- A, b, x stay as they are; iteration counts are the permuted system's
    logical, intent(in), optional :: incomplete
    integer, intent(in), optional :: level
    character(len=*), intent(in), optional :: reorder
    type(hip_linear_solver), pointer :: s
    allocate(s)
    s%kind = 12       ! like ldu_set_params :143-151: always ILDU(0)
    if (present(reorder)) then
        if (reorder == "colour" .or. reorder == "color") call s%set_option("ildu_reorder", 1)
    endif
end function

subroutine hip_solver_make_handle(s)
    ! the factory object becomes a library handle (no matrix needed yet: options can be set on it before setup)
    class(hip_linear_solver), intent(inout) :: s
    if (c_associated(s%handle)) return
    select case (s%kind)
    case (1)
        call hip_check(sgm_cg_create(s%handle, s%tolerance))
    case (2)
        call hip_check(sgm_bicgstab_create(s%handle, s%tolerance))
    case (3)
        call hip_check(sgm_gmres_create(s%handle, s%tolerance, int(s%restart, c_int32_t)))
    case (11)
        call hip_check(sgm_pc_create(s%handle, 1_c_int32_t))
    case (12)
        call hip_check(sgm_pc_create(s%handle, 2_c_int32_t))
    end select
end subroutine

subroutine hip_solver_set_option(s, name, value)
    ! this solver's / preconditioner's own option (sgm_solver_set_option / sgm_pc_set_option)
    class(hip_linear_solver), intent(inout) :: s
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    call hip_solver_make_handle(s)
    if (s%kind > 10) then
        call hip_check(sgm_pc_set_option(s%handle, trim(name) // c_null_char, int(value, c_int)))
    else
        call hip_check(sgm_solver_set_option(s%handle, trim(name) // c_null_char, int(value, c_int)))
    endif
end subroutine

subroutine hip_solver_pc_info(s, levels, path, colours, est_us, name, part)
    ! a preconditioner that has been set up (jacobi / ldu): which sweeps serve its applies and what one costs (sgm_pc_info) --
    ! levels(2) of L and U, path 0 diagonal / 1 row space / 2 strip pipeline / 3 slab pipeline / 4 level walkers, colours of
    ! the ordering (0 = A's own), estimated microseconds per apply, the same in words ("strip pipeline, 6323 levels")
    class(hip_linear_solver), intent(inout) :: s
    integer, intent(out) :: levels(2), path, colours
    real(dp), intent(out) :: est_us
    character(len=*), intent(out) :: name
    integer, intent(in), optional :: part
    integer(c_int32_t) :: o(4)
    real(c_double) :: us
    character(kind=c_char) :: buf(160)
    integer :: k, ip
    ip = 0
    if (present(part)) ip = part
    if (s%kind <= 10) then
        print *, 'info: not a preconditioner'
        print *, 'Terminating.'
        call exit(1)
    endif
    call hip_solver_make_handle(s)
    call hip_check(sgm_pc_info(s%handle, int(ip, c_int32_t), o, us, buf, 160_c_int))
    levels = int(o(1:2)); path = int(o(3)); colours = int(o(4)); est_us = real(us, dp)
    name = ' '
    do k = 1, min(len(name), 160)
        if (buf(k) == c_null_char) exit
        name(k:k) = buf(k)
    enddo
end subroutine

subroutine hip_solver_setup_handle(s, Ah, nrow)
    class(hip_linear_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: Ah
    integer, intent(in) :: nrow
    call hip_solver_make_handle(s)
    if (s%kind > 10) then
        call hip_check(sgm_pc_setup(s%handle, Ah))
    else
        call hip_check(sgm_solver_setup(s%handle, Ah))
    endif
    s%nn = nrow
    s%iterations = 0                    ! cg_solvers.f90:72
    s%initialized = .true.
end subroutine

subroutine hip_solver_setup(s, A)
    class(hip_linear_solver), intent(inout) :: s
    class(hip_matrix), intent(inout) :: A
    call A%upload()
    call hip_solver_setup_handle(s, A%handle, A%nrow)
end subroutine

subroutine hip_solver_solve_handle(s, Ah, x, b, pch)
    class(hip_linear_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: Ah, pch
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    integer(c_int64_t) :: its, last
    real(c_double) :: res2
    integer(c_int32_t) :: conv
    integer(c_int) :: rc
    if (s%kind > 10) then
        ! a preconditioner used as a solver: pc%solve(A, x, b)  (jacobi_solve / ldu_solve)
        call hip_check(sgm_pc_apply(s%handle, b, x, SGM_HOST))
        return
    endif
    ! s%tolerance is live: the reference's loop reads the field at every solve (cg_solvers.f90:133)
    call hip_check(sgm_solver_set_tolerance(s%handle, s%tolerance))
    rc = sgm_solver_solve(s%handle, Ah, x, b, pch, SGM_HOST)
    if (rc /= 5) call hip_check(rc)         ! 5 = stopped at the set_max_iter extension: the caller reads %iterations
    call hip_check(sgm_solver_info(s%handle, its, res2, conv, last))
    s%iterations = int(its)
end subroutine

subroutine hip_solver_set_params(s, tolerance)
    ! cg_set_params (cg_solvers.f90:95-111): callable again at any time; pushed to the handle in front of every solve
    class(hip_linear_solver), intent(inout) :: s
    real(dp), intent(in), optional :: tolerance
    if (present(tolerance)) then
        s%tolerance = tolerance
    else
        s%tolerance = 1.0d-16
    endif
end subroutine

subroutine hip_solver_solve(s, A, x, b)
    class(hip_linear_solver), intent(inout) :: s
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    call A%upload()
    call hip_solver_solve_handle(s, A%handle, x, b, c_null_ptr)
end subroutine

subroutine hip_solver_solve_pc(s, A, x, b, pc)
    class(hip_linear_solver), intent(inout) :: s
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    type(hip_linear_solver), intent(inout) :: pc
    call A%upload()
    call hip_solver_solve_handle(s, A%handle, x, b, pc%handle)
end subroutine

subroutine hip_solver_solve_device(s, A, x, b, pc)
    ! the same solve on vectors that already live in HBM: nothing crosses PCIe
    class(hip_linear_solver), intent(inout) :: s
    class(hip_matrix), intent(inout) :: A
    type(hip_device_vector), intent(inout) :: x
    type(hip_device_vector), intent(in) :: b
    type(hip_linear_solver), intent(inout), optional :: pc
    integer(c_int64_t) :: its, last
    real(c_double) :: res2
    integer(c_int32_t) :: conv
    type(c_ptr) :: hp
    integer(c_int) :: rc
    hp = c_null_ptr
    if (present(pc)) hp = pc%handle
    call A%upload()
    call hip_check(sgm_solver_set_tolerance(s%handle, s%tolerance))
    rc = sgm_solver_solve(s%handle, A%handle, x%view, b%view, hp, SGM_DEVICE)
    if (rc /= 5) call hip_check(rc)
    call hip_check(sgm_solver_info(s%handle, its, res2, conv, last))
    s%iterations = int(its)
end subroutine

subroutine hip_dvec_alloc(v, n)
    class(hip_device_vector), intent(inout) :: v
    integer, intent(in) :: n
    call hip_check(sgm_malloc(v%p, int(8, c_size_t) * max(n, 1)))
    v%n = n
    call c_f_pointer(v%p, v%view, [n])
end subroutine

subroutine hip_dvec_upload(v, x)
    class(hip_device_vector), intent(inout) :: v
    real(dp), intent(in), target :: x(:)
    call hip_check(sgm_memcpy(v%p, c_loc(x), int(8, c_size_t) * v%n, 0_c_int))
end subroutine

subroutine hip_dvec_download(v, x)
    class(hip_device_vector), intent(in) :: v
    real(dp), intent(out), target :: x(:)
    call hip_check(sgm_memcpy(c_loc(x), v%p, int(8, c_size_t) * v%n, 1_c_int))
end subroutine

subroutine hip_dvec_free(v)
    class(hip_device_vector), intent(inout) :: v
    if (c_associated(v%p)) call hip_check(sgm_free(v%p))
    v%p = c_null_ptr
    v%n = 0
    nullify(v%view)
end subroutine

function hip_dot(x, y) result(d)          ! dot_product(a, b), cg_solvers.f90:131 (device kernel, host vectors)
    real(dp), intent(in) :: x(:), y(:)
    real(dp) :: d
    call hip_check(sgm_dot(int(size(x), c_int64_t), x, y, d, SGM_HOST))
end function

subroutine hip_axpy(alpha, x, y)          ! y = y + alpha * x, cg_solvers.f90:137-138
    real(dp), intent(in) :: alpha, x(:)
    real(dp), intent(inout) :: y(:)
    call hip_check(sgm_axpy(int(size(x), c_int64_t), alpha, x, y, SGM_HOST))
end subroutine

subroutine hip_solver_set_max_iter(s, max_iter)
    ! extension (the reference has no iteration cap): takes effect at the next setup / solve
    class(hip_linear_solver), intent(inout) :: s
    integer, intent(in) :: max_iter
    if (.not. c_associated(s%handle)) then
        print *, 'hip_linear_solver%set_max_iter: call setup first'
        call exit(1)
    endif
    call hip_check(sgm_solver_set_max_iter(s%handle, int(max_iter, c_int64_t)))
end subroutine

subroutine hip_solver_destroy(s)                   ! cg_solvers.f90:199-212
    class(hip_linear_solver), intent(inout) :: s
    if (c_associated(s%handle)) then
        if (s%kind > 10) then
            call hip_check(sgm_pc_destroy(s%handle))
        else
            call hip_check(sgm_solver_destroy(s%handle))
        endif
    endif
    s%handle = c_null_ptr
    s%nn = 0
    s%iterations = 0
    s%initialized = .false.
end subroutine



!==========================================================================!
!==== assembly on the device (sgm_csr_from_edges)                       ====!
!==========================================================================!
subroutine hip_csr_from_edges(A, nrow, ncol, ei, ej, ev)
    ! g%add_edge(i,j)... ; convert_graph_type ; A%set_graph ; A%set_value(i,j,z)... (test/solver_test_jacobi.f90:73-128)
    ! from the edge list in insertion order: repeated edges ignored, the last value written wins.  The arrays the
    ! reference would hold are read back into A%ptr / A%node / A%val.
    type(hip_csr_matrix), intent(inout), target :: A
    integer, intent(in) :: nrow, ncol
    integer(c_int32_t), intent(in) :: ei(:), ej(:)
    real(dp), intent(in) :: ev(:)
    integer(c_int32_t) :: n32, m32, fmt
    integer(c_int64_t) :: nnz, xl
    call hip_check(sgm_csr_from_edges(A%handle, int(nrow, c_int32_t), int(ncol, c_int32_t), int(size(ei), c_int64_t), &
        & ei, ej, ev, SGM_HOST))
    call hip_check(sgm_mat_info(A%handle, n32, m32, nnz, fmt, xl))
    A%nrow = nrow
    A%ncol = ncol
    if (allocated(A%ptr)) deallocate(A%ptr, A%node, A%val)
    allocate(A%ptr(nrow + 1), A%node(nnz), A%val(nnz))
    call hip_csr_download(A)
end subroutine


!==========================================================================!
!==== the composite (sparse_matrix_composites.f90:41-162)               ====!
!==========================================================================!
subroutine hip_comp_set_num_blocks(A, num_row_mats, num_col_mats)      ! :203-223
    class(hip_sparse_matrix), intent(inout) :: A
    integer, intent(in) :: num_row_mats, num_col_mats
    A%num_row_mats = num_row_mats
    A%num_col_mats = num_col_mats
    allocate(A%row_ptr(num_row_mats + 1), A%col_ptr(num_col_mats + 1))
    A%row_ptr = 0
    A%col_ptr = 0
    allocate(A%sub_mats(num_row_mats, num_col_mats))
end subroutine

subroutine hip_comp_set_block_sizes(A, rows, cols)                     ! :228-263
    class(hip_sparse_matrix), intent(inout) :: A
    integer, intent(in) :: rows(:), cols(:)
    integer :: it
    if (.not. allocated(A%sub_mats)) call A%set_num_blocks(size(rows), size(cols))
    A%row_ptr(1) = 1
    A%col_ptr(1) = 1
    do it = 1, A%num_row_mats
        A%row_ptr(it + 1) = A%row_ptr(it) + rows(it)
    enddo
    do it = 1, A%num_col_mats
        A%col_ptr(it + 1) = A%col_ptr(it) + cols(it)
    enddo
    A%nrow = A%row_ptr(A%num_row_mats + 1) - 1
    A%ncol = A%col_ptr(A%num_col_mats + 1) - 1
end subroutine

subroutine hip_comp_set_submatrix(A, it, jt, B)                        ! :1031-1066
    class(hip_sparse_matrix), intent(inout) :: A
    integer, intent(in) :: it, jt
    class(hip_matrix), target :: B
    A%sub_mats(it, jt)%mat => B
end subroutine

subroutine hip_comp_upload(A)
    ! the leaves first (their host edits), then ONE device operator over their handles; it is re-made when a leaf
    ! handle is a new one (the leaf was re-created) or the layout changed
    class(hip_sparse_matrix), intent(inout) :: A
    type(c_ptr), allocatable :: blocks(:)
    logical :: remake
    integer :: it, jt, k
    allocate(blocks(A%num_row_mats * A%num_col_mats))
    do it = 1, A%num_row_mats
        do jt = 1, A%num_col_mats
            k = (it - 1) * A%num_col_mats + jt
            blocks(k) = c_null_ptr
            if (associated(A%sub_mats(it, jt)%mat)) then
                call A%sub_mats(it, jt)%mat%upload()
                blocks(k) = A%sub_mats(it, jt)%mat%handle
            endif
        enddo
    enddo
    remake = .not. c_associated(A%handle) .or. .not. allocated(A%built_from)
    if (.not. remake) then
        do k = 1, size(blocks)
            if (c_associated(blocks(k)) .neqv. c_associated(A%built_from(k))) remake = .true.
            if (c_associated(blocks(k)) .and. c_associated(A%built_from(k))) then
                if (.not. c_associated(blocks(k), A%built_from(k))) remake = .true.
            endif
        enddo
    endif
    if (remake) then
        if (c_associated(A%handle)) call hip_check(sgm_mat_destroy(A%handle))
        call hip_check(sgm_composite_create(A%handle, int(A%num_row_mats, c_int32_t), int(A%num_col_mats, c_int32_t), &
            & A%row_ptr, A%col_ptr, blocks))
        A%built_from = blocks
    endif
end subroutine

subroutine hip_comp_destroy(A)
    class(hip_sparse_matrix), intent(inout) :: A
    if (c_associated(A%handle)) call hip_check(sgm_mat_destroy(A%handle))
    A%handle = c_null_ptr
    if (allocated(A%sub_mats)) deallocate(A%sub_mats, A%row_ptr, A%col_ptr)
    if (allocated(A%built_from)) deallocate(A%built_from)
    A%num_row_mats = 0
    A%num_col_mats = 0
end subroutine


!==========================================================================!
!==== Lanczos (src/eigensolver.f90)                                     ====!
!==========================================================================!
subroutine hip_lanczos(A, T, Q, q1)
    ! lanczos(A, T, Q) (eigensolver.f90:27-90); q1 (optional): the start vector -- the reference draws it from a
    ! time-seeded RNG (:46-52); without it a random one is drawn here the same way (random_number, 2 q - 1)
    class(hip_matrix), intent(inout) :: A
    real(dp), intent(out) :: T(:,:), Q(:,:)
    real(dp), intent(in), optional :: q1(:)
    real(dp), allocatable :: q0(:), Tc(:,:), Qc(:,:)
    integer :: n
    n = size(T, 2)
    allocate(q0(A%nrow), Tc(3, n), Qc(A%nrow, n))
    if (present(q1)) then
        q0 = q1
    else
        call random_number(q0)
        q0 = 2 * q0 - 1
    endif
    call A%upload()
    call hip_check(sgm_lanczos(A%handle, int(n, c_int32_t), q0, Tc, Qc, SGM_HOST))
    T = 0.0_dp
    Q = 0.0_dp
    T(1:3, 1:n) = Tc
    Q(1:A%nrow, 1:n) = Qc
end subroutine

subroutine hip_generalized_lanczos(A, B, solver_for_B, T, Q, q1, pc)
    ! generalized_lanczos(A, B, T, Q) (eigensolver.f90:95-155): the reference finds B's solver in B%solver
    ! (B%set_solver); this layer's matrices carry no solver, so it is an argument (set up for B by the caller)
    class(hip_matrix), intent(inout) :: A, B
    type(hip_linear_solver), intent(inout) :: solver_for_B
    real(dp), intent(out) :: T(:,:), Q(:,:)
    real(dp), intent(in), optional :: q1(:)
    type(hip_linear_solver), intent(inout), optional :: pc
    real(dp), allocatable :: q0(:), Tc(:,:), Qc(:,:)
    type(c_ptr) :: hp
    integer :: n
    n = size(T, 2)
    allocate(q0(A%nrow), Tc(3, n), Qc(A%nrow, n))
    if (present(q1)) then
        q0 = q1
    else
        call random_number(q0)
        q0 = 2 * q0 - 1
    endif
    hp = c_null_ptr
    if (present(pc)) hp = pc%handle
    call A%upload()
    call B%upload()
    call hip_check(sgm_solver_set_tolerance(solver_for_B%handle, solver_for_B%tolerance))
    call hip_check(sgm_generalized_lanczos(A%handle, B%handle, solver_for_B%handle, hp, int(n, c_int32_t), q0, Tc, Qc, &
        & SGM_HOST))
    T = 0.0_dp
    Q = 0.0_dp
    T(1:3, 1:n) = Tc
    Q(1:A%nrow, 1:n) = Qc
end subroutine


!==========================================================================!
!==== row-partitioned multi-GPU: one process per GPU                    ====!
!==========================================================================!
subroutine hip_comm_init(comm, rank, nranks, id_file, device)
    ! rank 0 makes RCCL's unique id, writes it to `id_file` and then creates `id_file`.ready; the other ranks wait
    ! for the marker and read the id.  device (optional): the GPU this process drives (default: rank)
    class(hip_comm), intent(inout) :: comm
    integer, intent(in) :: rank, nranks
    character(len=*), intent(in) :: id_file
    integer, intent(in), optional :: device
    character(kind=c_char) :: id(128)
    integer :: u, dev, waited
    logical :: there
    dev = rank
    if (present(device)) dev = device
    call hip_check(sgm_init(int(dev, c_int)))
    comm%rank = rank
    comm%nranks = nranks
    if (rank == 0) then
        call hip_check(sgm_comm_unique_id(id))
        open(newunit=u, file=id_file, access='stream', form='unformatted', status='replace')
        write(u) id
        close(u)
        open(newunit=u, file=id_file // '.ready', status='replace')
        write(u, *) nranks
        close(u)
    else
        waited = 0
        do
            inquire(file=id_file // '.ready', exist=there)
            if (there) exit
            if (waited > 120 * 100) then
                print *, 'hip_comm%init: rank', rank, 'waited 120 s for ', id_file
                call exit(1)
            endif
            u = c_usleep(10000_c_int)
            waited = waited + 1
        enddo
        open(newunit=u, file=id_file, access='stream', form='unformatted', status='old')
        read(u) id
        close(u)
    endif
    call hip_check(sgm_comm_init(comm%handle, int(rank, c_int), int(nranks, c_int), id))
end subroutine

subroutine hip_comm_destroy(comm)
    class(hip_comm), intent(inout) :: comm
    if (c_associated(comm%handle)) call hip_check(sgm_comm_destroy(comm%handle))
    comm%handle = c_null_ptr
end subroutine

subroutine hip_dist_distribute(Ad, comm, A)
    ! every rank holds the assembled matrix A and keeps its own row block of it on its GPU: contiguous blocks
    ! balanced by stored entries (sgm_partition_rows_by_nnz), boundaries on even rows.  Collective.
    class(hip_dist_csr_matrix), intent(inout) :: Ad
    type(hip_comm), intent(in) :: comm
    type(hip_csr_matrix), intent(in) :: A
    integer(c_int32_t), allocatable :: lptr(:)
    integer(c_int32_t) :: n32, m32, fmt
    integer(c_int64_t) :: nnz, nnz_glob
    integer :: r0, r1, k0, k1
    allocate(Ad%row_starts(comm%nranks + 1))
    call hip_check(sgm_partition_rows_by_nnz(int(A%nrow, c_int32_t), A%ptr, int(comm%nranks, c_int32_t), 2_c_int32_t, &
        & Ad%row_starts))
    r0 = int(Ad%row_starts(comm%rank + 1))
    r1 = int(Ad%row_starts(comm%rank + 2))
    k0 = A%ptr(r0 + 1)
    k1 = A%ptr(r1 + 1)
    nnz = k1 - k0
    allocate(lptr(r1 - r0 + 1))
    lptr = A%ptr(r0 + 1 : r1 + 1) - k0 + 1
    call hip_check(sgm_csr_create_dist(Ad%handle, comm%handle, Ad%row_starts, nnz, lptr, A%node(k0 : k1 - 1), &
        & A%val(k0 : k1 - 1), SGM_HOST))
    call hip_check(sgm_mat_info(Ad%handle, n32, m32, nnz_glob, fmt, Ad%x_len))
    Ad%nrow_global = A%nrow
    Ad%row_first = r0 + 1
    Ad%row_last = r1
    Ad%nrow = r1 - r0
    Ad%ncol = r1 - r0
end subroutine

subroutine hip_dist_upload(A)
    class(hip_dist_csr_matrix), intent(inout) :: A
    if (.not. c_associated(A%handle)) then
        print *, 'hip_dist_csr_matrix used before distribute'
        call exit(1)
    endif
end subroutine

subroutine hip_dist_matvec_add(A, x, y)
    ! x, y: the owned slices; the library reads x as [owned | halo room] and fills the halo part itself
    class(hip_dist_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: y(:)
    real(dp), allocatable :: xext(:)
    allocate(xext(max(A%x_len, 1_c_int64_t)))
    xext = 0.0_dp
    xext(1 : A%nrow) = x(1 : A%nrow)
    call hip_check(sgm_mat_matvec_add(A%handle, xext, y, SGM_HOST))
end subroutine

subroutine hip_dist_matvec(A, x, y)
    class(hip_dist_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: y(:)
    y = 0.0_dp
    call hip_dist_matvec_add(A, x, y)
end subroutine

subroutine hip_dist_destroy(A)
    class(hip_dist_csr_matrix), intent(inout) :: A
    if (c_associated(A%handle)) call hip_check(sgm_mat_destroy(A%handle))
    A%handle = c_null_ptr
    if (allocated(A%row_starts)) deallocate(A%row_starts)
end subroutine

end module sigma_hip
