!==========================================================================!
! hip_binding.f90 -- the REFERENCE-SIDE binding of libsigma_hip.so.        !
!                                                                          !
! This is the file a SiGMA maintainer adds to src/ to put the MI355X path  !
! behind the library's own types.  It `use`s the reference's modules, so   !
! it only compiles where the reference does: oracle/build_ref.sh builds it !
! against oracle/_ref/obj/*.mod and links oracle/hip_binding_test.f90 with !
! libsigma_hip.so (test infrastructure; INTEGRATION.md quotes this file    !
! verbatim, tests/test_cabi_cpu.py checks that it does).                   !
!                                                                          !
!   hip_csr_matrix      extends csr_matrix       cs_matrices.f90:112-151   !
!   hip_ellpack_matrix  extends ellpack_matrix   ellpack_matrices.f90:28   !
!       override matvec_add / matvec_t_add (cs_matrices.f90:66-67; iface   !
!       linear_operator_interface.f90:82-87) and the mutators, which mark  !
!       the device copy stale; A%matvec, A%solve, operator sums/products,  !
!       the reference's own cg()/bicgstab() loops ... all reach the GPU    !
!       through these two bindings.                                        !
!   hip_cg_solver, hip_bicgstab_solver, hip_gmres_solver,                  !
!   hip_jacobi_solver, hip_ldu_solver   extend linear_solver               !
!       (linear_operator_interface.f90:61-73): setup / linear_solve /      !
!       linear_solve_pc / destroy run the whole loop on the device; public !
!       fields iterations, tolerance, nn, initialized as in                !
!       cg_solvers.f90:10-28.  Factories hip_cg(tol) ... return            !
!       class(linear_solver), pointer like cg() (cg_solvers.f90:36-47).    !
!                                                                          !
!   hip_sparse_matrix   extends sparse_matrix (the composite "matrix of    !
!       matrices", sparse_matrix_composites.f90:41-162): set_submatrix     !
!       (:1031-1066) also records hip leaves, so that the block loop of    !
!       composite_matvec_add (:1076-1099) runs leaf by leaf on the device  !
!       AND the device-resident solvers get ONE handle for the whole       !
!       composite (sgm_composite_create).                                  !
!   hip_csr_from_edges  the assembly sequence add_edge ... convert_graph_  !
!       type ... set_graph ... set_value (test/solver_test_jacobi.f90:     !
!       73-128) on the device; the host arrays are read back so that the   !
!       reference's own methods keep working on the matrix.                !
!   hip_lanczos, hip_generalized_lanczos   lanczos(A,T,Q) /                !
!       generalized_lanczos(A,B,T,Q) (eigensolver.f90:27-38,95-108), same  !
!       argument lists; B's solver is B%solver like the reference's        !
!       `call B%solve(w, v)` (:140).                                       !
!   hip_comm, hip_dist_csr_matrix   row-partitioned multi-GPU (nothing in  !
!       the reference; "This loop can be parallelized",                    !
!       sparse_matrix_composites.f90:1086): one process per GPU, the RCCL  !
!       id travels through a file (no MPI needed).                         !
!                                                                          !
! Errors: nonzero status -> print + call exit(1), the reference's own      !
! behaviour (cg_solvers.f90:61-65).  There is no CPU fallback.             !
!==========================================================================!
module hip_c_abi

use iso_c_binding

implicit none

integer(c_int), parameter :: SGM_HOST = 0

interface   ! include/sigma_hip.h
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
    function sgm_csr_set_values(A, val, where) bind(c, name='sgm_csr_set_values') result(rc)
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
    function sgm_ell_set_values(A, val, where) bind(c, name='sgm_ell_set_values') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: val(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec_add(A, x, y, where) bind(c, name='sgm_mat_matvec_add') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec_t_add(A, x, y, where) bind(c, name='sgm_mat_matvec_t_add') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_destroy(A) bind(c, name='sgm_mat_destroy') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: A
        integer(c_int) :: rc
    end function
    function sgm_cg_create(s, tol) bind(c, name='sgm_cg_create') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), intent(out) :: s
        real(c_double), value :: tol
        integer(c_int) :: rc
    end function
    function sgm_bicgstab_create(s, tol) bind(c, name='sgm_bicgstab_create') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), intent(out) :: s
        real(c_double), value :: tol
        integer(c_int) :: rc
    end function
    function sgm_gmres_create(s, tol, restart) bind(c, name='sgm_gmres_create') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_double
        type(c_ptr), intent(out) :: s
        real(c_double), value :: tol
        integer(c_int32_t), value :: restart
        integer(c_int) :: rc
    end function
    function sgm_solver_setup(s, A) bind(c, name='sgm_solver_setup') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: s, A
        integer(c_int) :: rc
    end function
    function sgm_solver_solve(s, A, x, b, pc, where) bind(c, name='sgm_solver_solve') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: s, A, pc
        real(c_double), intent(inout) :: x(*)
        real(c_double), intent(in) :: b(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_solver_info(s, iterations, res2, converged, last) bind(c, name='sgm_solver_info') result(rc)
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
    function sgm_init(device) bind(c, name='sgm_init') result(rc)
        import :: c_int
        integer(c_int), value :: device
        integer(c_int) :: rc
    end function
    function sgm_mat_matvec(A, x, y, where) bind(c, name='sgm_mat_matvec') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: A
        real(c_double), intent(in) :: x(*)
        real(c_double), intent(inout) :: y(*)
        integer(c_int), value :: where
        integer(c_int) :: rc
    end function
    function sgm_mat_info(A, nrow, ncol, nnz, fmt, x_len) bind(c, name='sgm_mat_info') result(rc)
        import :: c_ptr, c_int, c_int32_t, c_int64_t
        type(c_ptr), value :: A
        integer(c_int32_t), intent(out) :: nrow, ncol, fmt
        integer(c_int64_t), intent(out) :: nnz, x_len
        integer(c_int) :: rc
    end function
    function sgm_mat_get(A, name, out, bytes, needed) bind(c, name='sgm_mat_get') result(rc)
        import :: c_ptr, c_int, c_char, c_size_t
        type(c_ptr), value :: A
        character(kind=c_char), intent(in) :: name(*)
        type(c_ptr), value :: out
        integer(c_size_t), value :: bytes
        type(c_ptr), value :: needed
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
    function sgm_composite_create(A, nrb, ncb, row_ptr, col_ptr, blocks) bind(c, name='sgm_composite_create') result(rc)
        import :: c_ptr, c_int, c_int32_t
        type(c_ptr), intent(out) :: A
        integer(c_int32_t), value :: nrb, ncb
        integer(c_int32_t), intent(in) :: row_ptr(*), col_ptr(*)
        type(c_ptr), intent(in) :: blocks(*)
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
    function sgm_comm_destroy(comm) bind(c, name='sgm_comm_destroy') result(rc)
        import :: c_ptr, c_int
        type(c_ptr), value :: comm
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
    function sgm_mat_set_option(A, name, value) bind(c, name='sgm_mat_set_option') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: A
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), value :: value
        integer(c_int) :: rc
    end function
    function sgm_solver_set_tolerance(s, tolerance) bind(c, name='sgm_solver_set_tolerance') result(rc)
        import :: c_ptr, c_int, c_double
        type(c_ptr), value :: s
        real(c_double), value :: tolerance
        integer(c_int) :: rc
    end function
    function sgm_solver_set_option(s, name, value) bind(c, name='sgm_solver_set_option') result(rc)
        import :: c_ptr, c_int, c_char
        type(c_ptr), value :: s
        character(kind=c_char), intent(in) :: name(*)
        integer(c_int), value :: value
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
        integer(c_int32_t), value :: kind
        integer(c_int) :: rc
    end function
    function c_usleep(us) bind(c, name='usleep') result(rc)
        import :: c_int
        integer(c_int), value :: us
        integer(c_int) :: rc
    end function
    function c_strlen(s) bind(c, name='strlen') result(n)
        import :: c_ptr, c_size_t
        type(c_ptr), value :: s
        integer(c_size_t) :: n
    end function
end interface

contains

!--------------------------------------------------------------------------!
subroutine hip_check(rc, what)                                             !
!--------------------------------------------------------------------------!
! The reference's error behaviour (cg_solvers.f90:61-65): print, exit(1).  !
!--------------------------------------------------------------------------!
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: k, n

    if (rc == 0) return
    p = sgm_last_error()
    n = int(c_strlen(p))
    call c_f_pointer(p, msg, [n])
    print *, what, ' failed with sigma_hip status', rc
    print *, (msg(k), k = 1, n)
    print *, 'Terminating.'
    call exit(1)

end subroutine hip_check

end module hip_c_abi




!==========================================================================!
module hip_matrices                                                        !
!==========================================================================!

use iso_c_binding
use types, only: dp
use graph_interfaces
use linear_operator_interface
use cs_graphs
use cs_matrices
use ellpack_matrices
use sparse_matrix_interfaces
use sparse_matrix_composites
use hip_c_abi

implicit none

!--------------------------------------------------------------------------!
type :: hip_device_copy                                                    !
!--------------------------------------------------------------------------!
! matvec_add takes the matrix intent(in) (opvec_add_ifc,                   !
! linear_operator_interface.f90:82-87), so the device state sits behind a  !
! pointer: its target may be brought up to date from inside matvec_add.    !
!--------------------------------------------------------------------------!
    type(c_ptr) :: handle = c_null_ptr
    logical :: values_stale = .true., structure_stale = .true.
end type hip_device_copy


!--------------------------------------------------------------------------!
type, extends(csr_matrix) :: hip_csr_matrix                                !
!--------------------------------------------------------------------------!
    type(hip_device_copy), pointer :: dev => null()
contains
    procedure :: set_graph => hip_csr_set_graph
    procedure :: copy_graph => hip_csr_copy_graph
    procedure :: copy_matrix => hip_csr_copy_matrix
    procedure :: set_value => hip_csr_set_value
    procedure :: add_value => hip_csr_add_value
    procedure :: set_multiple_values => hip_csr_set_multiple_values
    procedure :: add_multiple_values => hip_csr_add_multiple_values
    procedure :: zero => hip_csr_zero
    procedure :: scalar_multiply => hip_csr_scalar_multiply
    procedure :: left_permute => hip_csr_left_permute
    procedure :: right_permute => hip_csr_right_permute
    procedure :: matvec_add => hip_csr_matvec_add
    procedure :: matvec_t_add => hip_csr_matvec_t_add
    procedure :: destroy => hip_csr_destroy
    ! for host code that writes A%val directly
    procedure :: values_changed => hip_csr_values_changed
    procedure :: device_handle => hip_csr_device_handle
    procedure :: set_option => hip_csr_set_option
end type hip_csr_matrix


!--------------------------------------------------------------------------!
type, extends(ellpack_matrix) :: hip_ellpack_matrix                        !
!--------------------------------------------------------------------------!
    type(hip_device_copy), pointer :: dev => null()
contains
    procedure :: set_graph => hip_ell_set_graph
    procedure :: copy_graph => hip_ell_copy_graph
    procedure :: copy_matrix => hip_ell_copy_matrix
    procedure :: set_value => hip_ell_set_value
    procedure :: add_value => hip_ell_add_value
    procedure :: set_multiple_values => hip_ell_set_multiple_values
    procedure :: add_multiple_values => hip_ell_add_multiple_values
    procedure :: zero => hip_ell_zero
    procedure :: scalar_multiply => hip_ell_scalar_multiply
    procedure :: left_permute => hip_ell_left_permute
    procedure :: right_permute => hip_ell_right_permute
    procedure :: matvec_add => hip_ell_matvec_add
    procedure :: matvec_t_add => hip_ell_matvec_t_add
    procedure :: destroy => hip_ell_destroy
    procedure :: values_changed => hip_ell_values_changed
    procedure :: device_handle => hip_ell_device_handle
    procedure :: set_option => hip_ell_set_option
end type hip_ellpack_matrix


!--------------------------------------------------------------------------!
type :: hip_leaf_pointer                                                   !
!--------------------------------------------------------------------------!
    class(sparse_matrix_interface), pointer :: mat => null()
end type hip_leaf_pointer


!--------------------------------------------------------------------------!
type :: hip_composite_copy                                                 !
!--------------------------------------------------------------------------!
! the composite's device operator and the leaf handles it was made of      !
!--------------------------------------------------------------------------!
    type(c_ptr) :: handle = c_null_ptr
    type(c_ptr), allocatable :: blocks(:)
    logical :: layout_stale = .true.
end type hip_composite_copy


!--------------------------------------------------------------------------!
type, extends(sparse_matrix) :: hip_sparse_matrix                          !
!--------------------------------------------------------------------------!
! The composite (sparse_matrix_composites.f90:41-162) over hip leaves.     !
! `sub_mats` is private to the reference's module, so set_submatrix keeps  !
! a record of its own; everything else is inherited: A%matvec_add is the   !
! reference's block loop (:1076-1099), which calls each leaf's matvec_add  !
! -- the device product of hip_csr_matrix / hip_ellpack_matrix.            !
! device_handle(): ONE device operator for the whole composite             !
! (sgm_composite_create over the leaves' handles), what the hip_* solvers  !
! and hip_lanczos run on; composite_matvec_add / _t_add apply it directly  !
! (one call, vectors cross PCIe once instead of once per block).           !
!--------------------------------------------------------------------------!
    type(hip_leaf_pointer), allocatable :: leaves(:,:)
    type(hip_composite_copy), pointer :: cdev => null()
contains
    procedure :: set_submatrix => hip_composite_set_submatrix
    procedure :: device_handle => hip_composite_device_handle
    procedure :: device_matvec_add => hip_composite_device_matvec_add
    procedure :: device_matvec_t_add => hip_composite_device_matvec_t_add
    procedure :: destroy => hip_composite_destroy
end type hip_sparse_matrix


!--------------------------------------------------------------------------!
type :: hip_comm                                                           !
!--------------------------------------------------------------------------!
! One process per GPU; rank r of nranks.  RCCL's 128-byte unique id is     !
! made by rank 0 and handed to the others through a FILE (any shared       !
! directory: no MPI is needed; a host that has MPI broadcasts it instead   !
! and calls sgm_comm_init itself).                                         !
!--------------------------------------------------------------------------!
    integer :: rank = 0, nranks = 1
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure :: init => hip_comm_init
    procedure :: destroy => hip_comm_destroy
end type hip_comm


!--------------------------------------------------------------------------!
type, extends(linear_operator) :: hip_dist_csr_matrix                      !
!--------------------------------------------------------------------------!
! This rank's contiguous row block of a square csr_matrix partitioned over !
! the ranks of a hip_comm (balanced by stored entries,                     !
! sgm_partition_rows_by_nnz).  As a linear_operator it is the LOCAL view:  !
! nrow = ncol = owned rows; matvec_add takes the owned slice of x, the     !
! library fetches the halo entries from the neighbour ranks (ncclSend /    !
! ncclRecv over xGMI) and adds this rank's rows of A x to the owned slice  !
! of y -- bit-identical to the same rows of csr_matvec_add.  The hip_*     !
! Krylov solvers run on it with all-reduced dot products; the reference's  !
! own cg() must not (its dot_product would be a local one).                !
!--------------------------------------------------------------------------!
    integer :: nrow_global = 0, row_first = 0, row_last = 0    ! owned global rows row_first .. row_last (1-based)
    integer(c_int64_t) :: x_len = 0                            ! owned + halo entries a product reads
    integer(c_int64_t), allocatable :: row_starts(:)           ! nranks + 1, 0-based
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure :: distribute => hip_dist_distribute
    procedure :: matvec_add => hip_dist_matvec_add
    procedure :: matvec_t_add => hip_dist_matvec_t_add
    procedure :: destroy => hip_dist_destroy
end type hip_dist_csr_matrix


contains


!--------------------------------------------------------------------------!
subroutine stale(dev, structure)                                           !
!--------------------------------------------------------------------------!
    type(hip_device_copy), pointer, intent(inout) :: dev
    logical, intent(in) :: structure

    if (.not. associated(dev)) allocate(dev)
    dev%values_stale = .true.
    if (structure) dev%structure_stale = .true.

end subroutine stale


!==========================================================================!
!==== hip_csr_matrix                                                   ====!
!==========================================================================!

!--------------------------------------------------------------------------!
function hip_csr_device_handle(A) result(h)                                !
!--------------------------------------------------------------------------!
! The device copy, brought up to date: sgm_csr_create takes g%ptr, g%node  !
! and val exactly as the reference holds them (1-based).                   !
!--------------------------------------------------------------------------!
    class(hip_csr_matrix), intent(in) :: A
    type(c_ptr) :: h
    type(hip_device_copy), pointer :: dev

    dev => A%dev
    if (.not. associated(dev)) then
        print *, 'hip_csr_matrix used before set_graph / copy_graph'
        print *, 'Terminating.'
        call exit(1)
    endif
    if (dev%structure_stale) then
        if (c_associated(dev%handle)) call hip_check(sgm_mat_destroy(dev%handle), 'sgm_mat_destroy')
        call hip_check(sgm_csr_create(dev%handle, A%nrow, A%ncol, &
            & int(A%g%ptr(A%nrow + 1) - 1, c_int64_t), A%g%ptr, A%g%node, A%val, SGM_HOST), 'sgm_csr_create')
    elseif (dev%values_stale) then
        call hip_check(sgm_csr_set_values(dev%handle, A%val, SGM_HOST), 'sgm_csr_set_values')
    endif
    dev%structure_stale = .false.
    dev%values_stale = .false.
    h = dev%handle

end function hip_csr_device_handle


!--------------------------------------------------------------------------!
subroutine hip_csr_matvec_add(A, x, y)                                     !
!--------------------------------------------------------------------------!
! replaces cs_matvec_add -> csr_matvec_add (cs_matrices.f90:500-508,       !
! 600-622); rows are summed in the same order: bit-identical y             !
!--------------------------------------------------------------------------!
    class(hip_csr_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)

    call hip_check(sgm_mat_matvec_add(A%device_handle(), x, y, SGM_HOST), 'sgm_mat_matvec_add')

end subroutine hip_csr_matvec_add


!--------------------------------------------------------------------------!
subroutine hip_csr_matvec_t_add(A, x, y)                                   !
!--------------------------------------------------------------------------!
! replaces cs_matvec_t_add -> csc_matvec_add (cs_matrices.f90:627-647)     !
!--------------------------------------------------------------------------!
    class(hip_csr_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)

    call hip_check(sgm_mat_matvec_t_add(A%device_handle(), x, y, SGM_HOST), 'sgm_mat_matvec_t_add')

end subroutine hip_csr_matvec_t_add


!--------------------------------------------------------------------------!
! Mutators: the reference's own code does the host work, the device copy   !
! is marked stale (values: re-uploaded; structure: re-created) and brought !
! up to date by the next product or solve.                                 !
!--------------------------------------------------------------------------!
subroutine hip_csr_set_graph(A, g)
    class(hip_csr_matrix), intent(inout) :: A
    class(graph_interface), target, intent(in) :: g
    call A%csr_matrix%set_graph(g)
    call stale(A%dev, .true.)
end subroutine hip_csr_set_graph

subroutine hip_csr_copy_graph(A, g)
    class(hip_csr_matrix), intent(inout) :: A
    class(graph_interface), intent(in) :: g
    call A%csr_matrix%copy_graph(g)
    call stale(A%dev, .true.)
end subroutine hip_csr_copy_graph

subroutine hip_csr_copy_matrix(A, B, trans)
    class(hip_csr_matrix), intent(inout) :: A
    class(sparse_matrix_interface), intent(in) :: B
    logical, intent(in), optional :: trans
    call A%csr_matrix%copy_matrix(B, trans)
    call stale(A%dev, .true.)
end subroutine hip_csr_copy_matrix

subroutine hip_csr_set_value(A, i, j, z)
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    call A%csr_matrix%set_value(i, j, z)
    call stale(A%dev, .false.)
end subroutine hip_csr_set_value

subroutine hip_csr_add_value(A, i, j, z)
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    call A%csr_matrix%add_value(i, j, z)
    call stale(A%dev, .false.)
end subroutine hip_csr_add_value

subroutine hip_csr_set_multiple_values(A, is, js, B)
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: is(:), js(:)
    real(dp), intent(in) :: B(:,:)
    call A%csr_matrix%set_multiple_values(is, js, B)
    call stale(A%dev, .false.)
end subroutine hip_csr_set_multiple_values

subroutine hip_csr_add_multiple_values(A, is, js, B)
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: is(:), js(:)
    real(dp), intent(in) :: B(:,:)
    call A%csr_matrix%add_multiple_values(is, js, B)
    call stale(A%dev, .false.)
end subroutine hip_csr_add_multiple_values

subroutine hip_csr_zero(A)
    class(hip_csr_matrix), intent(inout) :: A
    call A%csr_matrix%zero()
    call stale(A%dev, .false.)
end subroutine hip_csr_zero

subroutine hip_csr_scalar_multiply(A, alpha)
    class(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: alpha
    call A%csr_matrix%scalar_multiply(alpha)
    call stale(A%dev, .false.)
end subroutine hip_csr_scalar_multiply

subroutine hip_csr_left_permute(A, p)
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: p(:)
    call A%csr_matrix%left_permute(p)
    call stale(A%dev, .true.)
end subroutine hip_csr_left_permute

subroutine hip_csr_right_permute(A, p)
    class(hip_csr_matrix), intent(inout) :: A
    integer, intent(in) :: p(:)
    call A%csr_matrix%right_permute(p)
    call stale(A%dev, .true.)
end subroutine hip_csr_right_permute

subroutine hip_csr_values_changed(A)
    class(hip_csr_matrix), intent(inout) :: A
    call stale(A%dev, .false.)
end subroutine hip_csr_values_changed

subroutine hip_csr_set_option(A, name, value)
    ! this matrix's own kernel-selection option (sgm_mat_set_option); a structure change re-creates the device copy
    ! with the process-wide defaults again
    class(hip_csr_matrix), intent(in) :: A
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    call hip_check(sgm_mat_set_option(A%device_handle(), trim(name) // c_null_char, value), 'sgm_mat_set_option')
end subroutine hip_csr_set_option

subroutine hip_csr_destroy(A)
    class(hip_csr_matrix), intent(inout) :: A
    if (associated(A%dev)) then
        if (c_associated(A%dev%handle)) call hip_check(sgm_mat_destroy(A%dev%handle), 'sgm_mat_destroy')
        deallocate(A%dev)
    endif
    call A%csr_matrix%destroy()
end subroutine hip_csr_destroy


!==========================================================================!
!==== hip_ellpack_matrix                                               ====!
!==========================================================================!

!--------------------------------------------------------------------------!
function hip_ell_device_handle(A) result(h)                                !
!--------------------------------------------------------------------------!
! sgm_ell_create takes g%node(max_d,n) and val(max_d,n) column-major with  !
! their padding (ellpack_graphs.f90:14,164) as they are.                   !
!--------------------------------------------------------------------------!
    class(hip_ellpack_matrix), intent(in) :: A
    type(c_ptr) :: h
    type(hip_device_copy), pointer :: dev

    dev => A%dev
    if (.not. associated(dev)) then
        print *, 'hip_ellpack_matrix used before set_graph / copy_graph'
        print *, 'Terminating.'
        call exit(1)
    endif
    if (dev%structure_stale) then
        if (c_associated(dev%handle)) call hip_check(sgm_mat_destroy(dev%handle), 'sgm_mat_destroy')
        call hip_check(sgm_ell_create(dev%handle, A%nrow, A%ncol, A%g%max_d, A%g%node, A%val, SGM_HOST), &
            & 'sgm_ell_create')
    elseif (dev%values_stale) then
        call hip_check(sgm_ell_set_values(dev%handle, A%val, SGM_HOST), 'sgm_ell_set_values')
    endif
    dev%structure_stale = .false.
    dev%values_stale = .false.
    h = dev%handle

end function hip_ell_device_handle


!--------------------------------------------------------------------------!
subroutine hip_ell_matvec_add(A, x, y)                                     !
!--------------------------------------------------------------------------!
! replaces ellpack_matvec_add (ellpack_matrices.f90:640-665)               !
!--------------------------------------------------------------------------!
    class(hip_ellpack_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)

    call hip_check(sgm_mat_matvec_add(A%device_handle(), x, y, SGM_HOST), 'sgm_mat_matvec_add')

end subroutine hip_ell_matvec_add


!--------------------------------------------------------------------------!
subroutine hip_ell_matvec_t_add(A, x, y)                                   !
!--------------------------------------------------------------------------!
! replaces ellpack_matvec_t_add (ellpack_matrices.f90:670-693)             !
!--------------------------------------------------------------------------!
    class(hip_ellpack_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)

    call hip_check(sgm_mat_matvec_t_add(A%device_handle(), x, y, SGM_HOST), 'sgm_mat_matvec_t_add')

end subroutine hip_ell_matvec_t_add


subroutine hip_ell_set_graph(A, g)
    class(hip_ellpack_matrix), intent(inout) :: A
    class(graph_interface), target, intent(in) :: g
    call A%ellpack_matrix%set_graph(g)
    call stale(A%dev, .true.)
end subroutine hip_ell_set_graph

subroutine hip_ell_copy_graph(A, g)
    class(hip_ellpack_matrix), intent(inout) :: A
    class(graph_interface), intent(in) :: g
    call A%ellpack_matrix%copy_graph(g)
    call stale(A%dev, .true.)
end subroutine hip_ell_copy_graph

subroutine hip_ell_copy_matrix(A, B, trans)
    class(hip_ellpack_matrix), intent(inout) :: A
    class(sparse_matrix_interface), intent(in) :: B
    logical, intent(in), optional :: trans
    call A%ellpack_matrix%copy_matrix(B, trans)
    call stale(A%dev, .true.)
end subroutine hip_ell_copy_matrix

subroutine hip_ell_set_value(A, i, j, z)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    call A%ellpack_matrix%set_value(i, j, z)
    ! set_value on an entry outside the pattern re-builds the graph (ellpack_matrices.f90
    ! set_unallocated_matrix_value): treat the structure as changed too
    call stale(A%dev, .true.)
end subroutine hip_ell_set_value

subroutine hip_ell_add_value(A, i, j, z)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: i, j
    real(dp), intent(in) :: z
    call A%ellpack_matrix%add_value(i, j, z)
    call stale(A%dev, .true.)
end subroutine hip_ell_add_value

subroutine hip_ell_set_multiple_values(A, is, js, B)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: is(:), js(:)
    real(dp), intent(in) :: B(:,:)
    call A%ellpack_matrix%set_multiple_values(is, js, B)
    call stale(A%dev, .true.)
end subroutine hip_ell_set_multiple_values

subroutine hip_ell_add_multiple_values(A, is, js, B)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: is(:), js(:)
    real(dp), intent(in) :: B(:,:)
    call A%ellpack_matrix%add_multiple_values(is, js, B)
    call stale(A%dev, .true.)
end subroutine hip_ell_add_multiple_values

subroutine hip_ell_zero(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    call A%ellpack_matrix%zero()
    call stale(A%dev, .false.)
end subroutine hip_ell_zero

subroutine hip_ell_scalar_multiply(A, alpha)
    class(hip_ellpack_matrix), intent(inout) :: A
    real(dp), intent(in) :: alpha
    call A%ellpack_matrix%scalar_multiply(alpha)
    call stale(A%dev, .false.)
end subroutine hip_ell_scalar_multiply

subroutine hip_ell_left_permute(A, p)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: p(:)
    call A%ellpack_matrix%left_permute(p)
    call stale(A%dev, .true.)
end subroutine hip_ell_left_permute

subroutine hip_ell_right_permute(A, p)
    class(hip_ellpack_matrix), intent(inout) :: A
    integer, intent(in) :: p(:)
    call A%ellpack_matrix%right_permute(p)
    call stale(A%dev, .true.)
end subroutine hip_ell_right_permute

subroutine hip_ell_values_changed(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    call stale(A%dev, .false.)
end subroutine hip_ell_values_changed

subroutine hip_ell_set_option(A, name, value)
    class(hip_ellpack_matrix), intent(in) :: A
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    call hip_check(sgm_mat_set_option(A%device_handle(), trim(name) // c_null_char, value), 'sgm_mat_set_option')
end subroutine hip_ell_set_option

subroutine hip_ell_destroy(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    if (associated(A%dev)) then
        if (c_associated(A%dev%handle)) call hip_check(sgm_mat_destroy(A%dev%handle), 'sgm_mat_destroy')
        deallocate(A%dev)
    endif
    call A%ellpack_matrix%destroy()
end subroutine hip_ell_destroy


!==========================================================================!
!==== hip_csr_from_edges: assembly on the device                       ====!
!==========================================================================!

!--------------------------------------------------------------------------!
subroutine hip_csr_from_edges(A, nrow, ncol, ei, ej, ev)                   !
!--------------------------------------------------------------------------!
! What the reference's tests do on the host (solver_test_jacobi.f90:73-128)!
!     g%add_edge(i,j) ... ; convert_graph_type(g, "compressed sparse") ;   !
!     A%init ; A%set_graph(g) ; A%set_value(i,j,z) ...                     !
! done by the device from the edge list in INSERTION order (repeated edges !
! ignored like ll_graphs.f90:355-370, the last value written wins like     !
! cs_matrices.f90:840-863): sgm_csr_from_edges.  The arrays are read back  !
! into an ordinary cs_graph + val, bit-identical to the host sequence, so  !
! every inherited method of csr_matrix keeps working on A.                 !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix), intent(inout), target :: A
    integer, intent(in) :: nrow, ncol
    integer(c_int32_t), intent(in) :: ei(:), ej(:)
    real(dp), intent(in) :: ev(:)
    type(cs_graph), pointer :: g
    type(c_ptr) :: h
    integer(c_int32_t) :: n32, m32, fmt
    integer(c_int64_t) :: nnz, xl

    call hip_check(sgm_csr_from_edges(h, nrow, ncol, int(size(ei), c_int64_t), ei, ej, ev, SGM_HOST), 'sgm_csr_from_edges')
    call hip_check(sgm_mat_info(h, n32, m32, nnz, fmt, xl), 'sgm_mat_info')
    allocate(g)
    call g%init(nrow, ncol)
    deallocate(g%node)
    allocate(g%node(nnz))
    call download(h, 'ptr', c_loc(g%ptr), 4_c_size_t * (nrow + 1))
    if (nnz > 0) call download(h, 'node', c_loc(g%node), 4_c_size_t * nnz)
    g%ne = int(nnz)
    g%max_d = 0
    if (nrow > 0) g%max_d = maxval(g%ptr(2 : nrow + 1) - g%ptr(1 : nrow))
    call A%init(nrow, ncol)
    call A%set_graph(g)                  ! (marks the device copy stale; it is not: see below)
    if (nnz > 0) call download(h, 'val', c_loc(A%val), 8_c_size_t * nnz)
    A%dev%handle = h
    A%dev%structure_stale = .false.
    A%dev%values_stale = .false.

contains
    subroutine download(h, name, p, bytes)
        type(c_ptr), intent(in) :: h, p
        character(len=*), intent(in) :: name
        integer(c_size_t), intent(in) :: bytes
        call hip_check(sgm_mat_get(h, name // c_null_char, p, bytes, c_null_ptr), 'sgm_mat_get')
    end subroutine download

end subroutine hip_csr_from_edges


!==========================================================================!
!==== hip_sparse_matrix: the composite over hip leaves                 ====!
!==========================================================================!

!--------------------------------------------------------------------------!
subroutine hip_composite_set_submatrix(A, it, jt, B)                       !
!--------------------------------------------------------------------------!
! composite_mat_set_submatrix (sparse_matrix_composites.f90:1031-1066) +   !
! a record of the leaf for device_handle().                                !
!--------------------------------------------------------------------------!
    class(hip_sparse_matrix), intent(inout) :: A
    integer, intent(in) :: it, jt
    class(sparse_matrix_interface), target :: B

    call A%sparse_matrix%set_submatrix(it, jt, B)
    if (.not. allocated(A%leaves)) allocate(A%leaves(A%num_row_mats, A%num_col_mats))
    A%leaves(it, jt)%mat => B
    if (.not. associated(A%cdev)) allocate(A%cdev)
    A%cdev%layout_stale = .true.

end subroutine hip_composite_set_submatrix


!--------------------------------------------------------------------------!
function hip_composite_device_handle(A) result(h)                          !
!--------------------------------------------------------------------------!
! One device operator for the whole composite: the leaves' handles (each   !
! brought up to date first) in the reference's block layout, row_ptr /     !
! col_ptr as the composite keeps them (1-based offsets).                   !
!--------------------------------------------------------------------------!
    class(hip_sparse_matrix), intent(in) :: A
    type(c_ptr) :: h
    type(hip_composite_copy), pointer :: dev
    type(c_ptr), allocatable :: blocks(:)
    logical :: remake
    integer :: it, jt, k

    dev => A%cdev
    if (.not. associated(dev) .or. .not. allocated(A%leaves)) then
        print *, 'hip_sparse_matrix used before set_submatrix'
        print *, 'Terminating.'
        call exit(1)
    endif
    ! the leaves' own handles follow their host edits (values are re-uploaded in place; a structure change re-creates
    ! the handle); the composite operator only points at them, so it is re-made whenever a leaf handle is a new one
    allocate(blocks(A%num_row_mats * A%num_col_mats))
    do it = 1, A%num_row_mats
        do jt = 1, A%num_col_mats
            blocks((it - 1) * A%num_col_mats + jt) = leaf_handle(A%leaves(it, jt)%mat)
        enddo
    enddo
    remake = dev%layout_stale .or. .not. allocated(dev%blocks)
    if (.not. remake) then
        do k = 1, size(blocks)
            if (c_associated(blocks(k)) .neqv. c_associated(dev%blocks(k))) remake = .true.
            if (c_associated(blocks(k)) .and. c_associated(dev%blocks(k))) then
                if (.not. c_associated(blocks(k), dev%blocks(k))) remake = .true.
            endif
        enddo
    endif
    if (remake) then
        if (c_associated(dev%handle)) call hip_check(sgm_mat_destroy(dev%handle), 'sgm_mat_destroy')
        call hip_check(sgm_composite_create(dev%handle, A%num_row_mats, A%num_col_mats, A%row_ptr, A%col_ptr, blocks), &
            & 'sgm_composite_create')
        dev%blocks = blocks
        dev%layout_stale = .false.
    endif
    h = dev%handle

end function hip_composite_device_handle


!--------------------------------------------------------------------------!
function leaf_handle(B) result(h)                                          !
!--------------------------------------------------------------------------!
    class(sparse_matrix_interface), pointer, intent(in) :: B
    type(c_ptr) :: h

    h = c_null_ptr
    if (.not. associated(B)) return             ! an empty block
    select type(B)
        class is(hip_csr_matrix)
            h = B%device_handle()
        class is(hip_ellpack_matrix)
            h = B%device_handle()
        class default
            print *, 'hip_sparse_matrix%device_handle: every leaf must be a hip_csr_matrix or hip_ellpack_matrix'
            print *, 'Terminating.'
            call exit(1)
    end select

end function leaf_handle


!--------------------------------------------------------------------------!
subroutine hip_composite_device_matvec_add(A, x, y)                        !
!--------------------------------------------------------------------------!
! composite_matvec_add (:1076-1099) as ONE device call: same block order,  !
! same row sums -- bit-identical to the inherited A%matvec_add.            !
!--------------------------------------------------------------------------!
    class(hip_sparse_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)

    call hip_check(sgm_mat_matvec_add(A%device_handle(), x, y, SGM_HOST), 'sgm_mat_matvec_add')

end subroutine hip_composite_device_matvec_add


!--------------------------------------------------------------------------!
subroutine hip_composite_device_matvec_t_add(A, x, y)                      !
!--------------------------------------------------------------------------!
    class(hip_sparse_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)

    call hip_check(sgm_mat_matvec_t_add(A%device_handle(), x, y, SGM_HOST), 'sgm_mat_matvec_t_add')

end subroutine hip_composite_device_matvec_t_add


!--------------------------------------------------------------------------!
subroutine hip_composite_destroy(A)                                        !
!--------------------------------------------------------------------------!
    class(hip_sparse_matrix), intent(inout) :: A

    if (associated(A%cdev)) then
        if (c_associated(A%cdev%handle)) call hip_check(sgm_mat_destroy(A%cdev%handle), 'sgm_mat_destroy')
        deallocate(A%cdev)
    endif
    if (allocated(A%leaves)) deallocate(A%leaves)
    call A%sparse_matrix%destroy()

end subroutine hip_composite_destroy


!==========================================================================!
!==== hip_comm, hip_dist_csr_matrix: row-partitioned multi-GPU         ====!
!==========================================================================!

!--------------------------------------------------------------------------!
subroutine hip_comm_init(comm, rank, nranks, id_file, device)              !
!--------------------------------------------------------------------------!
! rank 0 makes RCCL's unique id and writes it to `id_file`, then creates   !
! `id_file`.ready; the other ranks wait for the marker and read the id.    !
! device (optional): the GPU this process drives (default: rank).          !
!--------------------------------------------------------------------------!
    class(hip_comm), intent(inout) :: comm
    integer, intent(in) :: rank, nranks
    character(len=*), intent(in) :: id_file
    integer, intent(in), optional :: device
    character(kind=c_char) :: id(128)
    integer :: u, dev, waited
    logical :: there

    dev = rank
    if (present(device)) dev = device
    call hip_check(sgm_init(dev), 'sgm_init')
    comm%rank = rank
    comm%nranks = nranks
    if (rank == 0) then
        call hip_check(sgm_comm_unique_id(id), 'sgm_comm_unique_id')
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
                print *, 'Terminating.'
                call exit(1)
            endif
            u = c_usleep(10000)
            waited = waited + 1
        enddo
        open(newunit=u, file=id_file, access='stream', form='unformatted', status='old')
        read(u) id
        close(u)
    endif
    call hip_check(sgm_comm_init(comm%handle, rank, nranks, id), 'sgm_comm_init')

end subroutine hip_comm_init


subroutine hip_comm_destroy(comm)
    class(hip_comm), intent(inout) :: comm
    if (c_associated(comm%handle)) call hip_check(sgm_comm_destroy(comm%handle), 'sgm_comm_destroy')
    comm%handle = c_null_ptr
end subroutine hip_comm_destroy


!--------------------------------------------------------------------------!
subroutine hip_dist_distribute(Ad, comm, A)                                !
!--------------------------------------------------------------------------!
! Every rank holds the assembled csr_matrix A (as after the reference's    !
! assembly sequence) and keeps its own row block of it on its GPU:         !
! contiguous blocks balanced by stored entries, boundaries on even rows.   !
! Collective over the ranks of `comm`.                                     !
!--------------------------------------------------------------------------!
    class(hip_dist_csr_matrix), intent(inout) :: Ad
    type(hip_comm), intent(in) :: comm
    class(csr_matrix), intent(in) :: A
    integer(c_int32_t), allocatable :: lptr(:)
    integer(c_int32_t) :: n32, m32, fmt
    integer(c_int64_t) :: nnz, nnz_glob
    integer :: r0, r1, k0, k1

    if (A%nrow /= A%ncol) then
        print *, 'hip_dist_csr_matrix%distribute: square matrices only'
        print *, 'Terminating.'
        call exit(1)
    endif
    allocate(Ad%row_starts(comm%nranks + 1))
    call hip_check(sgm_partition_rows_by_nnz(A%nrow, A%g%ptr, comm%nranks, 2, Ad%row_starts), 'sgm_partition_rows_by_nnz')
    r0 = int(Ad%row_starts(comm%rank + 1))           ! 0-based first owned row
    r1 = int(Ad%row_starts(comm%rank + 2))
    k0 = A%g%ptr(r0 + 1)
    k1 = A%g%ptr(r1 + 1)
    nnz = k1 - k0
    allocate(lptr(r1 - r0 + 1))
    lptr = A%g%ptr(r0 + 1 : r1 + 1) - k0 + 1
    ! (zero-length array sections are legal actual arguments: a rank without entries passes them as they are)
    call hip_check(sgm_csr_create_dist(Ad%handle, comm%handle, Ad%row_starts, nnz, lptr, A%g%node(k0 : k1 - 1), &
        & A%val(k0 : k1 - 1), SGM_HOST), 'sgm_csr_create_dist')
    call hip_check(sgm_mat_info(Ad%handle, n32, m32, nnz_glob, fmt, Ad%x_len), 'sgm_mat_info')
    Ad%nrow_global = A%nrow
    Ad%row_first = r0 + 1
    Ad%row_last = r1
    Ad%nrow = r1 - r0
    Ad%ncol = r1 - r0

end subroutine hip_dist_distribute


!--------------------------------------------------------------------------!
subroutine hip_dist_matvec_add(A, x, y)                                    !
!--------------------------------------------------------------------------!
! x, y: the owned slices.  The library reads x as [owned | halo room] and  !
! fills the halo part itself, so the slice is staged in a vector that has  !
! the room.                                                                !
!--------------------------------------------------------------------------!
    class(hip_dist_csr_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)
    real(dp), allocatable :: xext(:)

    allocate(xext(max(A%x_len, 1_c_int64_t)))
    xext = 0.0_dp
    xext(1 : A%nrow) = x(1 : A%nrow)
    call hip_check(sgm_mat_matvec_add(A%handle, xext, y, SGM_HOST), 'sgm_mat_matvec_add')

end subroutine hip_dist_matvec_add


subroutine hip_dist_matvec_t_add(A, x, y)
    class(hip_dist_csr_matrix), intent(in) :: A
    real(dp), intent(in)    :: x(:)
    real(dp), intent(inout) :: y(:)
    ! collective: A^T is built once as another distributed matrix (every entry travels to the rank owning its column)
    call hip_check(sgm_mat_matvec_t_add(A%handle, x, y, SGM_HOST), 'sgm_mat_matvec_t_add')
end subroutine hip_dist_matvec_t_add


subroutine hip_dist_destroy(A)
    class(hip_dist_csr_matrix), intent(inout) :: A
    if (c_associated(A%handle)) call hip_check(sgm_mat_destroy(A%handle), 'sgm_mat_destroy')
    A%handle = c_null_ptr
    if (allocated(A%row_starts)) deallocate(A%row_starts)
end subroutine hip_dist_destroy


end module hip_matrices




!==========================================================================!
module hip_solvers                                                         !
!==========================================================================!

use iso_c_binding
use types, only: dp
use linear_operator_interface
use hip_c_abi
use hip_matrices

implicit none

integer, parameter, private :: KIND_CG = 1, KIND_BICGSTAB = 2, KIND_GMRES = 3
integer, parameter, private :: PC_JACOBI = 1, PC_LDU = 2

!--------------------------------------------------------------------------!
type, extends(linear_solver) :: hip_krylov_solver                          !
!--------------------------------------------------------------------------!
! cg_solver / bicgstab_solver (cg_solvers.f90:10-28,                       !
! bicgstab_solvers.f90:10-29) with the loop on the device: the work        !
! vectors live in HBM behind `handle`.                                     !
!--------------------------------------------------------------------------!
    integer :: iterations = 0
    real(dp) :: tolerance = 1.0d-16
    integer :: restart = 30
    integer :: kind = KIND_CG
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure :: setup => hip_krylov_setup
    procedure :: set_params => hip_krylov_set_params
    procedure :: set_option => hip_krylov_set_option
    procedure :: linear_solve => hip_krylov_solve
    procedure :: linear_solve_pc => hip_krylov_solve_pc
    procedure :: destroy => hip_krylov_destroy
end type hip_krylov_solver


!--------------------------------------------------------------------------!
type, extends(linear_solver) :: hip_preconditioner                         !
!--------------------------------------------------------------------------!
! jacobi_solver / sparse_ldu_solver (jacobi_solvers.f90:10-21,             !
! ldu_solvers.f90:15-62): setup extracts / factors, linear_solve applies.  !
!--------------------------------------------------------------------------!
    integer :: kind = PC_JACOBI
    type(c_ptr) :: handle = c_null_ptr
contains
    procedure :: setup => hip_pc_setup
    procedure :: set_option => hip_pc_set_option
    procedure :: info => hip_pc_info
    procedure :: linear_solve => hip_pc_solve
    procedure :: destroy => hip_pc_destroy
end type hip_preconditioner


contains


!--------------------------------------------------------------------------!
function matrix_handle(A) result(h)                                        !
!--------------------------------------------------------------------------!
! (public: hip_eigensolver uses it too)                                    !
!--------------------------------------------------------------------------!
    class(linear_operator), intent(in) :: A
    type(c_ptr) :: h

    h = c_null_ptr
    select type(A)
        class is(hip_csr_matrix)
            h = A%device_handle()
        class is(hip_ellpack_matrix)
            h = A%device_handle()
        class is(hip_sparse_matrix)
            h = A%device_handle()
        class is(hip_dist_csr_matrix)
            h = A%handle
        class default
            print *, 'The hip_* solvers need a hip_csr_matrix, hip_ellpack_matrix, hip_sparse_matrix or hip_dist_csr_matrix;'
            print *, 'use cg() / bicgstab() for other operators.'
            print *, 'Terminating.'
            call exit(1)
    end select

end function matrix_handle


!--------------------------------------------------------------------------!
! Factories, like cg(tolerance) (cg_solvers.f90:36-47)                     !
!--------------------------------------------------------------------------!
function hip_cg(tolerance) result(solver)
    real(dp), intent(in), optional :: tolerance
    class(linear_solver), pointer :: solver
    solver => new_krylov(KIND_CG, tolerance)
end function hip_cg

function hip_bicgstab(tolerance) result(solver)
    real(dp), intent(in), optional :: tolerance
    class(linear_solver), pointer :: solver
    solver => new_krylov(KIND_BICGSTAB, tolerance)
end function hip_bicgstab

function hip_gmres(tolerance, restart) result(solver)
    real(dp), intent(in), optional :: tolerance
    integer, intent(in), optional :: restart
    class(linear_solver), pointer :: solver
    solver => new_krylov(KIND_GMRES, tolerance, restart)
end function hip_gmres

function new_krylov(kind, tolerance, restart) result(solver)
    integer, intent(in) :: kind
    real(dp), intent(in), optional :: tolerance
    integer, intent(in), optional :: restart
    class(linear_solver), pointer :: solver
    type(hip_krylov_solver), pointer :: s

    allocate(s)
    s%kind = kind
    if (present(tolerance)) s%tolerance = tolerance      ! default 1e-16: cg_solvers.f90:106
    if (present(restart)) s%restart = restart
    solver => s
end function new_krylov

function hip_jacobi() result(pc)
    class(linear_solver), pointer :: pc
    type(hip_preconditioner), pointer :: p
    allocate(p)
    p%kind = PC_JACOBI
    pc => p
end function hip_jacobi

function hip_ldu(reorder) result(pc)       ! ldu(incomplete = .true., level = 0), ldu_solvers.f90:73-86
    ! reorder = "colour" (an extension, off by default): ILDU(0) of the colour-ordered matrix P A P^T, P = the reference's
    ! greedy_color_ordering of A's graph (permutations.f90:162-205), applied as z = P^T M^-1 P r.  A, b and x stay as the
    ! caller holds them; the factors have one dependency level per colour, so an apply is a few bandwidth-bound launches
    ! instead of a chain of nx + ny levels.  The iteration counts are those of the permuted system.
    character(len=*), intent(in), optional :: reorder
    class(linear_solver), pointer :: pc
    type(hip_preconditioner), pointer :: p
    allocate(p)
    p%kind = PC_LDU
    if (present(reorder)) then
        if (reorder == "colour" .or. reorder == "color") then
            call hip_pc_set_option(p, "ildu_reorder", 1)
        elseif (reorder /= "natural") then
            print *, "hip_ldu: reorder is colour or natural"
            print *, "Terminating."
            call exit(1)
        endif
    endif
    pc => p
end function hip_ldu


!--------------------------------------------------------------------------!
subroutine krylov_handle(solver)                                           !
!--------------------------------------------------------------------------!
! the factory object (hip_cg(tol) ...) becomes a library handle; it has    !
! seen no matrix yet, so options can be set on it before setup             !
!--------------------------------------------------------------------------!
    class(hip_krylov_solver), intent(inout) :: solver

    if (c_associated(solver%handle)) return
    select case(solver%kind)
        case(KIND_CG)
            call hip_check(sgm_cg_create(solver%handle, solver%tolerance), 'sgm_cg_create')
        case(KIND_BICGSTAB)
            call hip_check(sgm_bicgstab_create(solver%handle, solver%tolerance), 'sgm_bicgstab_create')
        case default
            call hip_check(sgm_gmres_create(solver%handle, solver%tolerance, solver%restart), 'sgm_gmres_create')
    end select

end subroutine krylov_handle


!--------------------------------------------------------------------------!
subroutine hip_krylov_set_params(solver, tolerance)                        !
!--------------------------------------------------------------------------!
! cg_set_params (cg_solvers.f90:95-111) / bicgstab_set_params              !
! (bicgstab_solvers.f90:103-119): may be called again at any time; the     !
! field is pushed to the handle in front of every solve (run)              !
!--------------------------------------------------------------------------!
    class(hip_krylov_solver), intent(inout) :: solver
    real(dp), intent(in), optional :: tolerance

    if (present(tolerance)) then
        solver%tolerance = tolerance
    else
        solver%tolerance = 1.0d-16
    endif

end subroutine hip_krylov_set_params


!--------------------------------------------------------------------------!
subroutine hip_krylov_set_option(solver, name, value)                      !
!--------------------------------------------------------------------------!
! this solver's own option ("dot_order", "cg_small", "krylov_graph" ...:   !
! include/sigma_hip.h); e.g. solver%set_option("dot_order", 1) makes CG /  !
! BiCGStab add their dot products in the reference build's order -- the    !
! iterates are then the CPU build's bit for bit (validation runs)          !
!--------------------------------------------------------------------------!
    class(hip_krylov_solver), intent(inout) :: solver
    character(len=*), intent(in) :: name
    integer, intent(in) :: value

    call krylov_handle(solver)
    call hip_check(sgm_solver_set_option(solver%handle, trim(name) // c_null_char, value), 'sgm_solver_set_option')

end subroutine hip_krylov_set_option


!--------------------------------------------------------------------------!
subroutine hip_krylov_setup(solver, A)                                     !
!--------------------------------------------------------------------------!
    class(hip_krylov_solver), intent(inout) :: solver
    class(linear_operator), intent(in) :: A

    if (A%ncol /= A%nrow) then      ! cg_solvers.f90:61-65
        print *, 'Cannot make a Krylov solver for a non-square matrix'
        print *, 'Terminating.'
        call exit(1)
    endif
    solver%nn = A%nrow
    solver%iterations = 0
    call krylov_handle(solver)
    call hip_check(sgm_solver_setup(solver%handle, matrix_handle(A)), 'sgm_solver_setup')
    solver%initialized = .true.

end subroutine hip_krylov_setup


!--------------------------------------------------------------------------!
subroutine hip_krylov_solve(solver, A, x, b)                               !
!--------------------------------------------------------------------------!
! replaces cg_solve (cg_solvers.f90:116-150) / bicgstab_solve              !
! (bicgstab_solvers.f90:124-177): x and b cross PCIe once per solve        !
!--------------------------------------------------------------------------!
    class(hip_krylov_solver), intent(inout) :: solver
    class(linear_operator), intent(in) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)

    call run(solver, matrix_handle(A), x, b, c_null_ptr)

end subroutine hip_krylov_solve


!--------------------------------------------------------------------------!
subroutine hip_krylov_solve_pc(solver, A, x, b, pc)                        !
!--------------------------------------------------------------------------!
! replaces cg_solve_pc (cg_solvers.f90:155-194) / bicgstab_solve_pc        !
! (bicgstab_solvers.f90:182-237)                                           !
!--------------------------------------------------------------------------!
    class(hip_krylov_solver), intent(inout) :: solver
    class(linear_operator), intent(in) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    class(linear_solver), intent(inout) :: pc

    select type(pc)
        class is(hip_preconditioner)
            call run(solver, matrix_handle(A), x, b, pc%handle)
        class default
            print *, 'A hip_* solver needs a hip_jacobi() / hip_ldu() preconditioner'
            print *, 'Terminating.'
            call exit(1)
    end select

end subroutine hip_krylov_solve_pc


subroutine run(solver, hA, x, b, hpc)
    class(hip_krylov_solver), intent(inout) :: solver
    type(c_ptr), intent(in) :: hA, hpc
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    integer(c_int64_t) :: its, last
    real(c_double) :: res2
    integer(c_int32_t) :: conv

    ! solver%tolerance is a public field the reference's loop reads at every solve (cg_solvers.f90:133): an edit made after
    ! the handle exists -- directly or through set_params -- must count
    call hip_check(sgm_solver_set_tolerance(solver%handle, solver%tolerance), 'sgm_solver_set_tolerance')
    call hip_check(sgm_solver_solve(solver%handle, hA, x, b, hpc, SGM_HOST), 'sgm_solver_solve')
    call hip_check(sgm_solver_info(solver%handle, its, res2, conv, last), 'sgm_solver_info')
    solver%iterations = int(its)        ! accumulates across solves like cg_solvers.f90:145

end subroutine run


subroutine hip_krylov_destroy(solver)
    class(hip_krylov_solver), intent(inout) :: solver
    if (c_associated(solver%handle)) call hip_check(sgm_solver_destroy(solver%handle), 'sgm_solver_destroy')
    solver%handle = c_null_ptr
    solver%initialized = .false.
end subroutine hip_krylov_destroy


!--------------------------------------------------------------------------!
subroutine hip_pc_setup(solver, A)                                         !
!--------------------------------------------------------------------------!
! jacobi_setup (jacobi_solvers.f90:37-63) / sparse_ldu_setup               !
! (ldu_solvers.f90:95-130); calling it again after the values changed      !
! re-extracts / re-factors (test/solver_test_jacobi.f90:240-274)           !
!--------------------------------------------------------------------------!
    class(hip_preconditioner), intent(inout) :: solver
    class(linear_operator), intent(in) :: A

    solver%nn = A%nrow
    call pc_handle(solver)
    call hip_check(sgm_pc_setup(solver%handle, matrix_handle(A)), 'sgm_pc_setup')
    solver%initialized = .true.

end subroutine hip_pc_setup


subroutine pc_handle(solver)
    ! jacobi() / ldu() as factories (jacobi_solvers.f90:23-31, ldu_solvers.f90:73-86): the object before any matrix
    class(hip_preconditioner), intent(inout) :: solver
    if (c_associated(solver%handle)) return
    call hip_check(sgm_pc_create(solver%handle, solver%kind), 'sgm_pc_create')
end subroutine pc_handle


subroutine hip_pc_set_option(solver, name, value)
    ! this preconditioner's own option ("ildu_strips", "ildu_rows"); before the first setup it decides which sweeps are built
    class(hip_preconditioner), intent(inout) :: solver
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    call pc_handle(solver)
    call hip_check(sgm_pc_set_option(solver%handle, trim(name) // c_null_char, value), 'sgm_pc_set_option')
end subroutine hip_pc_set_option


subroutine hip_pc_info(solver, levels, path, colours, est_us, name, part)
    ! after setup: which sweeps serve the applies and what one costs (sgm_pc_info) -- ldu() of a naturally ordered grid
    ! (solver_test_incomplete_cholesky.f90:137-141) answers "strip pipeline, <nx + ny - 1> levels": a dependency chain;
    ! with solver%set_option("ildu_reorder", 1) before setup it answers "row space, 2 levels"
    class(hip_preconditioner), intent(inout) :: solver
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
    call pc_handle(solver)
    call hip_check(sgm_pc_info(solver%handle, int(ip, c_int32_t), o, us, buf, 160_c_int), 'sgm_pc_info')
    levels = int(o(1:2)); path = int(o(3)); colours = int(o(4)); est_us = real(us, dp)
    name = ' '
    do k = 1, min(len(name), 160)
        if (buf(k) == c_null_char) exit
        name(k:k) = buf(k)
    enddo
end subroutine hip_pc_info


!--------------------------------------------------------------------------!
subroutine hip_pc_solve(solver, A, x, b)                                   !
!--------------------------------------------------------------------------!
! jacobi_solve (jacobi_solvers.f90:68-81) / ldu_solve                      !
! (ldu_solvers.f90:160-176): x = M^-1 b                                    !
!--------------------------------------------------------------------------!
    class(hip_preconditioner), intent(inout) :: solver
    class(linear_operator), intent(in) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)

    call hip_check(sgm_pc_apply(solver%handle, b, x, SGM_HOST), 'sgm_pc_apply')

end subroutine hip_pc_solve


subroutine hip_pc_destroy(solver)
    class(hip_preconditioner), intent(inout) :: solver
    if (c_associated(solver%handle)) call hip_check(sgm_pc_destroy(solver%handle), 'sgm_pc_destroy')
    solver%handle = c_null_ptr
    solver%initialized = .false.
end subroutine hip_pc_destroy


end module hip_solvers




!==========================================================================!
module hip_eigensolver                                                     !
!==========================================================================!
! lanczos / generalized_lanczos of src/eigensolver.f90 with the recurrence !
! on the device: the Lanczos vectors stay in HBM for the whole process,    !
! T and Q come back once.                                                  !
!==========================================================================!

use iso_c_binding
use types, only: dp
use util, only: init_seed
use linear_operator_interface
use hip_c_abi
use hip_matrices
use hip_solvers

implicit none

contains


!--------------------------------------------------------------------------!
subroutine hip_lanczos(A, T, Q)                                            !
!--------------------------------------------------------------------------!
! lanczos(A, T, Q) (eigensolver.f90:27-90), same arguments: n = size(T,2)  !
! steps with full re-orthogonalisation, T(2,:) the diagonal, T(1,:) =      !
! T(3,:) the off-diagonal, Q(:,i) the Lanczos vectors.  The start vector   !
! is drawn like the reference's (:46-52: init_seed, random_number,         !
! 2 q - 1); the device normalises it.                                      !
!--------------------------------------------------------------------------!
    class(linear_operator), intent(in) :: A
    real(dp), intent(out) :: T(:,:), Q(:,:)
    real(dp), allocatable :: q1(:), Tc(:,:), Qc(:,:)
    integer :: n

    n = size(T, 2)
    allocate(q1(A%nrow), Tc(3, n), Qc(A%nrow, n))
    call init_seed()
    call random_number(q1)
    q1 = 2 * q1 - 1
    call hip_check(sgm_lanczos(matrix_handle(A), n, q1, Tc, Qc, SGM_HOST), 'sgm_lanczos')
    T = 0.0_dp
    Q = 0.0_dp
    T(1:3, 1:n) = Tc
    Q(1:A%nrow, 1:n) = Qc

end subroutine hip_lanczos


!--------------------------------------------------------------------------!
subroutine hip_generalized_lanczos(A, B, T, Q)                             !
!--------------------------------------------------------------------------!
! generalized_lanczos(A, B, T, Q) (eigensolver.f90:95-155): Lanczos for    !
! A x = lambda B x.  "It is assumed that `B` has a solver for it set":     !
! here a hip_* Krylov solver (B%set_solver(hip_cg(...))) and optionally a  !
! hip preconditioner (B%set_preconditioner(hip_jacobi())): every step's    !
! `call B%solve(w, v)` (:140) is the device solver, started from w = A q_i !
! like the reference's.                                                    !
!--------------------------------------------------------------------------!
    class(linear_operator), intent(in) :: A, B
    real(dp), intent(out) :: T(:,:), Q(:,:)
    real(dp), allocatable :: q1(:), Tc(:,:), Qc(:,:)
    type(c_ptr) :: hs, hp
    integer :: n

    hs = c_null_ptr
    hp = c_null_ptr
    if (associated(B%solver)) then
        select type(s => B%solver)
            class is(hip_krylov_solver)
                hs = s%handle
                if (c_associated(hs)) call hip_check(sgm_solver_set_tolerance(hs, s%tolerance), 'sgm_solver_set_tolerance')
        end select
    endif
    if (.not. c_associated(hs)) then
        print *, 'hip_generalized_lanczos: B needs a hip_cg / hip_bicgstab / hip_gmres solver (B%set_solver)'
        print *, 'Terminating.'
        call exit(1)
    endif
    if (associated(B%pc)) then
        select type(p => B%pc)
            class is(hip_preconditioner)
                hp = p%handle
        end select
    endif
    n = size(T, 2)
    allocate(q1(A%nrow), Tc(3, n), Qc(A%nrow, n))
    call init_seed()
    call random_number(q1)
    q1 = 2 * q1 - 1
    call hip_check(sgm_generalized_lanczos(matrix_handle(A), matrix_handle(B), hs, hp, n, q1, Tc, Qc, SGM_HOST), &
        & 'sgm_generalized_lanczos')
    T = 0.0_dp
    Q = 0.0_dp
    T(1:3, 1:n) = Tc
    Q(1:A%nrow, 1:n) = Qc

end subroutine hip_generalized_lanczos


end module hip_eigensolver
