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
! This file does NOT depend on the reference's modules, so that it builds  !
! on the GPU box; INTEGRATION.md shows the ~40-line variant that extends   !
! the reference's own csr_matrix / linear_solver types instead.            !
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
end interface


!--------------------------------------------------------------------------!
type :: hip_csr_matrix                                                     !
!--------------------------------------------------------------------------!
! Same host data as csr_matrix (cs_matrices.f90:32-38,112): ptr/node of    !
! the cs_graph, val of the matrix.                                         !
!--------------------------------------------------------------------------!
    integer :: nrow = 0, ncol = 0
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:)
    type(c_ptr) :: handle = c_null_ptr
    logical :: values_dirty = .true.
contains
    procedure :: init => hip_csr_init
    procedure :: get_value => hip_csr_get_value
    procedure :: set_value => hip_csr_set_value
    procedure :: add_value => hip_csr_add_value
    procedure :: zero => hip_csr_zero
    procedure :: upload => hip_csr_upload
    procedure :: matvec => hip_csr_matvec
    procedure :: matvec_add => hip_csr_matvec_add
    procedure :: matvec_t => hip_csr_matvec_t
    procedure :: matvec_t_add => hip_csr_matvec_t_add
    procedure :: left_permute => hip_csr_left_permute
    procedure :: right_permute => hip_csr_right_permute
    procedure :: destroy => hip_csr_destroy
end type hip_csr_matrix


!--------------------------------------------------------------------------!
type :: hip_ellpack_matrix                                                 !
!--------------------------------------------------------------------------!
! Same host data as ellpack_matrix (ellpack_matrices.f90:28-33):           !
! node(max_d,n), degrees(n), val(max_d,n); padding slots repeat the last   !
! neighbour (ellpack_graphs.f90:164).                                      !
!--------------------------------------------------------------------------!
    integer :: nrow = 0, ncol = 0, max_d = 0
    integer, allocatable :: node(:,:), degrees(:)
    real(dp), allocatable :: val(:,:)
    type(c_ptr) :: handle = c_null_ptr
    logical :: values_dirty = .true.
contains
    procedure :: init => hip_ell_init
    procedure :: add_edge => hip_ell_add_edge
    procedure :: set_value => hip_ell_set_value
    procedure :: zero => hip_ell_zero
    procedure :: upload => hip_ell_upload
    procedure :: matvec => hip_ell_matvec
    procedure :: matvec_add => hip_ell_matvec_add
    procedure :: destroy => hip_ell_destroy
end type hip_ellpack_matrix


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
    procedure :: setup_csr => hip_solver_setup_csr
    procedure :: setup_ell => hip_solver_setup_ell
    generic :: setup => setup_csr, setup_ell
    procedure :: solve_csr => hip_solver_solve_csr
    procedure :: solve_csr_pc => hip_solver_solve_csr_pc
    procedure :: solve_ell => hip_solver_solve_ell
    procedure :: solve_ell_pc => hip_solver_solve_ell_pc
    generic :: solve => solve_csr, solve_csr_pc, solve_ell, solve_ell_pc
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

subroutine hip_csr_matvec(A, x, y)      ! linear_operator_interface.f90:185-194
    class(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_csr_matvec_add(A, x, y)  ! cs_matrices.f90:600-622
    class(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_add(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_csr_matvec_t(A, x, y)    ! linear_operator_interface.f90:199-208
    class(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_t(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_csr_matvec_t_add(A, x, y)  ! csc_matvec_add, cs_matrices.f90:627-647
    class(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_t_add(A%handle, x, y, SGM_HOST))
end subroutine

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

subroutine hip_ell_matvec(A, x, y)
    class(hip_ellpack_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(out) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec(A%handle, x, y, SGM_HOST))
end subroutine

subroutine hip_ell_matvec_add(A, x, y)             ! ellpack_matrices.f90:640-665
    class(hip_ellpack_matrix), intent(inout) :: A
    real(dp), intent(in) :: x(:)
    real(dp), intent(inout) :: y(:)
    call A%upload()
    call hip_check(sgm_mat_matvec_add(A%handle, x, y, SGM_HOST))
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

function hip_ldu(incomplete, level) result(s)      ! ldu_solvers.f90:73-86
    logical, intent(in), optional :: incomplete
    integer, intent(in), optional :: level
    type(hip_linear_solver), pointer :: s
    allocate(s)
    s%kind = 12       ! like ldu_set_params :143-151: always ILDU(0)
end function

subroutine hip_solver_setup_handle(s, Ah, nrow)
    class(hip_linear_solver), intent(inout) :: s
    type(c_ptr), intent(in) :: Ah
    integer, intent(in) :: nrow
    select case (s%kind)
    case (1)
        if (.not. c_associated(s%handle)) &
            & call hip_check(sgm_cg_create(s%handle, s%tolerance))
        call hip_check(sgm_solver_setup(s%handle, Ah))
    case (2)
        if (.not. c_associated(s%handle)) &
            & call hip_check(sgm_bicgstab_create(s%handle, s%tolerance))
        call hip_check(sgm_solver_setup(s%handle, Ah))
    case (3)
        if (.not. c_associated(s%handle)) &
            & call hip_check(sgm_gmres_create(s%handle, s%tolerance, &
            & int(s%restart, c_int32_t)))
        call hip_check(sgm_solver_setup(s%handle, Ah))
    case (11)
        if (.not. c_associated(s%handle)) then
            call hip_check(sgm_jacobi_create(s%handle, Ah))
        else
            call hip_check(sgm_pc_setup(s%handle, Ah))
        endif
    case (12)
        if (.not. c_associated(s%handle)) then
            call hip_check(sgm_ildu0_create(s%handle, Ah))
        else
            call hip_check(sgm_pc_setup(s%handle, Ah))
        endif
    end select
    s%nn = nrow
    s%iterations = 0                    ! cg_solvers.f90:72
    s%initialized = .true.
end subroutine

subroutine hip_solver_setup_csr(s, A)
    class(hip_linear_solver), intent(inout) :: s
    type(hip_csr_matrix), intent(inout) :: A
    call A%upload()
    call hip_solver_setup_handle(s, A%handle, A%nrow)
end subroutine

subroutine hip_solver_setup_ell(s, A)
    class(hip_linear_solver), intent(inout) :: s
    type(hip_ellpack_matrix), intent(inout) :: A
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
    if (s%kind > 10) then
        ! a preconditioner used as a solver: pc%solve(A, x, b)  (jacobi_solve / ldu_solve)
        call hip_check(sgm_pc_apply(s%handle, b, x, SGM_HOST))
        return
    endif
    call hip_check(sgm_solver_solve(s%handle, Ah, x, b, pch, SGM_HOST))
    call hip_check(sgm_solver_info(s%handle, its, res2, conv, last))
    s%iterations = int(its)
end subroutine

subroutine hip_solver_solve_csr(s, A, x, b)
    class(hip_linear_solver), intent(inout) :: s
    type(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    call A%upload()
    call hip_solver_solve_handle(s, A%handle, x, b, c_null_ptr)
end subroutine

subroutine hip_solver_solve_csr_pc(s, A, x, b, pc)
    class(hip_linear_solver), intent(inout) :: s
    type(hip_csr_matrix), intent(inout) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    type(hip_linear_solver), intent(inout) :: pc
    call A%upload()
    call hip_solver_solve_handle(s, A%handle, x, b, pc%handle)
end subroutine

subroutine hip_solver_solve_ell(s, A, x, b)
    class(hip_linear_solver), intent(inout) :: s
    type(hip_ellpack_matrix), intent(inout) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    call A%upload()
    call hip_solver_solve_handle(s, A%handle, x, b, c_null_ptr)
end subroutine

subroutine hip_solver_solve_ell_pc(s, A, x, b, pc)
    class(hip_linear_solver), intent(inout) :: s
    type(hip_ellpack_matrix), intent(inout) :: A
    real(dp), intent(inout) :: x(:)
    real(dp), intent(in) :: b(:)
    type(hip_linear_solver), intent(inout) :: pc
    call A%upload()
    call hip_solver_solve_handle(s, A%handle, x, b, pc%handle)
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

end module sigma_hip
