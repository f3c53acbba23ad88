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
use cs_matrices
use ellpack_matrices
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
end type hip_ellpack_matrix


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

subroutine hip_ell_destroy(A)
    class(hip_ellpack_matrix), intent(inout) :: A
    if (associated(A%dev)) then
        if (c_associated(A%dev%handle)) call hip_check(sgm_mat_destroy(A%dev%handle), 'sgm_mat_destroy')
        deallocate(A%dev)
    endif
    call A%ellpack_matrix%destroy()
end subroutine hip_ell_destroy


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
    procedure :: linear_solve => hip_pc_solve
    procedure :: destroy => hip_pc_destroy
end type hip_preconditioner


contains


!--------------------------------------------------------------------------!
function matrix_handle(A) result(h)                                        !
!--------------------------------------------------------------------------!
    class(linear_operator), intent(in) :: A
    type(c_ptr) :: h

    h = c_null_ptr
    select type(A)
        class is(hip_csr_matrix)
            h = A%device_handle()
        class is(hip_ellpack_matrix)
            h = A%device_handle()
        class default
            print *, 'The hip_* solvers need a hip_csr_matrix or hip_ellpack_matrix;'
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

function hip_ldu() result(pc)       ! ldu(incomplete = .true., level = 0), ldu_solvers.f90:73-86
    class(linear_solver), pointer :: pc
    type(hip_preconditioner), pointer :: p
    allocate(p)
    p%kind = PC_LDU
    pc => p
end function hip_ldu


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
    if (.not. c_associated(solver%handle)) then
        select case(solver%kind)
            case(KIND_CG)
                call hip_check(sgm_cg_create(solver%handle, solver%tolerance), 'sgm_cg_create')
            case(KIND_BICGSTAB)
                call hip_check(sgm_bicgstab_create(solver%handle, solver%tolerance), 'sgm_bicgstab_create')
            case default
                call hip_check(sgm_gmres_create(solver%handle, solver%tolerance, solver%restart), 'sgm_gmres_create')
        end select
    endif
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
    if (c_associated(solver%handle)) then
        call hip_check(sgm_pc_setup(solver%handle, matrix_handle(A)), 'sgm_pc_setup')
    elseif (solver%kind == PC_JACOBI) then
        call hip_check(sgm_jacobi_create(solver%handle, matrix_handle(A)), 'sgm_jacobi_create')
    else
        call hip_check(sgm_ildu0_create(solver%handle, matrix_handle(A)), 'sgm_ildu0_create')
    endif
    solver%initialized = .true.

end subroutine hip_pc_setup


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
