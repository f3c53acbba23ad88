!==========================================================================!
! hip_binding_test -- TEST INFRASTRUCTURE.  The reference's own two        !
! deterministic solver tests (test/solver_test_diffusion_1d.f90:50-120,    !
! test/solver_test_advection_diffusion_1d.f90:55-122: same problem, same   !
! thresholds) run through oracle/hip_binding.f90, i.e. with the reference's!
! graph / matrix machinery on the host and every product and solve on the  !
! GPU behind the reference's own types:                                    !
!   1. the REFERENCE's cg() loop on a hip_ellpack_matrix   (operator seam: !
!      only A%matvec is on the device, cg_solvers.f90:116-150 runs as is)  !
!   2. hip_cg() on the same matrix                   (solver seam)         !
!   3. A%set_solver / A%set_preconditioner / A%solve  (facade,             !
!      linear_operator_interface.f90:213-280) with hip_cg + hip_jacobi     !
!   4. hip_bicgstab() (+ hip_ldu()) on the nonsymmetric CSR problem        !
!   5. a value edit after setup (test/solver_test_jacobi.f90:240-274)      !
! Without a GPU the first product ends the program with the reference's    !
! error behaviour: message + exit(1).                                      !
!==========================================================================!
program hip_binding_test

use types, only: dp
use graphs
use sparse_matrices
use linear_operator_interface
use cg_solvers
use hip_matrices
use hip_solvers

implicit none

    class(graph_interface), pointer :: g, h
    type(hip_ellpack_matrix) :: A
    type(hip_csr_matrix) :: B
    class(linear_solver), pointer :: solver, pc
    real(dp), allocatable :: u(:), v(:), f(:), y(:), yr(:)
    type(ellpack_matrix) :: Ar
    real(dp) :: dx, misfit, c
    integer :: i, nn, its_ref, its_hip

    !----------------------------------------------------------------------!
    ! - d^2/dx^2, ELLPACK, n = 127 (solver_test_diffusion_1d.f90:55-78)     !
    !----------------------------------------------------------------------!
    nn = 127
    dx = 1.0_dp / (nn + 1)
    allocate(ll_graph :: g)
    call g%init(nn, nn)
    do i = 1, nn - 1
        call g%add_edge(i, i)
        call g%add_edge(i, i + 1)
        call g%add_edge(i + 1, i)
    enddo
    call g%add_edge(nn, nn)
    call convert_graph_type(g, "ellpack")

    call A%init(nn, nn)
    call A%set_graph(g)
    call A%zero()
    call Ar%init(nn, nn)
    call Ar%set_graph(g)
    call Ar%zero()
    do i = 1, nn - 1
        call A%set_value(i, i,     +2.0_dp)
        call A%set_value(i, i + 1, -1.0_dp)
        call A%set_value(i + 1, i, -1.0_dp)
        call Ar%set_value(i, i,     +2.0_dp)
        call Ar%set_value(i, i + 1, -1.0_dp)
        call Ar%set_value(i + 1, i, -1.0_dp)
    enddo
    call A%set_value(nn, nn, 2.0_dp)
    call Ar%set_value(nn, nn, 2.0_dp)

    allocate(u(nn), v(nn), f(nn), y(nn), yr(nn))
    f = 2.0 * dx**2
    do i = 1, nn
        v(i) = i * dx * (1.0_dp - i * dx)
    enddo

    ! A%matvec (linear_operator_matvec -> hip matvec_add) == the reference's, bit for bit
    call A%matvec(v, y)
    call Ar%matvec(v, yr)
    if (any(y /= yr)) then
        print *, 'hip matvec differs from ellpack_matvec_add'
        call exit(1)
    endif
    print *, 'matvec through hip_ellpack_matrix: bit-identical to the reference'

    ! 1. the reference's own CG loop, products on the device
    u = 0.0_dp
    solver => cg(1.d-16)
    call solver%setup(A)
    call solver%solve(A, u, f)
    misfit = maxval(dabs(u - v))
    select type(solver)
        type is(cg_solver)
            its_ref = solver%iterations
    end select
    print *, 'reference cg() on hip matrix: iterations', its_ref, ' error', misfit
    if (misfit > 1.0e-14) then
        print *, 'CG solver failed.'
        call exit(1)
    endif
    call solver%destroy()
    deallocate(solver)

    ! 2. the device-resident loop
    u = 0.0_dp
    solver => hip_cg(1.d-16)
    call solver%setup(A)
    call solver%solve(A, u, f)
    misfit = maxval(dabs(u - v))
    select type(solver)
        type is(hip_krylov_solver)
            its_hip = solver%iterations
    end select
    print *, 'hip_cg(): iterations', its_hip, ' error', misfit
    if (misfit > 1.0e-14 .or. abs(its_hip - its_ref) > 1) then
        print *, 'hip CG solver failed.'
        call exit(1)
    endif

    ! 3. the A%solve facade with a Jacobi preconditioner
    pc => hip_jacobi()
    call A%set_solver(solver)
    call A%set_preconditioner(pc)
    u = 0.0_dp
    call A%solve(u, f)
    misfit = maxval(dabs(u - v))
    print *, 'A%solve (hip_cg + hip_jacobi): error', misfit
    if (misfit > 1.0e-14) then
        print *, 'preconditioned hip CG solver failed.'
        call exit(1)
    endif

    ! 5. edit the matrix after setup: A <- 2 A, so u <- u / 2
    call A%scalar_multiply(2.0_dp)
    call pc%setup(A)
    u = 0.0_dp
    call A%solve(u, f)
    misfit = maxval(dabs(2.0_dp * u - v))
    print *, 'after scalar_multiply(2): error', misfit
    if (misfit > 1.0e-14) then
        print *, 'solve after a value update failed.'
        call exit(1)
    endif
    call solver%destroy()
    call pc%destroy()
    deallocate(solver, pc)

    !----------------------------------------------------------------------!
    ! 4. - d^2/dx^2 + c d/dx, CSR, n = 1024                                 !
    !    (solver_test_advection_diffusion_1d.f90:58-122)                    !
    !----------------------------------------------------------------------!
    deallocate(u, v, f)
    nn = 1024
    dx = 1.0_dp / (nn + 1)
    c = 0.5_dp
    allocate(ll_graph :: h)
    call h%init(nn, nn)
    do i = 1, nn - 1
        call h%add_edge(i, i)
        call h%add_edge(i, i + 1)
        call h%add_edge(i + 1, i)
    enddo
    call h%add_edge(nn, nn)
    call convert_graph_type(h, "compressed sparse")
    call B%init(nn, nn)
    call B%set_graph(h)
    call B%zero()
    do i = 1, nn - 1
        call B%set_value(i, i, 2.0_dp)
        call B%set_value(i, i + 1, -1.0_dp + c * dx / 2)
        call B%set_value(i + 1, i, -1.0_dp - c * dx / 2)
    enddo
    call B%set_value(nn, nn, 2.0_dp)
    allocate(u(nn), v(nn), f(nn))
    f = 2.0 * dx**2
    do i = 1, nn
        v(i) = 2 * (i * dx - (dexp(c * i * dx) - 1) / (dexp(c) - 1)) / c
    enddo

    u = 0.0_dp
    solver => hip_bicgstab(1.d-12)
    call solver%setup(B)
    call solver%solve(B, u, f)
    misfit = maxval(dabs(u - v))
    select type(solver)
        type is(hip_krylov_solver)
            print *, 'hip_bicgstab(): iterations', solver%iterations, ' error', misfit
    end select
    if (misfit > 1.0e-8) then
        print *, 'BiCG-Stab solver failed.'
        call exit(1)
    endif
    pc => hip_ldu()
    call pc%setup(B)
    u = 0.0_dp
    call solver%solve(B, u, f, pc)
    misfit = maxval(dabs(u - v))
    print *, 'hip_bicgstab() + hip_ldu(): error', misfit
    if (misfit > 1.0e-8) then
        print *, 'preconditioned BiCG-Stab solver failed.'
        call exit(1)
    endif
    call solver%destroy()
    call pc%destroy()
    deallocate(solver, pc)
    call A%destroy()
    call B%destroy()

    print *, 'hip_binding_test: all passed'

end program hip_binding_test
