!--------------------------------------------------------------------------!
! The reference's test/solver_test_diffusion_1d.f90 (ELLPACK n=127, CG     !
! tol 1e-16, pass if max error <= 1e-14) and                               !
! test/solver_test_advection_diffusion_1d.f90 (ELLPACK n=1024, BiCGStab    !
! tol 1e-12, pass if max error <= 1e-8), re-written against the sigma_hip  !
! host layer: same problem set-up statements, same solver calls, same      !
! pass/fail thresholds and exit codes; the arithmetic runs in HIP kernels. !
! A third block runs Jacobi-PCG on the CSR form of the first matrix.       !
!--------------------------------------------------------------------------!
program solver_test_diffusion_1d_hip

use sigma_hip

implicit none

    type(hip_ellpack_matrix) :: A
    type(hip_csr_matrix) :: B
    type(hip_linear_solver), pointer :: solver, pc
    real(dp), allocatable :: u(:), v(:), f(:)
    integer, allocatable :: ptr(:), node(:), pp(:), cptrs(:)
    real(dp), allocatable :: fp(:), up(:)
    integer :: i, nn, k, ncol
    real(dp) :: dx, misfit, c, x
    character(len=16) :: arg
    logical :: verbose

    verbose = .false.
    call getarg(1, arg)
    if (trim(arg) == '-v' .or. trim(arg) == '--verbose') verbose = .true.

    !------------------------------------------------------------------!
    ! - d^2/dx^2, ELLPACK, CG(1e-16)        (solver_test_diffusion_1d)  !
    !------------------------------------------------------------------!
    nn = 127
    dx = 1.0_dp / (nn + 1)

    call A%init(nn, nn, 3)
    do i = 1, nn - 1
        call A%add_edge(i, i)
        call A%add_edge(i, i + 1)
        call A%add_edge(i + 1, i)
    enddo
    call A%add_edge(nn, nn)
    call A%zero()
    do i = 1, nn - 1
        call A%set_value(i, i,     +2.0_dp)
        call A%set_value(i, i + 1, -1.0_dp)
        call A%set_value(i + 1, i, -1.0_dp)
    enddo
    call A%set_value(nn, nn, 2.0_dp)

    allocate(u(nn), v(nn), f(nn))
    u = 0.0_dp
    f = 2.0 * dx**2
    do i = 1, nn
        v(i) = i * dx * (1.0_dp - i * dx)
    enddo

    solver => hip_cg(1.d-16)
    call solver%setup(A)
    call solver%solve(A, u, f)

    misfit = maxval(dabs(u - v))
    if (verbose) print *, 'CG iterations:', solver%iterations, ' error:', misfit
    if (misfit > 1.0e-14) then
        print *, 'CG solver failed.'
        print *, 'Should have error <', 1.0e-14
        print *, 'Error found:', misfit
        call exit(1)
    endif
    if (solver%iterations /= 64) then
        print *, 'CG took', solver%iterations, 'iterations; the reference takes 64'
        call exit(1)
    endif
    call solver%destroy()
    deallocate(solver)

    ! the same solve as a loop of kernel launches instead of one workgroup
    ! (library option "cg_small"): same count, same error bound
    call hip_set_option("cg_small", 0)
    u = 0.0_dp
    solver => hip_cg(1.d-16)
    call solver%setup(A)
    call solver%solve(A, u, f)
    call hip_set_option("cg_small", 1)
    misfit = maxval(dabs(u - v))
    if (verbose) print *, 'CG (launch loop) iterations:', solver%iterations, ' error:', misfit
    if (misfit > 1.0e-14 .or. solver%iterations /= 64) then
        print *, 'CG as a launch loop failed:', solver%iterations, misfit
        call exit(1)
    endif
    call solver%destroy()
    deallocate(solver)

    !------------------------------------------------------------------!
    ! same matrix in CSR + Jacobi-preconditioned CG                     !
    !------------------------------------------------------------------!
    allocate(ptr(nn + 1), node(3 * nn - 2))
    k = 1
    do i = 1, nn
        ptr(i) = k
        if (i > 1) then
            node(k) = i - 1
            k = k + 1
        endif
        node(k) = i
        k = k + 1
        if (i < nn) then
            node(k) = i + 1
            k = k + 1
        endif
    enddo
    ptr(nn + 1) = k
    call B%init(nn, nn, ptr, node)
    do i = 1, nn - 1
        call B%set_value(i, i,     +2.0_dp)
        call B%set_value(i, i + 1, -1.0_dp)
        call B%set_value(i + 1, i, -1.0_dp)
    enddo
    call B%set_value(nn, nn, 2.0_dp)

    ! B is symmetric: B^T f and B f agree to rounding (different summation order per entry)
    call B%matvec(f, u)
    call B%matvec_t(f, v)
    if (maxval(dabs(u - v)) > 1.0d-18) then
        print *, 'matvec_t differs from matvec on a symmetric matrix:', maxval(dabs(u - v))
        call exit(1)
    endif
    do i = 1, nn
        v(i) = i * dx * (1.0_dp - i * dx)
    enddo

    solver => hip_cg(1.d-16)
    pc => hip_jacobi()
    call solver%setup(B)
    call pc%setup(B)
    u = 0.0_dp
    call solver%solve(B, u, f, pc)
    misfit = maxval(dabs(u - v))
    if (verbose) print *, 'Jacobi-PCG iterations:', solver%iterations, ' error:', misfit
    if (misfit > 1.0e-14) then
        print *, 'Jacobi-preconditioned CG failed. Error found:', misfit
        call exit(1)
    endif
    call solver%destroy()
    call pc%destroy()
    deallocate(solver, pc)

    !------------------------------------------------------------------!
    ! the same system after greedy_color_ordering (permutations.f90)    !
    ! and a symmetric permutation: ILDU(0)-preconditioned CG             !
    !------------------------------------------------------------------!
    allocate(pp(nn), cptrs(nn + 2), fp(nn), up(nn))
    call hip_greedy_color_ordering(pp, cptrs, ncol, B)
    if (ncol /= 2 .or. cptrs(1) /= 1 .or. cptrs(3) /= nn + 1) then
        print *, 'a tridiagonal graph takes two colours; got', ncol, cptrs(1:3)
        call exit(1)
    endif
    call B%left_permute(pp)
    call B%right_permute(pp)
    do i = 1, nn
        fp(pp(i)) = f(i)
        if (B%get_value(pp(i), pp(i)) /= 2.0_dp) then
            print *, 'permuted diagonal entry is wrong at', i
            call exit(1)
        endif
    enddo
    solver => hip_cg(1.d-16)
    pc => hip_ldu(incomplete = .true., level = 0)
    call solver%setup(B)
    call pc%setup(B)
    up = 0.0_dp
    call solver%solve(B, up, fp, pc)
    do i = 1, nn
        u(i) = up(pp(i))
    enddo
    misfit = maxval(dabs(u - v))
    if (verbose) print *, 'colour-ordered ILDU-PCG iterations:', solver%iterations, ' error:', misfit
    if (misfit > 1.0e-14) then
        print *, 'ILDU-preconditioned CG on the colour-ordered matrix failed. Error found:', misfit
        call exit(1)
    endif
    call solver%destroy()
    call pc%destroy()
    deallocate(solver, pc, pp, cptrs, fp, up)

    call B%destroy()
    call A%destroy()
    deallocate(u, v, f)

    !------------------------------------------------------------------!
    ! - d^2/dx^2 + c d/dx, ELLPACK, BiCGStab(1e-12)                     !
    !                              (solver_test_advection_diffusion_1d) !
    !------------------------------------------------------------------!
    nn = 1024
    dx = 1.0_dp / (nn + 1)
    c = 0.5_dp

    call A%init(nn, nn, 3)
    do i = 1, nn - 1
        call A%add_edge(i, i)
        call A%add_edge(i, i + 1)
        call A%add_edge(i + 1, i)
    enddo
    call A%add_edge(nn, nn)
    call A%zero()
    do i = 1, nn - 1
        call A%set_value(i, i,     +2.0_dp)
        call A%set_value(i, i + 1, -1.0_dp + c * dx/2)
        call A%set_value(i + 1, i, -1.0_dp - c * dx/2)
    enddo
    call A%set_value(nn, nn, 2.0_dp)

    allocate(u(nn), v(nn), f(nn))
    u = 0.0_dp
    f = 2.0_dp * dx**2
    do i = 1, nn
        x = i * dx
        v(i) = 2.0_dp * ( x - (exp(c * x) - 1) / (exp(c) - 1) )/c
    enddo

    solver => hip_bicgstab(1.0d-12)
    call solver%setup(A)
    call solver%solve(A, u, f)

    misfit = maxval(dabs(u - v))
    if (verbose) print *, 'BiCGStab iterations:', solver%iterations, ' error:', misfit
    if (misfit > 1.0e-8) then
        print *, 'BiCG-Stab solver failed.'
        print *, 'Should have error <', 1.0e-8
        print *, 'Error found:', misfit
        call exit(1)
    endif

    call A%destroy()
    call solver%destroy()
    deallocate(solver)

    if (verbose) print *, 'all sigma_hip Fortran checks passed'

end program solver_test_diffusion_1d_hip
