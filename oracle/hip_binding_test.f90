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
! and, in the contained subroutines (each the flow of one more of the      !
! reference's tests, hip types, the reference's thresholds; the random     !
! inputs come from a FIXED seed instead of util.f90's time seed):          !
!   6. test/solver_test_jacobi.f90:62-296: random graph Laplacian + I,     !
!      Jacobi as a stationary solver, Jacobi-PCG, skew perturbation,       !
!      re-setup, Jacobi-PBiCGStab                                          !
!   7. test/solver_test_incomplete_cholesky.f90:62-226: ILDU(0) as a       !
!      stationary solver and as the preconditioner of CG                   !
!   8. test/matrix_test_basics.f90:333-362: matvec / matvec_t against the  !
!      dense product (hip_csr_matrix and hip_ellpack_matrix)               !
!   9. hip_csr_from_edges == the host assembly sequence, array for array   !
!  10. the composite: a 2 x 2 hip_sparse_matrix of hip leaves -- the        !
!      reference's block loop (sparse_matrix_composites.f90:1076-1099)     !
!      over device leaves, the one-handle device product, and hip_cg on    !
!      the composite, against the reference's own composite of csr leaves  !
!  11. test/eigensolver_test_lanczos.f90:98-165 through hip_lanczos        !
!  12. test/eigensolver_test_generalized_lanczos.f90:60-200 through        !
!      hip_generalized_lanczos, B%set_solver(hip_cg(1d-15))                !
! Without a GPU the first product ends the program with the reference's    !
! error behaviour: message + exit(1).                                      !
!==========================================================================!
program hip_binding_test

use iso_c_binding
use types, only: dp
use graphs
use sparse_matrices
use linear_operator_interface
use cg_solvers
use ldu_solvers
use hip_matrices
use hip_solvers
use hip_eigensolver

implicit none

    class(graph_interface), pointer :: g, h
    type(hip_ellpack_matrix) :: A
    type(hip_csr_matrix) :: B
    class(linear_solver), pointer :: solver, pc, rsolver, rpc
    real(dp), allocatable :: u(:), v(:), f(:), y(:), yr(:), ur(:)
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

    ! 2b. solver%tolerance is a live public field (cg_solvers.f90:17,133) and set_params may be called again (:95-111):
    !     the same handle, first with a tolerance the initial residual already meets (no iteration, u untouched), then the
    !     field edited directly, then through set_params -- `iterations` accumulates (cg_solvers.f90:72,145)
    select type(solver)
        type is(hip_krylov_solver)
            u = 0.0_dp
            solver%tolerance = 1.0_dp
            call solver%solve(A, u, f)
            if (solver%iterations /= its_hip .or. any(u /= 0.0_dp)) then
                print *, 'a tolerance of 1 edited on the live solver was ignored: iterations', solver%iterations
                call exit(1)
            endif
            solver%tolerance = 1.d-16
            call solver%solve(A, u, f)
            misfit = maxval(dabs(u - v))
            print *, 'hip_cg() after editing solver%tolerance: iterations', solver%iterations, ' error', misfit
            if (misfit > 1.0e-14 .or. abs(solver%iterations - 2 * its_hip) > 1) then
                print *, 'the edited tolerance did not reach the device loop.'
                call exit(1)
            endif
            u = 0.0_dp
            call solver%set_params(1.0_dp)
            call solver%solve(A, u, f)
            if (any(u /= 0.0_dp)) then
                print *, 'set_params(1) on the live solver was ignored'
                call exit(1)
            endif
            call solver%set_params()
            if (solver%tolerance /= 1.d-16) call exit(1)
    end select

    ! 2c. ldu() on the ELLPACK operand (sparse_ldu_setup takes any sparse_matrix_interface, ldu_solvers.f90:95-130, and
    !     reads it through the edge cursor: real entries, not padding): hip_cg + hip_ldu on A against the reference's
    !     cg + ldu on its own ellpack_matrix
    allocate(ur(nn))
    rsolver => cg(1.d-16)
    rpc => ldu(incomplete = .true., level = 0)
    call rsolver%setup(Ar)
    call rpc%setup(Ar)
    ur = 0.0_dp
    call rsolver%solve(Ar, ur, f, rpc)
    pc => hip_ldu()
    call pc%setup(A)
    call solver%setup(A)
    u = 0.0_dp
    call solver%solve(A, u, f, pc)
    misfit = maxval(dabs(u - ur))
    select type(solver)
        type is(hip_krylov_solver)
            select type(rsolver)
                type is(cg_solver)
                    print *, 'hip_cg() + hip_ldu() on ELLPACK: iterations', solver%iterations, ' reference', &
                        & rsolver%iterations, ' difference', misfit
                    if (misfit > 1.0e-14 .or. abs(solver%iterations - rsolver%iterations) > 1) then
                        print *, 'hip_ldu() on an ELLPACK matrix failed.'
                        call exit(1)
                    endif
            end select
    end select
    call pc%destroy()
    call rsolver%destroy()
    call rpc%destroy()
    deallocate(pc, rsolver, rpc, ur)

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

    call test_jacobi_flow()
    call test_incomplete_cholesky_flow()
    call test_matvec_against_dense()
    call test_from_edges()
    call test_composite()
    call test_lanczos()
    call test_generalized_lanczos()

    print *, 'hip_binding_test: all passed'


contains


!--------------------------------------------------------------------------!
subroutine fixed_seed(k)                                                   !
!--------------------------------------------------------------------------!
! the reference's tests seed from the clock (util.f90:72-102); a test that !
! is re-run by a harness wants the same inputs every time                  !
!--------------------------------------------------------------------------!
    integer, intent(in) :: k
    integer :: n, i
    integer, allocatable :: seed(:)
    call random_seed(size=n)
    allocate(seed(n))
    do i = 1, n
        seed(i) = 1234567 + 7919 * i + 104729 * k
    enddo
    call random_seed(put=seed)
end subroutine fixed_seed


subroutine fail(msg, val)
    character(len=*), intent(in) :: msg
    real(dp), intent(in) :: val
    print *, msg
    print *, 'Error:', val
    call exit(1)
end subroutine fail


!--------------------------------------------------------------------------!
subroutine random_spd_matrix(M, gr, nn)                                    !
!--------------------------------------------------------------------------!
! test/solver_test_jacobi.f90:62-128: Erdos-Renyi graph, Laplacian with    !
! random weights + I                                                       !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix), intent(inout) :: M
    class(graph_interface), pointer, intent(out) :: gr
    integer, intent(in) :: nn
    integer :: i, j, k, d
    integer, allocatable :: nodes(:)
    real(dp) :: z, p

    p = log(1.0_dp * nn) / log(2.0_dp) / nn
    allocate(ll_graph :: gr)
    call gr%init(nn)
    do i = 1, nn
        call gr%add_edge(i, i)
        do j = i + 1, nn
            call random_number(z)
            if (z < p) then
                call gr%add_edge(i, j)
                call gr%add_edge(j, i)
            endif
        enddo
    enddo
    d = gr%get_max_degree()
    allocate(nodes(d))
    call convert_graph_type(gr, "compressed sparse")
    call M%init(nn, nn)
    call M%set_graph(gr)
    call M%zero()
    do i = 1, nn
        call M%add_value(i, i, 1.0_dp)
        call gr%get_neighbors(nodes, i)
        d = gr%get_degree(i)
        do k = 1, d
            j = nodes(k)
            call random_number(z)
            if (j > i) then
                call M%set_value(i, j, -z)
                call M%add_value(i, i, +z)
                call M%set_value(j, i, -z)
                call M%add_value(j, j, +z)
            endif
        enddo
    enddo
end subroutine random_spd_matrix


!--------------------------------------------------------------------------!
subroutine smooth_rhs(M, pcs, w, rhs)                                      !
!--------------------------------------------------------------------------!
! solver_test_jacobi.f90:160-178: w random, smoothed once by the           !
! preconditioner; rhs = M w                                                !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix), intent(inout) :: M
    class(linear_solver), intent(inout) :: pcs
    real(dp), intent(out) :: w(:), rhs(:)
    real(dp), allocatable :: r(:), q(:)
    allocate(r(size(w)), q(size(w)))
    call random_number(w)
    call M%matvec(w, q)
    r = w - q
    call pcs%solve(M, w, r)
    call M%matvec(w, rhs)
end subroutine smooth_rhs


!--------------------------------------------------------------------------!
subroutine stationary(M, pcs, rhs, x, sweeps)                              !
!--------------------------------------------------------------------------!
! solver_test_jacobi.f90:187-206: x += M_pc^-1 (rhs - M x), `sweeps` times !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix), intent(inout) :: M
    class(linear_solver), intent(inout) :: pcs
    real(dp), intent(in) :: rhs(:)
    real(dp), intent(out) :: x(:)
    integer, intent(in) :: sweeps
    real(dp), allocatable :: r(:), q(:)
    integer :: it
    allocate(r(size(x)), q(size(x)))
    x = 0.0_dp
    r = rhs
    q = 0.0_dp
    do it = 1, sweeps
        call pcs%solve(M, q, r)
        x = x + q
        call M%matvec(x, q)
        r = rhs - q
    enddo
end subroutine stationary


!--------------------------------------------------------------------------!
subroutine test_jacobi_flow()                                              !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: M
    class(graph_interface), pointer :: gr
    class(linear_solver), pointer :: ks, pcs
    real(dp), allocatable :: x(:), w(:), rhs(:)
    integer, allocatable :: nodes(:)
    integer :: nn, i, j, k, d
    real(dp) :: z, err

    nn = 128
    call fixed_seed(1)
    call random_spd_matrix(M, gr, nn)
    ks => hip_cg(1.d-16)
    pcs => hip_jacobi()
    call ks%setup(M)
    call pcs%setup(M)
    allocate(x(nn), w(nn), rhs(nn), nodes(gr%get_max_degree()))
    call smooth_rhs(M, pcs, w, rhs)

    ! the Jacobi method as a solver (:187-222)
    call stationary(M, pcs, rhs, x, 10 * nn)
    err = maxval(dabs(x - w))
    if (err > 1.0e-14) call fail('hip Jacobi method failed to produce a sufficiently accurate solution', err)
    print *, 'jacobi flow: stationary iteration error', err

    ! ... as a preconditioner (:229-246)
    x = 0.0_dp
    call ks%solve(M, x, rhs, pcs)
    err = maxval(dabs(x - w))
    if (err > 1.0e-15) call fail('Jacobi-preconditioned hip CG failed', err)
    print *, 'jacobi flow: hip_cg + hip_jacobi error', err

    ! a random skew-symmetric perturbation (:253-270), solver and preconditioner rebuilt (:277-293)
    do i = 1, nn
        call gr%get_neighbors(nodes, i)
        d = gr%get_degree(i)
        do k = 1, d
            j = nodes(k)
            if (j > i) then
                call random_number(z)
                z = (2 * z - 1) / 16
                call M%add_value(i, j, +z)
                call M%add_value(j, i, -z)
            endif
        enddo
    enddo
    call ks%destroy()
    deallocate(ks)
    ks => hip_bicgstab(1.d-16)
    call ks%setup(M)
    call pcs%setup(M)
    call M%matvec(w, rhs)
    x = 0.0_dp
    call ks%solve(M, x, rhs, pcs)
    err = maxval(dabs(x - w))
    if (err > 1.0e-15) call fail('Jacobi-preconditioned hip BiCG-Stab failed on the non-symmetric system', err)
    print *, 'jacobi flow: hip_bicgstab + hip_jacobi on the perturbed matrix, error', err
    call ks%destroy()
    call pcs%destroy()
    deallocate(ks, pcs)
    call M%destroy()
end subroutine test_jacobi_flow


!--------------------------------------------------------------------------!
subroutine test_incomplete_cholesky_flow()                                 !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: M
    class(graph_interface), pointer :: gr
    class(linear_solver), pointer :: ks, pcs
    real(dp), allocatable :: x(:), w(:), rhs(:)
    integer :: nn
    real(dp) :: err

    nn = 128
    call fixed_seed(2)
    call random_spd_matrix(M, gr, nn)
    ks => hip_cg(1.d-16)
    pcs => hip_ldu()                     ! ldu(incomplete = .true., level = 0)
    call ks%setup(M)
    call pcs%setup(M)
    allocate(x(nn), w(nn), rhs(nn))
    call smooth_rhs(M, pcs, w, rhs)
    call stationary(M, pcs, rhs, x, 10 * nn)            ! :186-205
    err = maxval(dabs(x - w))
    if (err > 1.0e-14) call fail('hip incomplete Cholesky failed as a stationary solver', err)
    print *, 'incomplete cholesky flow: stationary iteration error', err
    x = 0.0_dp
    call ks%solve(M, x, rhs, pcs)                       ! :218-226
    err = maxval(dabs(x - w))
    if (err > 1.0e-15) call fail('ILDU-preconditioned hip CG failed', err)
    print *, 'incomplete cholesky flow: hip_cg + hip_ldu error', err
    ! the same solve with the reordering preconditioner: ILDU(0) of the colour-ordered matrix (greedy_color_ordering of
    ! this random graph runs on the host: it is not bipartite), M, rhs and x untouched
    call pcs%destroy()
    deallocate(pcs)
    pcs => hip_ldu(reorder = "colour")
    call pcs%setup(M)
    block      ! which sweeps will serve the applies (sgm_pc_info): the colour-ordered factors have one level per colour
        integer :: lv(2), path, ncol
        real(dp) :: est
        character(len=80) :: what
        select type (pcs)
        class is (hip_preconditioner)
            call pcs%info(lv, path, ncol, est, what)
            print *, 'incomplete cholesky flow: hip_ldu(reorder = colour) is served by: ', trim(what), ',', ncol, 'colours'
            if (ncol < 2 .or. lv(1) /= ncol .or. lv(2) /= ncol) call fail('sgm_pc_info: levels differ from the colours of the ordering', real(lv(1), dp))
        end select
    end block
    x = 0.0_dp
    call ks%solve(M, x, rhs, pcs)
    err = maxval(dabs(x - w))
    if (err > 1.0e-15) call fail('hip CG with the reordering ILDU preconditioner failed', err)
    print *, 'incomplete cholesky flow: hip_cg + hip_ldu(reorder = colour) error', err
    call ks%destroy()
    call pcs%destroy()
    deallocate(ks, pcs)
    call M%destroy()
end subroutine test_incomplete_cholesky_flow


!--------------------------------------------------------------------------!
subroutine test_matvec_against_dense()                                     !
!--------------------------------------------------------------------------!
! matrix_test_basics.f90:333-362 on a random rectangular-free pattern:     !
! A%matvec / A%matvec_t against matmul with the dense copy, <= 1e-15       !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: Mc
    type(hip_ellpack_matrix) :: Me
    class(graph_interface), pointer :: gc, ge
    real(dp), allocatable :: D(:,:), x(:), y1(:), y2(:)
    integer :: nn, i, j
    real(dp) :: z, p, w

    nn = 64
    call fixed_seed(3)
    p = 0.1_dp
    allocate(D(nn, nn), x(nn), y1(nn), y2(nn))
    D = 0.0_dp
    allocate(ll_graph :: gc)
    allocate(ll_graph :: ge)
    call gc%init(nn, nn)
    call ge%init(nn, nn)
    do i = 1, nn
        do j = 1, nn
            call random_number(z)
            if (z < p .or. i == j) then
                call gc%add_edge(i, j)
                call ge%add_edge(i, j)
                call random_number(D(i, j))
            endif
        enddo
    enddo
    call convert_graph_type(gc, "compressed sparse")
    call convert_graph_type(ge, "ellpack")
    call Mc%init(nn, nn)
    call Mc%set_graph(gc)
    call Mc%zero()
    call Me%init(nn, nn)
    call Me%set_graph(ge)
    call Me%zero()
    do i = 1, nn
        do j = 1, nn
            if (D(i, j) /= 0) then
                call Mc%set_value(i, j, D(i, j))
                call Me%set_value(i, j, D(i, j))
            endif
        enddo
    enddo
    call random_number(x)
    y2 = matmul(D, x)
    call Mc%matvec(x, y1)
    w = maxval(dabs(y1 - y2)) / maxval(dabs(y2))
    if (w > 1.0e-15) call fail('hip_csr_matrix: matrix-vector multiplication failed', w)
    call Me%matvec(x, y1)
    w = maxval(dabs(y1 - y2)) / maxval(dabs(y2))
    if (w > 1.0e-15) call fail('hip_ellpack_matrix: matrix-vector multiplication failed', w)
    call random_number(x)
    y2 = matmul(transpose(D), x)
    call Mc%matvec_t(x, y1)
    w = maxval(dabs(y1 - y2)) / maxval(dabs(y2))
    if (w > 1.0e-15) call fail('hip_csr_matrix: transpose product failed', w)
    call Me%matvec_t(x, y1)
    w = maxval(dabs(y1 - y2)) / maxval(dabs(y2))
    if (w > 1.0e-15) call fail('hip_ellpack_matrix: transpose product failed', w)
    print *, 'matvec / matvec_t against the dense product: within 1e-15 (csr and ellpack)'
    call Mc%destroy()
    call Me%destroy()
end subroutine test_matvec_against_dense


!--------------------------------------------------------------------------!
subroutine test_from_edges()                                               !
!--------------------------------------------------------------------------!
! the edge list of the 1-D problem above, with a repeated edge and a       !
! value written twice: the device assembly and the host sequence must      !
! leave the same ptr / node / val                                          !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: Md
    type(csr_matrix) :: Mh
    class(graph_interface), pointer :: gh
    integer(c_int32_t), allocatable :: ei(:), ej(:)
    real(dp), allocatable :: ev(:), x(:), y1(:), y2(:)
    integer :: nn, i, k, ne

    nn = 200
    ne = 3 * (nn - 1) + 1 + 2
    allocate(ei(ne), ej(ne), ev(ne))
    k = 0
    do i = 1, nn - 1
        ei(k + 1) = i;     ej(k + 1) = i;     ev(k + 1) = 2.0_dp + 0.01_dp * i
        ei(k + 2) = i;     ej(k + 2) = i + 1; ev(k + 2) = -1.0_dp
        ei(k + 3) = i + 1; ej(k + 3) = i;     ev(k + 3) = -1.5_dp
        k = k + 3
    enddo
    ei(k + 1) = nn; ej(k + 1) = nn; ev(k + 1) = 2.0_dp
    ei(k + 2) = 5;  ej(k + 2) = 6;  ev(k + 2) = -7.0_dp      ! a repeated edge: ignored by add_edge, its value wins
    ei(k + 3) = 1;  ej(k + 3) = nn; ev(k + 3) = 0.25_dp      ! a far entry appended to row 1

    allocate(ll_graph :: gh)
    call gh%init(nn, nn)
    do k = 1, ne
        call gh%add_edge(ei(k), ej(k))
    enddo
    call convert_graph_type(gh, "compressed sparse")
    call Mh%init(nn, nn)
    call Mh%set_graph(gh)
    call Mh%zero()
    do k = 1, ne
        call Mh%set_value(ei(k), ej(k), ev(k))
    enddo

    call hip_csr_from_edges(Md, nn, nn, ei, ej, ev)
    if (any(Md%g%ptr /= Mh%g%ptr) .or. any(Md%g%node /= Mh%g%node) .or. any(Md%val /= Mh%val)) then
        print *, 'hip_csr_from_edges: arrays differ from the host assembly sequence'
        call exit(1)
    endif
    allocate(x(nn), y1(nn), y2(nn))
    do i = 1, nn
        x(i) = dsin(0.37_dp * i)
    enddo
    call Md%matvec(x, y1)
    call Mh%matvec(x, y2)
    if (any(y1 /= y2)) then
        print *, 'hip_csr_from_edges: product differs from the reference matrix'
        call exit(1)
    endif
    ! the host copy is a full csr_matrix: an edit through the inherited mutators reaches the device
    call Md%set_value(5, 6, -1.0_dp)
    call Mh%set_value(5, 6, -1.0_dp)
    call Md%matvec(x, y1)
    call Mh%matvec(x, y2)
    if (any(y1 /= y2)) then
        print *, 'hip_csr_from_edges: product after set_value differs'
        call exit(1)
    endif
    print *, 'hip_csr_from_edges: ptr / node / val and products identical to the host assembly'
    call Md%destroy()
    call Mh%destroy()
end subroutine test_from_edges


!--------------------------------------------------------------------------!
subroutine poisson_blocks(nx, ny, n1, C11, C12, C21, C22)                  !
!--------------------------------------------------------------------------!
! the 5-point matrix of an nx x ny grid cut into 2 x 2 blocks at row /     !
! column n1 (each block a csr_matrix-type leaf on a graph of its own)      !
!--------------------------------------------------------------------------!
    integer, intent(in) :: nx, ny, n1
    class(csr_matrix), intent(inout) :: C11, C12, C21, C22
    class(graph_interface), pointer :: g11, g12, g21, g22
    integer :: n, pass, k, i, j, l, t
    integer :: nb(5)
    real(dp) :: vb(5)

    n = nx * ny
    allocate(ll_graph :: g11)
    allocate(ll_graph :: g12)
    allocate(ll_graph :: g21)
    allocate(ll_graph :: g22)
    call g11%init(n1, n1)
    call g12%init(n1, n - n1)
    call g21%init(n - n1, n1)
    call g22%init(n - n1, n - n1)
    do pass = 1, 2
        do k = 1, n
            i = mod(k - 1, nx) + 1
            j = (k - 1) / nx + 1
            t = 0
            if (j > 1)  then; t = t + 1; nb(t) = k - nx; vb(t) = -1.0_dp; endif
            if (i > 1)  then; t = t + 1; nb(t) = k - 1;  vb(t) = -1.0_dp; endif
            t = t + 1; nb(t) = k; vb(t) = 4.0_dp
            if (i < nx) then; t = t + 1; nb(t) = k + 1;  vb(t) = -1.0_dp; endif
            if (j < ny) then; t = t + 1; nb(t) = k + nx; vb(t) = -1.0_dp; endif
            do l = 1, t
                if (k <= n1 .and. nb(l) <= n1) then
                    if (pass == 1) call g11%add_edge(k, nb(l))
                    if (pass == 2) call C11%set_value(k, nb(l), vb(l))
                elseif (k <= n1) then
                    if (pass == 1) call g12%add_edge(k, nb(l) - n1)
                    if (pass == 2) call C12%set_value(k, nb(l) - n1, vb(l))
                elseif (nb(l) <= n1) then
                    if (pass == 1) call g21%add_edge(k - n1, nb(l))
                    if (pass == 2) call C21%set_value(k - n1, nb(l), vb(l))
                else
                    if (pass == 1) call g22%add_edge(k - n1, nb(l) - n1)
                    if (pass == 2) call C22%set_value(k - n1, nb(l) - n1, vb(l))
                endif
            enddo
        enddo
        if (pass == 1) then
            call convert_graph_type(g11, "compressed sparse")
            call convert_graph_type(g12, "compressed sparse")
            call convert_graph_type(g21, "compressed sparse")
            call convert_graph_type(g22, "compressed sparse")
            call C11%init(n1, n1);         call C11%set_graph(g11); call C11%zero()
            call C12%init(n1, n - n1);     call C12%set_graph(g12); call C12%zero()
            call C21%init(n - n1, n1);     call C21%set_graph(g21); call C21%zero()
            call C22%init(n - n1, n - n1); call C22%set_graph(g22); call C22%zero()
        endif
    enddo
end subroutine poisson_blocks


!--------------------------------------------------------------------------!
subroutine test_composite()                                                !
!--------------------------------------------------------------------------!
    type(hip_sparse_matrix) :: Ch
    type(sparse_matrix) :: Cr
    type(hip_csr_matrix), target :: H11, H12, H21, H22
    type(csr_matrix), target :: R11, R12, R21, R22
    class(linear_solver), pointer :: ks
    real(dp), allocatable :: x(:), y1(:), y2(:), y3(:), rhs(:), u1(:), u2(:)
    integer :: nx, ny, n, n1, i, its_r, its_h
    real(dp) :: err

    nx = 32
    ny = 24
    n = nx * ny
    n1 = 400
    call poisson_blocks(nx, ny, n1, H11, H12, H21, H22)
    call poisson_blocks(nx, ny, n1, R11, R12, R21, R22)
    call Ch%set_dimensions(n, n)
    call Ch%set_num_blocks(2, 2)
    call Ch%set_block_sizes([n1, n - n1], [n1, n - n1])
    call Ch%set_submatrix(1, 1, H11)
    call Ch%set_submatrix(1, 2, H12)
    call Ch%set_submatrix(2, 1, H21)
    call Ch%set_submatrix(2, 2, H22)
    call Cr%set_dimensions(n, n)
    call Cr%set_num_blocks(2, 2)
    call Cr%set_block_sizes([n1, n - n1], [n1, n - n1])
    call Cr%set_submatrix(1, 1, R11)
    call Cr%set_submatrix(1, 2, R12)
    call Cr%set_submatrix(2, 1, R21)
    call Cr%set_submatrix(2, 2, R22)

    allocate(x(n), y1(n), y2(n), y3(n), rhs(n), u1(n), u2(n))
    do i = 1, n
        x(i) = dsin(0.001_dp * i)
    enddo
    ! the reference's block loop over device leaves == over csr leaves == the one-handle device product
    call Cr%matvec(x, y1)
    call Ch%matvec(x, y2)
    y3 = 0.0_dp
    call Ch%device_matvec_add(x, y3)
    if (any(y1 /= y2) .or. any(y1 /= y3)) then
        print *, 'composite over hip leaves: products differ from the reference composite'
        call exit(1)
    endif
    call Cr%matvec_t(x, y1)
    call Ch%matvec_t(x, y2)
    y3 = 0.0_dp
    call Ch%device_matvec_t_add(x, y3)
    if (any(y1 /= y2) .or. any(y1 /= y3)) then
        print *, 'composite over hip leaves: transpose products differ from the reference composite'
        call exit(1)
    endif
    print *, 'composite (2 x 2 hip leaves): block loop and one-handle product bit-identical to the reference composite'

    ! the reference's cg on its composite vs the device-resident hip_cg on the composite handle
    rhs = 1.0_dp / n
    u1 = 0.0_dp
    ks => cg(1.d-12)
    call ks%setup(Cr)
    call ks%solve(Cr, u1, rhs)
    select type(ks)
        type is(cg_solver)
            its_r = ks%iterations
    end select
    call ks%destroy()
    deallocate(ks)
    u2 = 0.0_dp
    ks => hip_cg(1.d-12)
    call ks%setup(Ch)
    call ks%solve(Ch, u2, rhs)
    select type(ks)
        type is(hip_krylov_solver)
            its_h = ks%iterations
    end select
    err = maxval(dabs(u1 - u2)) / maxval(dabs(u1))
    print *, 'composite: reference cg iterations', its_r, ' hip_cg iterations', its_h, ' relative difference', err
    if (abs(its_r - its_h) > 1 .or. err > 1.0e-12) call fail('hip_cg on the composite differs from the reference', err)
    ! a value edit in one leaf reaches both products
    call H11%set_value(1, 1, 5.0_dp)
    call R11%set_value(1, 1, 5.0_dp)
    call Cr%matvec(x, y1)
    y3 = 0.0_dp
    call Ch%device_matvec_add(x, y3)
    if (any(y1 /= y3)) then
        print *, 'composite: product after a leaf edit differs'
        call exit(1)
    endif
    call ks%destroy()
    deallocate(ks)
end subroutine test_composite


!--------------------------------------------------------------------------!
subroutine test_lanczos()                                                  !
!--------------------------------------------------------------------------!
! eigensolver_test_lanczos.f90:61-165: graph Laplacian of a random graph,  !
! three-term recurrence and orthogonality of the Lanczos vectors           !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: L
    class(graph_interface), pointer :: gr
    integer, allocatable :: nodes(:)
    real(dp), allocatable :: T(:,:), V(:,:), Q(:,:), x(:), y(:)
    integer :: nn, nq, i, j, k, d
    real(dp) :: z, p, err

    nn = 128
    p = log(1.0_dp * nn) / log(2.0_dp) / nn
    call fixed_seed(4)
    allocate(ll_graph :: gr)
    call gr%init(nn)
    do i = 1, nn
        call gr%add_edge(i, i)
        do j = i + 1, nn
            call random_number(z)
            if (z < p) then
                call gr%add_edge(i, j)
                call gr%add_edge(j, i)
            endif
        enddo
    enddo
    allocate(nodes(gr%get_max_degree()))
    call convert_graph_type(gr, "compressed sparse")
    call L%init(nn, nn)
    call L%set_graph(gr)
    call L%zero()
    do i = 1, nn
        d = gr%get_degree(i)
        call gr%get_neighbors(nodes, i)
        do k = 1, d
            j = nodes(k)
            call L%add_value(i, j, -1.0_dp)
            call L%add_value(i, i, +1.0_dp)
        enddo
    enddo
    nq = int(dsqrt(1.0_dp * nn))
    allocate(T(3, nq), V(nn, nq), Q(nq, nq), x(nn), y(nn))
    call hip_lanczos(L, T, V)
    do i = 2, nq - 1
        call L%matvec(V(:, i), x)
        y = T(2, i) * V(:, i) + T(1, i-1) * V(:, i-1) + T(3, i) * V(:, i+1)
        err = dsqrt(sum((y - x) * (y - x)) / sum(x * x))
        if (err > 1.0e-14) call fail('hip_lanczos: three-term recurrence failed', err)
    enddo
    Q = matmul(transpose(V), V)
    do i = 1, nq
        Q(i, i) = Q(i, i) - 1.0_dp
    enddo
    Q = matmul(transpose(Q), Q)
    err = 0.0_dp
    do i = 1, nq
        err = err + Q(i, i)
    enddo
    err = dsqrt(err) / nq
    if (err > 1.0e-14) call fail('hip_lanczos: Lanczos vectors are not orthogonal', err)
    print *, 'hip_lanczos: three-term recurrence and orthogonality within 1e-14; || V^t V - I ||_F / n =', err
    call L%destroy()
end subroutine test_lanczos


!--------------------------------------------------------------------------!
subroutine test_generalized_lanczos()                                      !
!--------------------------------------------------------------------------!
! eigensolver_test_generalized_lanczos.f90:60-200: P1 stiffness and mass   !
! matrices of a periodic 48 x 32 grid, B%set_solver(cg(1d-15)) ->          !
! B%set_solver(hip_cg(1d-15)); the reference prints (and does not stop) on !
! a recurrence error above 1e-14 -- here it is a failure.                  !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: S, M
    type(ll_graph) :: gr
    class(linear_solver), pointer :: ks
    real(dp), allocatable :: T(:,:), U(:,:), V(:,:), Q(:,:), w(:), z(:)
    real(dp) :: AE(3, 3), BE(3, 3), area, err
    integer :: nx, ny, nn, nq, i, j, k, l, elem(3)

    nx = 48
    ny = 32
    nn = ny * nx
    call fixed_seed(5)
    call gr%init(nn)
    do i = 1, ny
        do j = 1, nx
            k = idx(i, j)
            call gr%add_edge(k, k)
            l = idx(mod(i, ny) + 1, j)
            call gr%add_edge(k, l)
            call gr%add_edge(l, k)
            l = idx(i, mod(j, nx) + 1)
            call gr%add_edge(k, l)
            call gr%add_edge(l, k)
            l = idx(mod(i, ny) + 1, mod(j, nx) + 1)
            call gr%add_edge(k, l)
            call gr%add_edge(l, k)
        enddo
    enddo
    call S%init(nn, nn)
    call M%init(nn, nn)
    call S%copy_graph(gr)
    call M%copy_graph(gr)
    call S%zero()
    call M%zero()
    area = 0.5d0
    BE = area / 12.0_dp
    do k = 1, 3
        BE(k, k) = area / 6.0_dp
    enddo
    AE(:, 1) = [+area,  -area, 0.0_dp]
    AE(:, 2) = [-area,  2*area, -area]
    AE(:, 3) = [0.0_dp, -area,  +area]
    do i = 1, ny
        do j = 1, nx
            elem(1) = idx(i, j)
            elem(2) = idx(i, mod(j, nx) + 1)
            elem(3) = idx(mod(i, ny) + 1, mod(j, nx) + 1)
            call S%add(elem, elem, AE)
            call M%add(elem, elem, BE)
            elem(2) = idx(mod(i, ny) + 1, j)
            call S%add(elem, elem, AE)
            call M%add(elem, elem, BE)
        enddo
    enddo
    nq = max(nx, ny)
    allocate(T(3, nq), U(nn, nq), V(nn, nq), Q(nq, nq), w(nn), z(nn))
    ks => hip_cg(1.0d-15)
    call M%set_solver(ks)
    call hip_generalized_lanczos(S, M, T, V)
    do i = 1, nq
        call M%matvec(V(:, i), U(:, i))
    enddo
    do i = 2, nq - 1
        call S%matvec(V(:, i), w)
        z = T(2, i) * U(:, i) + T(1, i-1) * U(:, i-1) + T(3, i) * U(:, i+1)
        err = dsqrt(sum((w - z) * (w - z)) / sum(w * w))
        if (err > 1.0e-14) call fail('hip_generalized_lanczos: three-term recurrence failed', err)
    enddo
    Q = matmul(transpose(V), U)
    do i = 1, nq
        Q(i, i) = Q(i, i) - 1.0_dp
    enddo
    Q = matmul(transpose(Q), Q)
    err = 0.0_dp
    do i = 1, nq
        err = err + Q(i, i)
    enddo
    err = dsqrt(err) / nq
    print *, 'hip_generalized_lanczos: recurrence within 1e-14; || V^t B V - I ||_F / n =', err
    ! (no re-orthogonalisation in the generalized process, eigensolver.f90:128-147: the reference's own run loses
    ! B-orthogonality at the same rate; its test prints and goes on)
    call ks%destroy()
    deallocate(ks)
    call S%destroy()
    call M%destroy()
contains
    integer function idx(i, j)
        integer, intent(in) :: i, j
        idx = ny * (j - 1) + i        ! eigensolver_test_generalized_lanczos.f90:208-214 (indx)
    end function idx
end subroutine test_generalized_lanczos


end program hip_binding_test
