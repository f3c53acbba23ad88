!==========================================================================!
! hip_dist_test -- TEST INFRASTRUCTURE.  One RANK of a row-partitioned     !
! solve through oracle/hip_binding.f90 (hip_comm + hip_dist_csr_matrix):   !
!     hip_dist_test <rank> <nranks> <id file> [<device>]                   !
! started once per rank by tests/test_gpu_multirank.py (on the GPU boxes   !
! of this project: every rank on the one GPU, SGM_RCCL_LIB pointing at the !
! host-staged stand-in transport; on a multi-GPU node: over real RCCL).    !
! Every rank assembles the SAME 5-point matrix with the reference's own    !
! graph / matrix code, keeps its row block on the device, and checks       !
!   * its rows of A x against the same rows of the reference's             !
!     csr_matvec_add on the whole matrix: bit for bit;                     !
!   * hip_cg / hip_cg + hip_jacobi on the distributed operator (dots       !
!     all-reduced across the ranks) against the reference's own cg() on    !
!     the whole matrix on the host: iterations +-1, 1e-12 relative.        !
! No MPI: the RCCL id travels through <id file>.                           !
!==========================================================================!
program hip_dist_test

use iso_c_binding
use types, only: dp
use graphs
use sparse_matrices
use linear_operator_interface
use cg_solvers
use hip_matrices
use hip_solvers

implicit none

    type(hip_comm) :: comm
    type(hip_dist_csr_matrix) :: Ad
    type(csr_matrix) :: A
    class(graph_interface), pointer :: g
    class(linear_solver), pointer :: solver, pc
    real(dp), allocatable :: x(:), y(:), yl(:), f(:), u(:), ul(:)
    character(len=256) :: arg, id_file
    integer :: rank, nranks, device, nx, ny, n, i, j, k, r0, r1, its_ref, its_hip
    real(dp) :: err

    call getarg(1, arg); read(arg, *) rank
    call getarg(2, arg); read(arg, *) nranks
    call getarg(3, id_file)
    device = 0
    if (command_argument_count() >= 4) then
        call getarg(4, arg); read(arg, *) device
    endif

    !----------------------------------------------------------------------!
    ! the 5-point matrix of a 48 x 40 grid, assembled by the reference      !
    !----------------------------------------------------------------------!
    nx = 48
    ny = 40
    n = nx * ny
    allocate(ll_graph :: g)
    call g%init(n, n)
    do k = 1, n
        i = mod(k - 1, nx) + 1
        j = (k - 1) / nx + 1
        if (j > 1)  call g%add_edge(k, k - nx)
        if (i > 1)  call g%add_edge(k, k - 1)
        call g%add_edge(k, k)
        if (i < nx) call g%add_edge(k, k + 1)
        if (j < ny) call g%add_edge(k, k + nx)
    enddo
    call convert_graph_type(g, "compressed sparse")
    call A%init(n, n)
    call A%set_graph(g)
    call A%zero()
    do k = 1, n
        i = mod(k - 1, nx) + 1
        j = (k - 1) / nx + 1
        if (j > 1)  call A%set_value(k, k - nx, -1.0_dp)
        if (i > 1)  call A%set_value(k, k - 1, -1.0_dp)
        call A%set_value(k, k, 4.0_dp + 0.001_dp * mod(k, 7))
        if (i < nx) call A%set_value(k, k + 1, -1.0_dp)
        if (j < ny) call A%set_value(k, k + nx, -1.0_dp)
    enddo

    !----------------------------------------------------------------------!
    ! this rank's row block on its GPU                                      !
    !----------------------------------------------------------------------!
    call comm%init(rank, nranks, trim(id_file), device)
    call Ad%distribute(comm, A)
    r0 = Ad%row_first
    r1 = Ad%row_last
    print *, 'rank', rank, 'owns rows', r0, '..', r1, ' reads', Ad%x_len, 'entries of x per product'
    if (nranks > 1 .and. Ad%x_len <= Ad%nrow) then
        print *, 'rank', rank, ': no halo on a partitioned grid?'
        call exit(1)
    endif

    allocate(x(n), y(n), yl(Ad%nrow), f(n), u(n), ul(Ad%nrow))
    do k = 1, n
        x(k) = dsin(0.001_dp * k)
    enddo
    call A%matvec(x, y)                       ! the reference, whole matrix, host
    call Ad%matvec(x(r0 : r1), yl)            ! this rank's rows, device, halo fetched from the neighbours
    if (any(yl /= y(r0 : r1))) then
        print *, 'rank', rank, ': distributed product differs from the reference rows'
        call exit(1)
    endif
    print *, 'rank', rank, ': rows of A x bit-identical to csr_matvec_add'

    !----------------------------------------------------------------------!
    ! reference cg() on the host vs hip_cg on the distributed operator      !
    !----------------------------------------------------------------------!
    f = 1.0_dp / n
    u = 0.0_dp
    solver => cg(1.d-12)
    call solver%setup(A)
    call solver%solve(A, u, f)
    select type(solver)
        type is(cg_solver)
            its_ref = solver%iterations
    end select
    call solver%destroy()
    deallocate(solver)

    ul = 0.0_dp
    solver => hip_cg(1.d-12)
    call solver%setup(Ad)
    call solver%solve(Ad, ul, f(r0 : r1))
    select type(solver)
        type is(hip_krylov_solver)
            its_hip = solver%iterations
    end select
    err = maxval(dabs(ul - u(r0 : r1))) / maxval(dabs(u))
    print *, 'rank', rank, ': reference cg', its_ref, 'iterations, distributed hip_cg', its_hip, ' relative difference', err
    if (abs(its_hip - its_ref) > 1 .or. err > 1.0e-12) then
        print *, 'rank', rank, ': distributed hip_cg differs from the reference solve'
        call exit(1)
    endif

    pc => hip_jacobi()
    call pc%setup(Ad)
    ul = 0.0_dp
    call solver%solve(Ad, ul, f(r0 : r1), pc)
    err = maxval(dabs(ul - u(r0 : r1))) / maxval(dabs(u))
    print *, 'rank', rank, ': distributed hip_cg + hip_jacobi relative difference', err
    if (err > 1.0e-11) then
        print *, 'rank', rank, ': distributed Jacobi-PCG differs from the reference solve'
        call exit(1)
    endif

    call solver%destroy()
    call pc%destroy()
    deallocate(solver, pc)
    call Ad%destroy()
    call comm%destroy()
    call A%destroy()
    print *, 'rank', rank, ': hip_dist_test passed'

end program hip_dist_test
