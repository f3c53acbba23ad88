!==========================================================================!
! dist_test_hip -- one RANK of a row-partitioned solve through module      !
! sigma_hip (hip_comm + hip_dist_csr_matrix):                              !
!     dist_test_hip <rank> <nranks> <id file> [<device>]                   !
! started once per rank (tests/test_gpu_multirank.py; on a one-GPU box     !
! every rank on device 0 over the host-staged stand-in for RCCL).  Every   !
! rank assembles the same 5-point matrix, keeps its row block on the       !
! device and checks its rows of A x bit for bit and the distributed hip_cg !
! against the same solve on one GPU (iterations +-1, 1e-12).               !
!==========================================================================!
program dist_test_hip

use iso_c_binding
use sigma_hip

implicit none

    type(hip_comm) :: comm
    type(hip_csr_matrix) :: A
    type(hip_dist_csr_matrix) :: Ad
    type(hip_linear_solver), pointer :: s
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:), x(:), y(:), yl(:), f(:), u(:), ul(:)
    character(len=256) :: arg, id_file
    integer :: rank, nranks, device, nx, ny, n, k, i, j, t, r0, r1, its1
    real(dp) :: err

    call getarg(1, arg); read(arg, *) rank
    call getarg(2, arg); read(arg, *) nranks
    call getarg(3, id_file)
    device = 0
    if (command_argument_count() >= 4) then
        call getarg(4, arg); read(arg, *) device
    endif

    nx = 48
    ny = 40
    n = nx * ny
    allocate(ptr(n + 1), node(5 * n), val(5 * n))
    t = 0
    do k = 1, n
        ptr(k) = t + 1
        i = mod(k - 1, nx) + 1
        j = (k - 1) / nx + 1
        if (j > 1)  then; t = t + 1; node(t) = k - nx; val(t) = -1.0_dp; endif
        if (i > 1)  then; t = t + 1; node(t) = k - 1;  val(t) = -1.0_dp; endif
        t = t + 1; node(t) = k; val(t) = 4.0_dp + 0.001_dp * mod(k, 7)
        if (i < nx) then; t = t + 1; node(t) = k + 1;  val(t) = -1.0_dp; endif
        if (j < ny) then; t = t + 1; node(t) = k + nx; val(t) = -1.0_dp; endif
    enddo
    ptr(n + 1) = t + 1

    call comm%init(rank, nranks, trim(id_file), device)
    call A%init(n, n, ptr, node(1 : t))
    A%val = val(1 : t)
    call Ad%distribute(comm, A)
    r0 = Ad%row_first
    r1 = Ad%row_last
    print *, 'rank', rank, 'owns rows', r0, '..', r1

    allocate(x(n), y(n), yl(Ad%nrow), f(n), u(n), ul(Ad%nrow))
    do k = 1, n
        x(k) = dsin(0.001_dp * k)
    enddo
    call A%matvec(x, y)                        ! the whole matrix on this GPU
    call Ad%matvec(x(r0 : r1), yl)             ! this rank's rows, halo from the neighbours
    if (any(yl /= y(r0 : r1))) then
        print *, 'rank', rank, ': distributed product differs'
        call exit(1)
    endif

    f = 1.0_dp / n
    u = 0.0_dp
    s => hip_cg(1.d-12)
    call s%setup(A)
    call s%solve(A, u, f)
    its1 = s%iterations
    call s%destroy()
    deallocate(s)
    ul = 0.0_dp
    s => hip_cg(1.d-12)
    call s%setup(Ad)
    call s%solve(Ad, ul, f(r0 : r1))
    err = maxval(dabs(ul - u(r0 : r1))) / maxval(dabs(u))
    print *, 'rank', rank, ': one GPU', its1, 'iterations, distributed', s%iterations, ' relative difference', err
    if (abs(s%iterations - its1) > 1 .or. err > 1.0e-12) then
        print *, 'rank', rank, ': distributed hip_cg differs'
        call exit(1)
    endif
    call s%destroy()
    deallocate(s)
    ! the reordering preconditioner over the ranks: every rank orders its own diagonal block (no communication), block-Jacobi
    ! ILDU(0) of the ordered blocks, the solve in the permuted order rank by rank -- against plain CG's solution
    block
        type(hip_linear_solver), pointer :: pc
        integer :: lv(2), path, ncol
        real(dp) :: est
        character(len=80) :: what
        pc => hip_ldu(reorder = "colour")
        call pc%setup(Ad)
        call pc%info(lv, path, ncol, est, what)
        print *, 'rank', rank, ': hip_ldu(reorder = colour) on its rows: ', trim(what), ',', ncol, 'colours'
        if (ncol /= 2 .or. path /= 1) then
            print *, 'rank', rank, ': the colour-ordered block is not two row-space levels'
            call exit(1)
        endif
        ul = 0.0_dp
        s => hip_cg(1.d-12)
        call s%setup(Ad)
        call s%solve(Ad, ul, f(r0 : r1), pc)
        err = maxval(dabs(ul - u(r0 : r1))) / maxval(dabs(u))
        print *, 'rank', rank, ': colour-ordered block-Jacobi ILDU-PCG', s%iterations, 'iterations, relative difference', err
        if (s%iterations >= its1 .or. err > 1.0e-10) then
            print *, 'rank', rank, ': distributed ILDU-PCG with the reordering preconditioner differs'
            call exit(1)
        endif
        call s%destroy()
        call pc%destroy()
        deallocate(s, pc)
    end block
    call Ad%destroy()
    call comm%destroy()
    call A%destroy()
    print *, 'rank', rank, ': dist_test_hip passed'

end program dist_test_hip
