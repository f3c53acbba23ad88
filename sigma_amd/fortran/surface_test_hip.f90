!==========================================================================!
! surface_test_hip -- the rest of the C ABI through module sigma_hip (no   !
! dependency on the reference; builds and runs on the GPU box):            !
!   1. hip_csr_from_edges: device assembly == a host loop with the         !
!      reference's semantics (first occurrence kept, last value wins)      !
!   2. hip_sparse_matrix: a 2 x 2 composite of hip_csr_matrix leaves, the  !
!      block loop of sparse_matrix_composites.f90:1076-1099 == the product !
!      of the whole matrix, bit for bit; hip_cg on the composite           !
!   3. hip_lanczos: three-term recurrence + orthogonality                  !
!      (test/eigensolver_test_lanczos.f90:131-165, same thresholds)        !
!   4. hip_generalized_lanczos with B's solver = hip_cg(1d-15)             !
!      (test/eigensolver_test_generalized_lanczos.f90:150-180)             !
!   5. device vectors (sgm_malloc / sgm_memcpy / sgm_free): a solve whose  !
!      vectors never leave HBM == the same solve on host vectors           !
!   6. hip_dot / hip_axpy (vectors.f90 statements)                         !
! Exit code = verdict, like the reference's CTest programs.                !
!==========================================================================!
program surface_test_hip

use iso_c_binding
use sigma_hip

implicit none

    call test_from_edges()
    call test_composite()
    call test_lanczos()
    call test_generalized_lanczos()
    call test_device_vectors()
    print *, 'all sigma_hip surface checks passed'

contains

subroutine fail(msg, val)
    character(len=*), intent(in) :: msg
    real(dp), intent(in) :: val
    print *, msg
    print *, 'Error:', val
    call exit(1)
end subroutine fail


!--------------------------------------------------------------------------!
subroutine poisson_csr(nx, ny, ptr, node, val)                             !
!--------------------------------------------------------------------------!
! 5-point matrix, rows in the insertion order S, W, C, E, N                !
!--------------------------------------------------------------------------!
    integer, intent(in) :: nx, ny
    integer, allocatable, intent(out) :: ptr(:), node(:)
    real(dp), allocatable, intent(out) :: val(:)
    integer :: n, k, i, j, t
    n = nx * ny
    allocate(ptr(n + 1), node(5 * n), val(5 * n))
    t = 0
    do k = 1, n
        ptr(k) = t + 1
        i = mod(k - 1, nx) + 1
        j = (k - 1) / nx + 1
        if (j > 1)  then; t = t + 1; node(t) = k - nx; val(t) = -1.0_dp; endif
        if (i > 1)  then; t = t + 1; node(t) = k - 1;  val(t) = -1.0_dp; endif
        t = t + 1; node(t) = k; val(t) = 4.0_dp
        if (i < nx) then; t = t + 1; node(t) = k + 1;  val(t) = -1.0_dp; endif
        if (j < ny) then; t = t + 1; node(t) = k + nx; val(t) = -1.0_dp; endif
    enddo
    ptr(n + 1) = t + 1
    node = node(1 : t)
    val = val(1 : t)
end subroutine poisson_csr


!--------------------------------------------------------------------------!
subroutine test_from_edges()                                               !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: A
    integer(c_int32_t), allocatable :: ei(:), ej(:)
    real(dp), allocatable :: ev(:), x(:), y(:), z(:)
    integer, allocatable :: cnt(:), hptr(:), hnode(:)
    real(dp), allocatable :: hval(:)
    integer :: nn, ne, i, k, t
    logical :: found

    nn = 150
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
    ei(k + 2) = 5;  ej(k + 2) = 6;  ev(k + 2) = -7.0_dp      ! a repeated edge: its value wins, its place is the first one's
    ei(k + 3) = 1;  ej(k + 3) = nn; ev(k + 3) = 0.25_dp

    ! host restatement of the reference's sequence: add_edge ignores repeats, rows keep insertion order, set_value overwrites
    allocate(cnt(nn), hptr(nn + 1), hnode(4 * nn), hval(4 * nn))
    cnt = 0
    hnode = 0
    ! (rows of at most 4 distinct entries here: a fixed stride of 4 slots per row, compacted afterwards)
    do k = 1, ne
        found = .false.
        do t = 1, cnt(ei(k))
            if (hnode((ei(k) - 1) * 4 + t) == ej(k)) then
                hval((ei(k) - 1) * 4 + t) = ev(k)
                found = .true.
            endif
        enddo
        if (.not. found) then
            cnt(ei(k)) = cnt(ei(k)) + 1
            hnode((ei(k) - 1) * 4 + cnt(ei(k))) = ej(k)
            hval((ei(k) - 1) * 4 + cnt(ei(k))) = ev(k)
        endif
    enddo
    call hip_csr_from_edges(A, nn, nn, ei, ej, ev)
    t = 0
    do i = 1, nn
        if (A%ptr(i) /= t + 1) call fail('hip_csr_from_edges: ptr differs at row', 1.0_dp * i)
        do k = 1, cnt(i)
            t = t + 1
            if (A%node(t) /= hnode((i - 1) * 4 + k) .or. A%val(t) /= hval((i - 1) * 4 + k)) &
                & call fail('hip_csr_from_edges: entry differs in row', 1.0_dp * i)
        enddo
    enddo
    if (A%ptr(nn + 1) /= t + 1) call fail('hip_csr_from_edges: nnz differs', 1.0_dp * t)
    allocate(x(nn), y(nn), z(nn))
    do i = 1, nn
        x(i) = dsin(0.37_dp * i)
    enddo
    call A%matvec(x, y)
    do i = 1, nn
        z(i) = 0.0_dp
        do k = A%ptr(i), A%ptr(i + 1) - 1
            z(i) = z(i) + A%val(k) * x(A%node(k))
        enddo
    enddo
    if (any(y /= z)) call fail('hip_csr_from_edges: product differs from the row loop', maxval(dabs(y - z)))
    print *, 'hip_csr_from_edges: arrays and product identical to the host sequence; kernel ', A%kernel_name()
    call A%destroy()
end subroutine test_from_edges


!--------------------------------------------------------------------------!
subroutine test_composite()                                                !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: W
    type(hip_csr_matrix), target :: B11, B12, B21, B22
    type(hip_sparse_matrix) :: C
    type(hip_linear_solver), pointer :: s
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:), x(:), y1(:), y2(:), f(:), u1(:), u2(:)
    integer :: nx, ny, n, n1, i, its1
    real(dp) :: err

    nx = 32
    ny = 24
    n = nx * ny
    n1 = 400
    call poisson_csr(nx, ny, ptr, node, val)
    call W%init(n, n, ptr, node)
    W%val = val
    call split(1, n1, 1, n1, B11)
    call split(1, n1, n1 + 1, n, B12)
    call split(n1 + 1, n, 1, n1, B21)
    call split(n1 + 1, n, n1 + 1, n, B22)
    call C%set_num_blocks(2, 2)
    call C%set_block_sizes([n1, n - n1], [n1, n - n1])
    call C%set_submatrix(1, 1, B11)
    call C%set_submatrix(1, 2, B12)
    call C%set_submatrix(2, 1, B21)
    call C%set_submatrix(2, 2, B22)
    allocate(x(n), y1(n), y2(n), f(n), u1(n), u2(n))
    do i = 1, n
        x(i) = dsin(0.001_dp * i)
    enddo
    ! per row the blocks' partial sums are added block by block (composite_matvec_add): for the rows cut by the split
    ! that is a different association than the whole row's left-to-right sum -- compare against that loop, not against W
    call C%matvec(x, y1)
    y2 = 0.0_dp
    call B11%matvec_add(x(1 : n1), y2(1 : n1))
    call B12%matvec_add(x(n1 + 1 : n), y2(1 : n1))
    call B21%matvec_add(x(1 : n1), y2(n1 + 1 : n))
    call B22%matvec_add(x(n1 + 1 : n), y2(n1 + 1 : n))
    if (any(y1 /= y2)) call fail('composite product differs from the block loop over its leaves', maxval(dabs(y1 - y2)))
    call W%matvec(x, y2)
    if (maxval(dabs(y1 - y2)) > 1.0e-15) call fail('composite product differs from the whole matrix', maxval(dabs(y1 - y2)))
    call C%matvec_t(x, y1)
    call W%matvec_t(x, y2)
    if (maxval(dabs(y1 - y2)) > 1.0e-15) call fail('composite transpose product differs', maxval(dabs(y1 - y2)))
    f = 1.0_dp / n
    u1 = 0.0_dp
    s => hip_cg(1.d-12)
    call s%setup(W)
    call s%solve(W, u1, f)
    its1 = s%iterations
    call s%destroy()
    deallocate(s)
    u2 = 0.0_dp
    s => hip_cg(1.d-12)
    call s%setup(C)
    call s%solve(C, u2, f)
    err = maxval(dabs(u1 - u2)) / maxval(dabs(u1))
    print *, 'composite: hip_cg iterations', s%iterations, ' whole matrix', its1, ' relative difference', err
    if (abs(s%iterations - its1) > 1 .or. err > 1.0e-12) call fail('hip_cg on the composite differs from the whole matrix', err)
    call s%destroy()
    deallocate(s)
    call C%destroy()
    call B11%destroy(); call B12%destroy(); call B21%destroy(); call B22%destroy()
    call W%destroy()
contains
    subroutine split(i0, i1, j0, j1, B)
        ! block (rows i0..i1, columns j0..j1) of W, entries in W's stored order
        integer, intent(in) :: i0, i1, j0, j1
        type(hip_csr_matrix), intent(inout) :: B
        integer, allocatable :: bp(:), bn(:)
        real(dp), allocatable :: bv(:)
        integer :: r, k, t
        allocate(bp(i1 - i0 + 2), bn(size(node)), bv(size(node)))
        t = 0
        do r = i0, i1
            bp(r - i0 + 1) = t + 1
            do k = ptr(r), ptr(r + 1) - 1
                if (node(k) >= j0 .and. node(k) <= j1) then
                    t = t + 1
                    bn(t) = node(k) - j0 + 1
                    bv(t) = val(k)
                endif
            enddo
        enddo
        bp(i1 - i0 + 2) = t + 1
        call B%init(i1 - i0 + 1, j1 - j0 + 1, bp, bn(1 : t))
        B%val = bv(1 : t)
    end subroutine split
end subroutine test_composite


!--------------------------------------------------------------------------!
subroutine test_lanczos()                                                  !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: A
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:), T(:,:), V(:,:), Q(:,:), x(:), y(:), q1(:)
    integer :: n, nq, i
    real(dp) :: err

    call poisson_csr(16, 12, ptr, node, val)
    n = 16 * 12
    call A%init(n, n, ptr, node)
    A%val = val
    nq = int(dsqrt(1.0_dp * n))
    allocate(T(3, nq), V(n, nq), Q(nq, nq), x(n), y(n), q1(n))
    do i = 1, n
        q1(i) = dcos(1.7_dp * i) + 0.3_dp * dsin(0.11_dp * i * i)
    enddo
    call hip_lanczos(A, T, V, q1)
    do i = 2, nq - 1
        call A%matvec(V(:, i), x)
        y = T(2, i) * V(:, i) + T(1, i - 1) * V(:, i - 1) + T(3, i) * V(:, i + 1)
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
    print *, 'hip_lanczos: recurrence and orthogonality within 1e-14'
    call A%destroy()
end subroutine test_lanczos


!--------------------------------------------------------------------------!
subroutine test_generalized_lanczos()                                      !
!--------------------------------------------------------------------------!
! A = the 5-point matrix, B = a diagonally dominant "mass" matrix on the   !
! same pattern; B w = v by hip_cg(1d-15) in every step                     !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: A, B
    type(hip_linear_solver), pointer :: sb
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:), T(:,:), V(:,:), U(:,:), w(:), z(:), q1(:)
    integer :: n, nq, i, k
    real(dp) :: err

    call poisson_csr(16, 12, ptr, node, val)
    n = 16 * 12
    call A%init(n, n, ptr, node)
    A%val = val
    call B%init(n, n, ptr, node)
    do i = 1, n
        do k = ptr(i), ptr(i + 1) - 1
            if (node(k) == i) then
                B%val(k) = 1.0_dp / 2.0_dp
            else
                B%val(k) = 1.0_dp / 12.0_dp
            endif
        enddo
    enddo
    nq = 12
    allocate(T(3, nq), V(n, nq), U(n, nq), w(n), z(n), q1(n))
    do i = 1, n
        q1(i) = dcos(1.7_dp * i) + 0.3_dp * dsin(0.11_dp * i * i)
    enddo
    sb => hip_cg(1.0d-15)
    call sb%setup(B)
    call hip_generalized_lanczos(A, B, sb, T, V, q1)
    do i = 1, nq
        call B%matvec(V(:, i), U(:, i))
    enddo
    do i = 2, nq - 1
        call A%matvec(V(:, i), w)
        z = T(2, i) * U(:, i) + T(1, i - 1) * U(:, i - 1) + T(3, i) * U(:, i + 1)
        err = dsqrt(sum((w - z) * (w - z)) / sum(w * w))
        if (err > 1.0e-13) call fail('hip_generalized_lanczos: three-term recurrence failed', err)
    enddo
    print *, 'hip_generalized_lanczos: three-term recurrence within 1e-13'
    call sb%destroy()
    deallocate(sb)
    call A%destroy()
    call B%destroy()
end subroutine test_generalized_lanczos


!--------------------------------------------------------------------------!
subroutine test_device_vectors()                                           !
!--------------------------------------------------------------------------!
    type(hip_csr_matrix) :: A
    type(hip_linear_solver), pointer :: s, pc
    type(hip_device_vector) :: xd, bd
    integer, allocatable :: ptr(:), node(:)
    real(dp), allocatable :: val(:), f(:), u1(:), u2(:)
    integer :: n, its1
    real(dp) :: d

    call poisson_csr(40, 30, ptr, node, val)
    n = 40 * 30
    call A%init(n, n, ptr, node)
    A%val = val
    allocate(f(n), u1(n), u2(n))
    f = 1.0_dp / n
    u1 = 0.0_dp
    s => hip_cg(1.d-12)
    pc => hip_jacobi()
    call s%setup(A)
    call pc%setup(A)
    call s%solve(A, u1, f, pc)
    its1 = s%iterations
    call xd%alloc(n)
    call bd%alloc(n)
    u2 = 0.0_dp
    call xd%upload(u2)
    call bd%upload(f)
    call s%setup(A)                       ! iterations back to 0 (cg_solvers.f90:72)
    call s%solve_device(A, xd, bd, pc)
    call xd%download(u2)
    if (s%iterations /= its1 .or. any(u1 /= u2)) call fail('solve on device vectors differs from the one on host vectors', &
        & maxval(dabs(u1 - u2)))
    print *, 'device vectors: the same solve, bit for bit,', its1, 'iterations'
    ! s%tolerance is live (cg_solvers.f90:17,133; set_params :95-111): a loose tolerance on the same handle stops early, the
    ! tight one set afterwards continues from there, and `iterations` accumulates over the two solves (cg_solvers.f90:145)
    block
        integer :: it_loose
        call s%setup(A)
        u2 = 0.0_dp
        s%tolerance = 1.d-6
        call s%solve(A, u2, f, pc)
        it_loose = s%iterations
        if (it_loose >= its1 .or. maxval(dabs(u1 - u2)) < 1.d-11) call fail('a tolerance of 1e-6 edited on the live solver was ignored', &
            & real(it_loose, dp))
        call s%set_params(1.d-12)
        call s%solve(A, u2, f, pc)
        if (s%iterations <= it_loose .or. maxval(dabs(u1 - u2)) > 1.d-9 * maxval(dabs(u1))) &
            & call fail('set_params(1e-12) on the live solver did not reach the device loop', maxval(dabs(u1 - u2)))
        print *, 'live tolerance: 1e-6 ->', it_loose, ' iterations, then 1e-12 ->', s%iterations, ' in total'
    end block
    ! the reordering ILDU(0) preconditioner: A, f, u stay in natural order, the factors are those of the colour-ordered matrix
    block
        type(hip_linear_solver), pointer :: pr, pn
        real(dp), allocatable :: u3(:), u4(:)
        integer :: itn
        allocate(u3(n), u4(n))
        pn => hip_ldu()
        pr => hip_ldu(reorder = "colour")
        call pn%setup(A)
        call pr%setup(A)
        block      ! sgm_pc_info: what will serve the applies -- natural order: many dependency levels; colour order: two
            integer :: lv(2), lc(2), path, ncol
            real(dp) :: est
            character(len=80) :: what
            call pn%info(lv, path, ncol, est, what)
            print *, 'hip_ldu():                  ', trim(what), ' about', est, 'us per apply'
            if (ncol /= 0) call fail('natural-order ILDU reports an ordering', real(ncol, dp))
            call pr%info(lc, path, ncol, est, what)
            print *, 'hip_ldu(reorder = "colour"):', trim(what), ' about', est, 'us per apply'
            if (ncol /= 2 .or. lc(1) /= 2 .or. lc(2) /= 2 .or. path /= 1 .or. lv(1) <= lc(1)) &
                & call fail('colour-ordered ILDU is not two row-space levels', real(lc(1), dp))
        end block
        call s%setup(A)
        u3 = 0.0_dp
        call s%solve(A, u3, f, pn)
        itn = s%iterations
        call s%setup(A)
        u4 = 0.0_dp
        call s%solve(A, u4, f, pr)
        print *, 'ILDU-PCG iterations: natural order', itn, ' colour order (reorder)', s%iterations
        if (maxval(dabs(u3 - u4)) > 1.0e-11 * maxval(dabs(u3))) call fail('reordering ILDU-PCG differs from the natural-order solve', &
            & maxval(dabs(u3 - u4)))
        call pn%destroy()
        call pr%destroy()
        deallocate(pn, pr)
    end block
    ! dot_product / axpy statements (cg_solvers.f90:131,137)
    d = hip_dot(f, u1)
    if (dabs(d - dot_product(f, u1)) > 1.0e-15 * dabs(d)) call fail('hip_dot differs from dot_product', d)
    u2 = u1
    call hip_axpy(0.5_dp, f, u2)
    if (any(u2 /= u1 + 0.5_dp * f)) call fail('hip_axpy differs from y + alpha x', 0.0_dp)
    call xd%free()
    call bd%free()
    call s%destroy()
    call pc%destroy()
    deallocate(s, pc)
    call A%destroy()
end subroutine test_device_vectors

end program surface_test_hip
