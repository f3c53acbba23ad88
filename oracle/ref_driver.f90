!==========================================================================!
! ref_driver -- TEST INFRASTRUCTURE ONLY (never shipped, never measured as   !
! the product).  This program is OUR code: it `use`s the modules of the     !
! real reference (danshapero/sigma, compiled in place from /root/reference  !
! by oracle/build_ref.sh) and drives the hot path named in SURVEY.md §8:    !
!                                                                          !
!   ll_graph%add_edge -> convert_graph_type -> A%set_value    (§3 D)       !
!   A%matvec                      linear_operator_interface.f90:185-194    !
!   cg / bicgstab %solve          cg_solvers.f90:116-194,                   !
!                                 bicgstab_solvers.f90:124-237             !
!   jacobi / ldu %setup, %solve   jacobi_solvers.f90:37-81,                 !
!                                 ldu_solvers.f90:95-176                   !
!                                                                          !
! It reads one problem file (written by oracle/make_golden.py), runs it    !
! through the reference and dumps every array as a raw little-endian file  !
! `<outprefix>.<name>.<i4|f8>`, which make_golden.py packs into the        !
! committed fixtures under tests/golden/.                                   !
!                                                                          !
! With a third argument `time:<reps>` nothing is dumped: the driver times  !
! <reps> calls of A%matvec and the solves of the problem file and prints   !
! `matvec_seconds_each=` / `solve ... seconds=` lines (bench.py's           !
! cpu_baseline of kind "reference").                                       !
!                                                                          !
! Problem file (unformatted stream, int32 / float64):                      !
!   n, m, ne, fmt(1=csr,2=ellpack), nsolve                                  !
!   ei(ne), ej(ne)      edges in INSERTION order (1-based)                  !
!   ev(ne)              value given to set_value(ei,ej)                     !
!   x(m)                matvec input                                        !
!   b(n)                right-hand side                                     !
!   nsolve x { solver(1=cg,2=bicgstab), pc(0,1=jacobi,2=ldu), tol(f8) }     !
!==========================================================================!
program ref_driver

use types, only: dp
use graphs
use sparse_matrices
use linear_operator_interface
use cg_solvers
use bicgstab_solvers
use jacobi_solvers
use ldu_solvers
use permutations
use eigensolver

implicit none

    character(len=512) :: infile, outprefix, mode
    logical :: timing, perm_mode, eig_mode, comp_mode
    integer :: nb1, it, jt, bi, bj, r0(3), nbe
    type(sparse_matrix), target :: Scomp
    class(graph_interface), pointer :: gb
    type(csr_matrix), pointer :: Cb
    character(len=8) :: btag
    integer :: nsteps
    real(dp), allocatable :: Tl(:,:), Ql(:,:)
    type(csr_matrix), target :: Bcsr
    class(linear_solver), pointer :: bsolver
    integer :: reps, rep, ncol
    integer, allocatable :: pp(:), cptrs(:)
    integer(8) :: c0, c1, crate
    integer :: n, m, ne, fmt, nsolve, k, s, its
    integer, allocatable :: ei(:), ej(:), skind(:), pkind(:)
    real(dp), allocatable :: ev(:), x(:), b(:), y(:), u(:), z(:), tols(:), yt(:)
    class(graph_interface), pointer :: g
    type(csr_matrix), target :: Acsr
    type(ellpack_matrix), target :: Aell
    class(sparse_matrix_interface), pointer :: A
    class(linear_solver), pointer :: solver, pc
    character(len=16) :: tag
    real(dp) :: t0, t1

    call getarg(1, infile)
    call getarg(2, outprefix)
    timing = .false.
    perm_mode = .false.
    eig_mode = .false.
    comp_mode = .false.
    reps = 0
    if (command_argument_count() >= 3) then
        call getarg(3, mode)
        if (mode(1:4) == 'perm') perm_mode = .true.
        if (mode(1:5) == 'comp:') then
            comp_mode = .true.
            read(mode(6:), *) nb1
        endif
        if (mode(1:4) == 'eig:') then
            eig_mode = .true.
            read(mode(5:), *) nsteps
        endif
        if (mode(1:5) == 'time:') then
            timing = .true.
            read(mode(6:), *) reps
        endif
    endif

    if (infile(1:14) == 'gen:poisson2d:') then
        ! no problem file: the driver lays down the nx x ny 5-point grid itself, edges in the
        ! insertion order S,W,C,E,N of sigma_amd/problems.py (bench.py's cpu_baseline on the full
        ! C2 workload: a 800 MB problem file would only add I/O to the sample)
        call gen_poisson2d(infile(15:))
    else
    open(unit=21, file=trim(infile), access='stream', form='unformatted', &
        & status='old')
    read(21) n, m, ne, fmt, nsolve
    allocate(ei(ne), ej(ne), ev(ne), x(m), b(n), y(n), u(n), z(n))
    allocate(skind(nsolve), pkind(nsolve), tols(nsolve))
    read(21) ei
    read(21) ej
    read(21) ev
    read(21) x
    read(21) b
    do s = 1, nsolve
        read(21) skind(s), pkind(s), tols(s)
    enddo
    close(21)
    endif

    !------------------------------------------------------------------!
    ! Graph: same call sequence as test/solver_test_jacobi.f90:73-101   !
    !------------------------------------------------------------------!
    allocate(ll_graph :: g)
    call g%init(n, m)
    do k = 1, ne
        call g%add_edge(ei(k), ej(k))
    enddo

    if (fmt == 1) then
        call convert_graph_type(g, "compressed sparse")
        call Acsr%init(n, m)
        call Acsr%set_graph(g)
        call Acsr%zero()
        A => Acsr
    else
        call convert_graph_type(g, "ellpack")
        call Aell%init(n, m)
        call Aell%set_graph(g)
        call Aell%zero()
        A => Aell
    endif

    do k = 1, ne
        call A%set_value(ei(k), ej(k), ev(k))
    enddo

    if (timing) then
        y = 0.0_dp
        call A%matvec(x, y)
        call system_clock(c0, crate)
        do rep = 1, reps
            call A%matvec(x, y)
        enddo
        call system_clock(c1)
        print '(a,es12.5,a,i0,a,es12.5)', 'matvec_seconds_each=', &
            & dble(c1 - c0) / dble(crate) / max(reps, 1), ' reps=', reps, &
            & ' checksum=', sum(y)
    endif

    !------------------------------------------------------------------!
    ! Index arrays + values exactly as the reference holds them         !
    !------------------------------------------------------------------!
    if (timing) then
        continue
    elseif (fmt == 1) then
        call dump_i4('ptr', Acsr%g%ptr, size(Acsr%g%ptr))
        call dump_i4('node', Acsr%g%node, size(Acsr%g%node))
        call dump_f8('val', Acsr%val, size(Acsr%val))
    else
        call dump_i4('max_d', [Aell%g%max_d], 1)
        call dump_i4('degrees', Aell%g%degrees, size(Aell%g%degrees))
        call dump_i4('node', reshape(Aell%g%node, [size(Aell%g%node)]), &
            & size(Aell%g%node))
        call dump_f8('val', reshape(Aell%val, [size(Aell%val)]), &
            & size(Aell%val))
    endif

    !------------------------------------------------------------------!
    ! comp:<nb1> -- the same entries as a 2 x 2 composite `sparse_matrix` !
    ! (sparse_matrix_composites.f90:41-162): rows / columns split at nb1, !
    ! every block a csr_matrix built from the edges that fall into it, in !
    ! the insertion order of the edge list.  From here on A is the        !
    ! composite: the products and solves below go through                 !
    ! composite_matvec_add (:1076-1099) / composite_matvec_t_add          !
    ! (:1104-1127) and composite_mat_get_value (:465-485).                !
    !------------------------------------------------------------------!
    if (comp_mode .and. fmt == 1) then
        r0 = [0, nb1, n]
        call Scomp%set_dimensions(n, m)
        call Scomp%set_block_sizes([nb1, n - nb1], [nb1, m - nb1])
        do it = 1, 2
            do jt = 1, 2
                allocate(ll_graph :: gb)
                call gb%init(r0(it + 1) - r0(it), r0(jt + 1) - r0(jt))
                nbe = 0
                do k = 1, ne
                    if (ei(k) > r0(it) .and. ei(k) <= r0(it + 1) .and. ej(k) > r0(jt) .and. ej(k) <= r0(jt + 1)) then
                        call gb%add_edge(ei(k) - r0(it), ej(k) - r0(jt))
                        nbe = nbe + 1
                    endif
                enddo
                call convert_graph_type(gb, "compressed sparse")
                allocate(Cb)
                call Cb%init(r0(it + 1) - r0(it), r0(jt + 1) - r0(jt))
                call Cb%set_graph(gb)
                call Cb%zero()
                do k = 1, ne
                    if (ei(k) > r0(it) .and. ei(k) <= r0(it + 1) .and. ej(k) > r0(jt) .and. ej(k) <= r0(jt + 1)) then
                        call Cb%set_value(ei(k) - r0(it), ej(k) - r0(jt), ev(k))
                    endif
                enddo
                write(btag, '(a,i0,i0)') 'blk', it, jt
                call dump_i4(trim(btag)//'_ptr', Cb%g%ptr, size(Cb%g%ptr))
                call dump_i4(trim(btag)//'_node', Cb%g%node, size(Cb%g%node))
                call dump_f8(trim(btag)//'_val', Cb%val, size(Cb%val))
                call Scomp%set_submatrix(it, jt, Cb)
            enddo
        enddo
        call dump_i4('comp_row_ptr', Scomp%row_ptr, 3)
        call dump_i4('comp_col_ptr', Scomp%col_ptr, 3)
        A => Scomp
    endif

    !------------------------------------------------------------------!
    ! y = A x   and   y2 = y + A x  (matvec, then matvec_add on top)    !
    !------------------------------------------------------------------!
    y = -7.0_dp     ! garbage on purpose: matvec must overwrite
    call A%matvec(x, y)
    call dump_f8('y', y, n)
    call A%matvec_add(x, y)
    call dump_f8('y_add', y, n)

    !------------------------------------------------------------------!
    ! transpose products: yt = A^T b ; yt = yt + A^T b                  !
    ! (linear_operator_interface.f90:199-208 -> csc_matvec_add          !
    !  cs_matrices.f90:627-647 / ellpack_matvec_t_add                   !
    !  ellpack_matrices.f90:670-693)                                    !
    !------------------------------------------------------------------!
    allocate(yt(m))
    yt = -7.0_dp
    call A%matvec_t(b, yt)
    call dump_f8('yt', yt, m)
    call A%matvec_t_add(b, yt)
    call dump_f8('yt_add', yt, m)

    !------------------------------------------------------------------!
    ! Reorderings of the matrix graph (permutations.f90) and the         !
    ! symmetric permutation of the matrix by the colour ordering         !
    ! (cs_matrices.f90:471-490).  From here on A is the permuted matrix: !
    ! the solves below run on it.                                        !
    !------------------------------------------------------------------!
    if (perm_mode .and. fmt == 1) then
        ! text dump of the matrix as assembled (sparse_matrix_interfaces.f90:601-653)
        call A%to_file(trim(outprefix)//'.matrix.txt')
        allocate(pp(n), cptrs(n + 2))
        call breadth_first_search(pp, Acsr%g)
        call dump_i4('bfs_p', pp, n)
        call greedy_coloring(pp, Acsr%g)
        call dump_i4('colors', pp, n)
        cptrs = 0
        call greedy_color_ordering(pp, cptrs, ncol, Acsr%g)
        call dump_i4('color_p', pp, n)
        call dump_i4('color_ptrs', cptrs, ncol + 1)
        call dump_i4('num_colors', [ncol], 1)
        call Acsr%left_permute(pp)
        call Acsr%right_permute(pp)
        call dump_i4('perm_ptr', Acsr%g%ptr, size(Acsr%g%ptr))
        call dump_i4('perm_node', Acsr%g%node, size(Acsr%g%node))
        call dump_f8('perm_val', Acsr%val, size(Acsr%val))
        y = -7.0_dp
        call A%matvec(x, y)
        call dump_f8('perm_y', y, n)
    elseif (perm_mode .and. fmt == 2) then
        ! the same for an ELLPACK matrix (ellpack_graphs.f90:486-541, ellpack_matrices.f90:601-632)
        allocate(pp(n), cptrs(n + 2))
        call breadth_first_search(pp, Aell%g)
        call dump_i4('bfs_p', pp, n)
        cptrs = 0
        call greedy_color_ordering(pp, cptrs, ncol, Aell%g)
        call dump_i4('color_p', pp, n)
        call dump_i4('color_ptrs', cptrs, ncol + 1)
        call dump_i4('num_colors', [ncol], 1)
        call Aell%left_permute(pp)
        call Aell%right_permute(pp)
        call dump_i4('perm_degrees', Aell%g%degrees, size(Aell%g%degrees))
        call dump_i4('perm_node', reshape(Aell%g%node, [size(Aell%g%node)]), size(Aell%g%node))
        call dump_f8('perm_val', reshape(Aell%val, [size(Aell%val)]), size(Aell%val))
        y = -7.0_dp
        call A%matvec(x, y)
        call dump_f8('perm_y', y, n)
    endif

    !------------------------------------------------------------------!
    ! Lanczos (src/eigensolver.f90:27-90) and generalized Lanczos        !
    ! (:95-155) with B = a diagonally dominant matrix on A's graph and   !
    ! B%solve = the reference's CG.  Q(:,1) of each dump is the          !
    ! (time-seeded) start vector the run used.                           !
    !------------------------------------------------------------------!
    if (eig_mode .and. fmt == 1) then
        allocate(Tl(3, nsteps), Ql(n, nsteps))
        call lanczos(A, Tl, Ql)
        call dump_f8('lanczos_T', reshape(Tl, [3 * nsteps]), 3 * nsteps)
        call dump_f8('lanczos_Q', reshape(Ql, [n * nsteps]), n * nsteps)

        call Bcsr%init(n, m)
        call Bcsr%set_graph(g)
        call Bcsr%zero()
        do k = 1, ne
            if (ei(k) == ej(k)) then
                call Bcsr%set_value(ei(k), ej(k), 1.0_dp + mod(ei(k), 7) / 16.0_dp)
            else
                call Bcsr%set_value(ei(k), ej(k), -1.0_dp / 16.0_dp)
            endif
        enddo
        call dump_f8('B_val', Bcsr%val, size(Bcsr%val))
        bsolver => cg(1.0d-14)
        call Bcsr%set_solver(bsolver)
        call generalized_lanczos(A, Bcsr, Tl, Ql)
        call dump_f8('glanczos_T', reshape(Tl, [3 * nsteps]), 3 * nsteps)
        call dump_f8('glanczos_Q', reshape(Ql, [n * nsteps]), n * nsteps)
    endif

    !------------------------------------------------------------------!
    ! Solves                                                            !
    !------------------------------------------------------------------!
    do s = 1, nsolve
        write(tag, '(a,i0)') 's', s

        if (skind(s) == 1) then
            solver => cg(tols(s))
        else
            solver => bicgstab(tols(s))
        endif
        call solver%setup(A)

        nullify(pc)
        if (pkind(s) == 1) then
            pc => jacobi()
        elseif (pkind(s) == 2) then
            pc => ldu(incomplete = .true., level = 0)
        endif

        if (associated(pc)) then
            call pc%setup(A)
            ! one stand-alone preconditioner apply z = M^{-1} b
            z = 0.0_dp
            call pc%solve(A, z, b)
            call dump_f8(trim(tag)//'_pcz', z, n)
            select type(pc)
                type is(jacobi_solver)
                    call dump_f8(trim(tag)//'_idiag', pc%idiag, n)
                type is(sparse_ldu_solver)
                    call dump_i4(trim(tag)//'_Lptr', pc%L%g%ptr, n + 1)
                    call dump_i4(trim(tag)//'_Lnode', pc%L%g%node, &
                        & size(pc%L%g%node))
                    call dump_f8(trim(tag)//'_Lval', pc%L%val, &
                        & size(pc%L%val))
                    call dump_i4(trim(tag)//'_Uptr', pc%U%g%ptr, n + 1)
                    call dump_i4(trim(tag)//'_Unode', pc%U%g%node, &
                        & size(pc%U%g%node))
                    call dump_f8(trim(tag)//'_Uval', pc%U%val, &
                        & size(pc%U%val))
                    call dump_f8(trim(tag)//'_D', pc%D, n)
            end select
        endif

        u = 0.0_dp
        call system_clock(c0, crate)
        if (associated(pc)) then
            call solver%solve(A, u, b, pc)
        else
            call solver%solve(A, u, b)
        endif
        call system_clock(c1)
        t0 = 0.0_dp
        t1 = dble(c1 - c0) / dble(crate)

        its = -1
        select type(solver)
            type is(cg_solver)
                its = solver%iterations
            type is(bicgstab_solver)
                its = solver%iterations
        end select

        call dump_f8(trim(tag)//'_u', u, n)
        call dump_i4(trim(tag)//'_iterations', [its], 1)
        print '(a,i0,a,i0,a,i0,a,i0,a,es10.3)', 'solve ', s, ': solver=', &
            & skind(s), ' pc=', pkind(s), ' iterations=', its, &
            & ' seconds=', t1 - t0

        call solver%destroy()
        deallocate(solver)
        if (associated(pc)) then
            call pc%destroy()
            deallocate(pc)
        endif
    enddo

contains

    subroutine gen_poisson2d(spec)
        character(len=*), intent(in) :: spec
        integer :: nx, ny, ix, iy, row, c
        c = index(spec, ':')
        read(spec(1:c-1), *) nx
        read(spec(c+1:), *) ny
        n = nx * ny
        m = n
        fmt = 1
        nsolve = 0
        ne = 5 * n - 2 * nx - 2 * ny
        allocate(ei(ne), ej(ne), ev(ne), x(m), b(n), y(n), u(n), z(n))
        allocate(skind(0), pkind(0), tols(0))
        k = 0
        do iy = 1, ny
            do ix = 1, nx
                row = (iy - 1) * nx + ix
                if (iy > 1) call put(row, row - nx, -1.0_dp)
                if (ix > 1) call put(row, row - 1, -1.0_dp)
                call put(row, row, 4.0_dp)
                if (ix < nx) call put(row, row + 1, -1.0_dp)
                if (iy < ny) call put(row, row + nx, -1.0_dp)
            enddo
        enddo
        do row = 1, n
            x(row) = sin(0.001_dp * row)
        enddo
        b = 1.0_dp / n
    end subroutine gen_poisson2d

    subroutine put(i, j, v)
        integer, intent(in) :: i, j
        real(dp), intent(in) :: v
        k = k + 1
        ei(k) = i
        ej(k) = j
        ev(k) = v
    end subroutine put

    subroutine dump_i4(name, arr, cnt)
        character(len=*), intent(in) :: name
        integer, intent(in) :: cnt
        integer, intent(in) :: arr(cnt)
        if (timing) return
        open(unit=22, file=trim(outprefix)//'.'//name//'.i4', &
            & access='stream', form='unformatted', status='replace')
        write(22) arr
        close(22)
    end subroutine dump_i4

    subroutine dump_f8(name, arr, cnt)
        character(len=*), intent(in) :: name
        integer, intent(in) :: cnt
        real(dp), intent(in) :: arr(cnt)
        if (timing) return
        open(unit=22, file=trim(outprefix)//'.'//name//'.f8', &
            & access='stream', form='unformatted', status='replace')
        write(22) arr
        close(22)
    end subroutine dump_f8

end program ref_driver
