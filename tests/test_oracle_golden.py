"""Pin the CPU restatement (oracle/) against the REAL reference's outputs.

The fixtures in tests/golden/ were produced by the reference compiled in place with
amdflang (oracle/build_ref.sh, oracle/ref_driver.f90, oracle/make_golden.py).
Bar: bit-exact for index arrays, stored values and matvec; solver solutions within
1e-12 relative with the same iteration count (+-1) -- dot_product's summation order is
compiler-chosen in the reference, so the recurrences are not bitwise comparable.
Also re-checks the known answers of the reference's own tests
(test/solver_test_diffusion_1d.f90:115, test/solver_test_advection_diffusion_1d.f90:122).
"""
import numpy as np
import pytest

from conftest import comp_golden_names, eig_golden_names, golden_names, perm_golden_names
import oracle as orc

CG, BICGSTAB = 1, 2


def build(g):
    n, m = int(g["n"]), int(g["m"])
    cls = orc.CsrMatrix if int(g["fmt"]) == 1 else orc.EllMatrix
    return cls.from_edges(n, m, g["ei"], g["ej"], g["ev"])


@pytest.mark.parametrize("name", golden_names())
def test_index_arrays_and_values_bit_exact(golden, name):
    g = golden(name)
    A = build(g)
    if A.fmt == 1:
        assert np.array_equal(A.ptr, g["ref_ptr"])
        assert np.array_equal(A.node, g["ref_node"])
        assert np.array_equal(A.val, g["ref_val"])
    else:
        assert A.max_d == int(g["ref_max_d"][0])
        assert np.array_equal(A.degrees, g["ref_degrees"])
        assert np.array_equal(A.node.ravel(), g["ref_node"])
        assert np.array_equal(A.val.ravel(), g["ref_val"])


@pytest.mark.parametrize("name", golden_names())
def test_matvec_bit_exact(golden, name):
    g = golden(name)
    A = build(g)
    y = A.matvec(g["x"])
    assert np.array_equal(y, g["ref_y"])
    A.matvec_add(g["x"], y)
    assert np.array_equal(y, g["ref_y_add"])


@pytest.mark.parametrize("name", golden_names())
def test_transpose_matvec_bit_exact(golden, name):
    g = golden(name)
    A = build(g)
    yt = A.matvec_t(g["b"])
    assert np.array_equal(yt, g["ref_yt"])
    A.matvec_t_add(g["b"], yt)
    assert np.array_equal(yt, g["ref_yt_add"])


def _pc(A, kind):
    return {0: lambda A: None, 1: orc.Jacobi, 2: orc.Ildu}[kind](A)


@pytest.mark.parametrize("name", golden_names())
def test_preconditioners(golden, name):
    g = golden(name)
    A = build(g)
    for s, (skind, pkind, tol) in enumerate(g["solves"], 1):
        if int(pkind) == 1:
            pc = orc.Jacobi(A)
            assert np.array_equal(pc.idiag, g[f"ref_s{s}_idiag"])
            assert np.array_equal(pc.solve(g["b"]), g[f"ref_s{s}_pcz"])
        elif int(pkind) == 2:
            pc = orc.Ildu(A)
            assert np.array_equal(pc.Lptr, g[f"ref_s{s}_Lptr"])
            assert np.array_equal(pc.Lnode, g[f"ref_s{s}_Lnode"])
            assert np.array_equal(pc.Uptr, g[f"ref_s{s}_Uptr"])
            assert np.array_equal(pc.Unode, g[f"ref_s{s}_Unode"])
            assert np.array_equal(pc.D, g[f"ref_s{s}_D"])
            assert np.array_equal(pc.Lval, g[f"ref_s{s}_Lval"])
            assert np.array_equal(pc.Uval, g[f"ref_s{s}_Uval"])
            assert np.array_equal(pc.solve(g["b"]), g[f"ref_s{s}_pcz"])


@pytest.mark.parametrize("name", golden_names())
def test_solvers(golden, name):
    """The pinned reference build (amdflang -O2, x86-64) sums `dot_product` with one accumulator, first element to
    last: with that order (the oracle's default dot mode) every solve of every fixture -- CG, PCG (Jacobi, ILDU),
    BiCGStab, PBiCGStab, up to 9388 iterations -- reproduces the reference's iteration count and its solution BIT FOR BIT."""
    g = golden(name)
    A = build(g)
    for s, (skind, pkind, tol) in enumerate(g["solves"], 1):
        pc = _pc(A, int(pkind))
        fn = orc.cg if int(skind) == CG else orc.bicgstab
        u, its, res2, _ = fn(A, g["b"], pc=pc, tol=tol)
        assert its == int(g[f"ref_s{s}_iterations"][0]), (name, s, its)
        assert np.array_equal(u, g[f"ref_s{s}_u"]), (name, s)


def test_other_dot_order_is_only_close(golden):
    """The 4-lane interleaved order (another legal dot_product) drifts from the reference: same answer to the solver
    tolerance, not the same bits -- which is why bit-identity needs the reference's own order (sgm option dot_order = 1)."""
    g = golden("poisson2d_32x24")
    A = build(g)
    orc.set_dot_mode(1)
    try:
        u, its, _, _ = orc.cg(A, g["b"], tol=1e-12)
    finally:
        orc.set_dot_mode(0)
    uref = g["ref_s1_u"]
    assert abs(its - int(g["ref_s1_iterations"][0])) <= 1
    assert not np.array_equal(u, uref) and np.abs(u - uref).max() / np.abs(uref).max() <= 1e-12


def test_reference_known_answers(golden):
    # test/solver_test_diffusion_1d.f90:104-115 -- 64 iterations, max error <= 1e-14
    g = golden("diffusion1d_ell_127")
    A = build(g)
    u, its, _, _ = orc.cg(A, g["b"], tol=1e-16)
    assert its == 64 == int(g["ref_s1_iterations"][0])
    assert np.abs(u - g["analytic"]).max() <= 1e-14
    assert np.abs(g["ref_s1_u"] - g["analytic"]).max() <= 1e-14
    # test/solver_test_advection_diffusion_1d.f90:111-122 -- max error <= 1e-8
    g = golden("advdiff1d_ell_1024")
    A = build(g)
    u, its, _, _ = orc.bicgstab(A, g["b"], tol=1e-12)
    assert int(g["ref_s1_iterations"][0]) == 1133
    assert np.abs(u - g["analytic"]).max() <= 1e-8
    assert np.abs(g["ref_s1_u"] - g["analytic"]).max() <= 1e-8


def test_gmres_unpinned_but_consistent(golden):
    """GMRES has no reference counterpart: pinned by the analytic solution
    and by agreement with the BiCGStab reference solution on the same matrix."""
    g = golden("advdiff1d_csr_1024")
    A = build(g)
    u, its, res, _ = orc.gmres(A, g["b"], tol=1e-12, restart=30, max_iter=200000)
    assert res <= 1e-12
    assert np.abs(u - g["analytic"]).max() <= 1e-8
    # cond(A) ~ n^2 ~ 1e6: two solutions with residual <= 1e-12 agree to ~1e-6 at best
    assert np.abs(u - g["ref_s1_u"]).max() / np.abs(g["ref_s1_u"]).max() <= 1e-7
    g = golden("random_skew_128")
    A = build(g)
    u, its, res, _ = orc.gmres(A, g["b"], tol=1e-13, restart=30)
    assert np.abs(u - g["ref_s1_u"]).max() / np.abs(g["ref_s1_u"]).max() <= 1e-11


def test_direct_csr_generators_match_graph_build():
    """The vectorised generators used at benchmark sizes produce exactly the arrays the
    reference's graph build produces from the edge list."""
    from sigma_amd import problems as P
    for (ptr, node, val), edges, n in [
        (P.poisson2d_csr(7, 5), P.poisson2d_edges(7, 5), 35),
        (P.laplace3d_csr(4, 3, 5), P.laplace3d_edges(4, 3, 5), 60),
        (P.tridiag_csr(9, 2.0, -0.5, -1.5), P.tridiag_edges(9, 2.0, -0.5, -1.5), 9),
        (P.tridiag_csr(1, 2.0, -0.5, -1.5), P.tridiag_edges(1, 2.0, -0.5, -1.5), 1),
    ]:
        A = orc.CsrMatrix.from_edges(n, n, *edges)
        assert np.array_equal(A.ptr, ptr)
        assert np.array_equal(A.node, node)
        assert np.array_equal(A.val, val)


@pytest.mark.parametrize("name", perm_golden_names())
def test_reorderings_and_permuted_matrix_bit_exact(golden, name):
    """permutations.f90 (BFS numbering, greedy colouring, colour ordering) and the symmetric
    permutation of the matrix, against the reference's own output: index work, bit-exact."""
    g = golden(name)
    A = build(g)
    assert np.array_equal(orc.bfs_order(A), g["ref_bfs_p"])
    assert np.array_equal(orc.greedy_coloring(A), g["ref_colors"])
    p, ptrs, nc = orc.greedy_color_ordering(A)
    assert nc == int(g["ref_num_colors"][0])
    assert np.array_equal(p, g["ref_color_p"])
    assert np.array_equal(ptrs, g["ref_color_ptrs"])
    B = orc.permuted(A, p, p)
    assert np.array_equal(B.ptr, g["ref_perm_ptr"])
    assert np.array_equal(B.node, g["ref_perm_node"])
    assert np.array_equal(B.val, g["ref_perm_val"])
    assert np.array_equal(B.matvec(g["x"]), g["ref_perm_y"])


@pytest.mark.parametrize("name", perm_golden_names())
def test_solves_on_the_permuted_matrix(golden, name):
    """The fixture's solves ran on the colour-ordered matrix (ILDU(0) there has as many
    dependency levels as colours)."""
    g = golden(name)
    A = build(g)
    p, _, _ = orc.greedy_color_ordering(A)
    B = orc.permuted(A, p, p)
    for k, (skind, pkind, tol) in enumerate(g["solves"], start=1):
        pc = _pc(B, int(pkind))
        if int(pkind) == 2:
            assert np.array_equal(pc.D, g[f"ref_s{k}_D"])
            assert np.array_equal(pc.Lval, g[f"ref_s{k}_Lval"])
            assert np.array_equal(pc.solve(g["b"]), g[f"ref_s{k}_pcz"])
        fn = orc.cg if int(skind) == CG else orc.bicgstab
        u, its, _, _ = fn(B, g["b"], pc=pc, tol=tol)
        ref_u, ref_its = g[f"ref_s{k}_u"], int(g[f"ref_s{k}_iterations"][0])
        assert abs(its - ref_its) <= 1, (k, its, ref_its)
        assert np.abs(u - ref_u).max() / np.abs(ref_u).max() <= 1e-12, k


@pytest.mark.parametrize("name", perm_golden_names(ell=True))
def test_ellpack_reorderings_and_permutation_bit_exact(golden, name):
    """The same on an ELLPACK matrix: BFS / colour ordering over the first degrees(i) slots, then
    ellpack left/right permute (ellpack_matrices.f90:601-632), against the reference's output."""
    g = golden(name)
    E = build(g)
    G = orc.ell_graph_as_csr(E)
    assert np.array_equal(orc.bfs_order(G), g["ref_bfs_p"])
    p, ptrs, nc = orc.greedy_color_ordering(G)
    assert nc == int(g["ref_num_colors"][0]) and np.array_equal(p, g["ref_color_p"]) and np.array_equal(ptrs, g["ref_color_ptrs"])
    node, val, deg = orc.ell_permuted(E, p, p)
    assert np.array_equal(node.ravel(), g["ref_perm_node"])
    assert np.array_equal(val.ravel(), g["ref_perm_val"])
    assert np.array_equal(deg, g["ref_perm_degrees"])
    B = orc.EllMatrix(E.n, E.m, E.max_d, node, val, deg)
    assert np.array_equal(B.matvec(g["x"]), g["ref_perm_y"])


@pytest.mark.parametrize("name", eig_golden_names())
def test_lanczos_and_generalized_lanczos_vs_the_reference(golden, name):
    """lanczos / generalized_lanczos (eigensolver.f90:27-155) run by the reference itself
    (oracle/ref_driver.f90, mode eig:<n>); the fixture's Q(:,1) is the time-seeded start vector
    of that run and is fed back as the oracle's.  B%solve in the reference run = cg(1e-14)."""
    g = golden(name)
    n, ns = int(g["n"]), int(g["nsteps"])
    A = orc.CsrMatrix.from_edges(n, n, g["ei"], g["ej"], g["ev"])
    assert np.array_equal(A.val, g["ref_val"])
    T, Q = g["ref_lanczos_T"].reshape(ns, 3).T, g["ref_lanczos_Q"].reshape(ns, n).T
    To, Qo = orc.lanczos(A, ns, Q[:, 0].copy())
    assert np.abs(To - T).max() <= 1e-12 and np.abs(Qo - Q).max() <= 1e-12
    assert np.array_equal(T[0], T[2]) and T[0, -1] == 0.0
    B = orc.CsrMatrix(n, n, A.ptr, A.node, g["ref_B_val"])
    T, Q = g["ref_glanczos_T"].reshape(ns, 3).T, g["ref_glanczos_Q"].reshape(ns, n).T
    To, Qo = orc.generalized_lanczos(A, B, ns, Q[:, 0].copy(), 1e-14)
    assert np.abs(To - T).max() <= 1e-11 and np.abs(Qo - Q).max() <= 1e-11
    # the vectors are B-orthonormal: Q^T B Q = I
    BQ = np.stack([B.matvec(Q[:, k].copy()) for k in range(ns)], axis=1)
    assert np.abs(Q.T @ BQ - np.eye(ns)).max() <= 1e-8


def composite_blocks(g):
    """Oracle leaves of a comp_* fixture + the block offsets (0-based)."""
    rp, cp = g["ref_comp_row_ptr"] - 1, g["ref_comp_col_ptr"] - 1
    blocks = [[orc.CsrMatrix(int(rp[i + 1] - rp[i]), int(cp[j + 1] - cp[j]), g[f"ref_blk{i + 1}{j + 1}_ptr"],
                             g[f"ref_blk{i + 1}{j + 1}_node"], g[f"ref_blk{i + 1}{j + 1}_val"]) for j in range(2)]
              for i in range(2)]
    return rp, cp, blocks


@pytest.mark.parametrize("name", comp_golden_names())
def test_composite_block_loops_bit_exact(golden, name):
    """composite_matvec_add / composite_matvec_t_add (sparse_matrix_composites.f90:1076-1127) run by the
    reference on a 2 x 2 composite of csr leaves: the block loops over oracle leaves reproduce y, y_add,
    yt, yt_add bit for bit, and get_value through the owning block gives the reference's Jacobi idiag."""
    g = golden(name)
    n = int(g["n"])
    rp, cp, B = composite_blocks(g)
    for i in range(2):
        for j in range(2):      # every block = the edges that fall into it, in the list's insertion order
            sel = (g["ei"] > rp[i]) & (g["ei"] <= rp[i + 1]) & (g["ej"] > cp[j]) & (g["ej"] <= cp[j + 1])
            L = orc.CsrMatrix.from_edges(B[i][j].n, B[i][j].m, g["ei"][sel] - rp[i], g["ej"][sel] - cp[j], g["ev"][sel])
            assert np.array_equal(L.ptr, B[i][j].ptr) and np.array_equal(L.node, B[i][j].node) and np.array_equal(L.val, B[i][j].val)

    def matvec_add(x, y):
        for i in range(2):
            for j in range(2):
                B[i][j].matvec_add(x[cp[j]:cp[j + 1]].copy(), y[rp[i]:rp[i + 1]])
        return y

    def matvec_t_add(x, y):
        for j in range(2):
            for i in range(2):
                B[i][j].matvec_t_add(x[rp[i]:rp[i + 1]].copy(), y[cp[j]:cp[j + 1]])
        return y

    y = matvec_add(g["x"], np.zeros(n))
    assert np.array_equal(y, g["ref_y"])
    assert np.array_equal(matvec_add(g["x"], y), g["ref_y_add"])
    yt = matvec_t_add(g["b"], np.zeros(n))
    assert np.array_equal(yt, g["ref_yt"])
    assert np.array_equal(matvec_t_add(g["b"], yt), g["ref_yt_add"])
    # jacobi_setup on the composite: idiag(i) = 1 / A%get_value(i,i), answered by the diagonal blocks here
    idiag = np.concatenate([orc.Jacobi(B[0][0]).idiag, orc.Jacobi(B[1][1]).idiag])
    assert np.array_equal(idiag, g["ref_s2_idiag"])
    assert np.array_equal(idiag * g["b"], g["ref_s2_pcz"])
