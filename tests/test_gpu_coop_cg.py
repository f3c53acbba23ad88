"""k_cg_coop: CG on a mid-sized system as ONE launch of up to 256 co-resident workgroups (sgm_cg.hip, sgm_coop.hpp; VERDICT r03
item 6).  Same statements as the launch loop and the reference's cg_solve; only the dot products' summation order differs,
so the gates are the launch loop's: iteration count within +-1 of the oracle's and of the launch loop's, solutions within
max(1e-12, kappa * tol)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import sigma_amd as sg
from sigma_amd import problems as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def orc():
    import oracle
    return oracle


@pytest.fixture(scope="module", autouse=True)
def _init():
    sg.init(0)


def _cases():
    yield "2-D 5-point 150 x 131 (n = 19650: two workgroups, halo 150)", 150 * 131, P.poisson2d_csr(150, 131)
    yield "2-D 5-point 320 x 317 (n = 101440)", 320 * 317, P.poisson2d_csr(320, 317)
    yield "3-D 7-point 40 x 37 x 33 (halo = a plane of 1480 rows)", 40 * 37 * 33, P.laplace3d_csr(40, 37, 33)
    yield "1-D tridiagonal n = 50001 (odd: a scalar tail, halo 2)", 50001, P.tridiag_csr(50001, 2.5, -1.0, -1.0)
    yield "2-D 5-point 300 x 300 (n = 90000: three rows per thread on one XCD)", 300 * 300, P.poisson2d_csr(300, 300)
    yield "2-D 5-point 520 x 410 (n = 213200: beyond one XCD, the all-CU variant)", 520 * 410, P.poisson2d_csr(520, 410)
    yield "3-D 7-point 70 x 66 x 30 (halo = a plane of 4620 rows, wider than a workgroup's 1024)", 70 * 66 * 30, P.laplace3d_csr(70, 66, 30)


@pytest.mark.parametrize("jac", [False, True])
def test_cooperative_cg_vs_oracle_and_launch_loop(orc, jac):
    for label, n, (ptr, node, val) in _cases():
        val = val * (1.0 + 0.1 * np.cos(np.arange(val.size) * 0.37)) if "5-point 150" in label else val
        if "5-point 150" in label:                     # keep it symmetric: average with the transpose entry by entry
            import scipy.sparse as sp
            S = sp.csr_matrix((val, node - 1, ptr - 1), shape=(n, n))
            S = ((S + S.T) * 0.5).tocsr()
            S.sort_indices()
            S2 = sp.csr_matrix((np.ones_like(val), node - 1, ptr - 1), shape=(n, n))
            assert (S2 != (S != 0)).nnz == 0
            # same pattern in the same stored order (columns were ascending already except the reference's insertion order)
            lookup = {(i, j): v for i, j, v in zip(*sp.find(S))}
            rows = np.repeat(np.arange(n), np.diff(ptr))
            val = np.array([lookup[(i, j - 1)] for i, j in zip(rows, node)])
        A = orc.CsrMatrix(n, n, ptr, node, val)
        H = sg.csr_matrix(n, n, ptr, node, val)
        assert H.kernel.startswith("k_csr_sl<"), (label, H.kernel)
        b = np.sin(0.01 * np.arange(1, n + 1)) + 0.5
        tol = 1e-9
        pco = orc.Jacobi(A) if jac else None
        ur, itr, _, hr = orc.cg(A, b, tol=tol, pc=pco, history=100000)
        out = {}
        for mode in ("coop", "loop"):
            pc = None
            if jac:
                pc = sg.jacobi(); pc.setup(H)
            s = sg.cg(tol)
            s.set_history(100000)
            if mode == "loop":
                s.set_option("cg_small", 0)
            s.setup(H)
            u = np.full(n, 0.25)                        # a non-zero initial guess
            ur0, itr0, _, _ = orc.cg(A, b, x0=np.full(n, 0.25), tol=tol, pc=pco)
            s.solve(H, u, b, pc)
            out[mode] = (u, s.iterations, np.array(s.history), s.converged, s.res2)
            assert s.converged and np.sqrt(s.res2) <= tol, (label, mode)
            assert abs(s.iterations - itr0) <= 1, (label, mode, s.iterations, itr0)
            assert np.abs(u - ur0).max() <= 1e-9 * max(1.0, np.abs(ur0).max()), (label, mode)
        assert abs(out["coop"][1] - out["loop"][1]) <= 1, label
        k = min(len(out["coop"][2]), len(out["loop"][2]), 50)
        assert np.abs(out["coop"][2][:k] - out["loop"][2][:k]).max() <= 1e-10 * out["loop"][2][:k].max(), label


def test_cooperative_cg_cut_into_launches_and_capped(orc):
    """A launch runs at most `cg_small` = n iterations and hands r, p, res2 to the next one through the solver's work
    vectors -- bit-identical to the uncut solve; set_max_iter stops it where the launch loop stops."""
    nx, ny = 200, 160
    n = nx * ny
    ptr, node, val = P.poisson2d_csr(nx, ny)
    H = sg.csr_matrix(n, n, ptr, node, val)
    b = np.full(n, 1.0 / n)
    res = {}
    for chunk in (1, 7, 64):
        s = sg.cg(1e-10)
        s.set_history(10000)
        s.set_option("cg_small", chunk)
        s.setup(H)
        u = np.zeros(n)
        s.solve(H, u, b)
        res[chunk] = (u, s.iterations, np.array(s.history))
    for chunk in (7, 64):
        assert res[chunk][1] == res[1][1] and np.array_equal(res[chunk][0], res[1][0]) and np.array_equal(res[chunk][2], res[1][2])
    s = sg.cg(1e-300)
    s.set_max_iter(37)
    s.setup(H)
    u = np.zeros(n)
    s.solve(H, u, b, check=False)
    assert s.last_iterations == 37 and not s.converged
    # `iterations` accumulates across solves like the reference's (cg_solvers.f90:72,145)
    s.solve(H, u, b, check=False)
    assert s.iterations == 74


def test_cooperative_cg_gives_up_loudly_and_the_launch_loop_takes_over(tmp_path):
    """Every wait is bounded.  With a spin limit of 1 (solver option coop_spin_limit) a hand-off gives up at once: nothing of x has been written, the solver says so on stderr, retires the
    cooperative kernel for that handle and runs the launch loop -- same answer."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, sigma_amd as sg\n"
            "from sigma_amd import problems as P\n"
            "sg.init(0)\n"
            "n = 200 * 160\n"
            "ptr, node, val = P.poisson2d_csr(200, 160)\n"
            "A = sg.csr_matrix(n, n, ptr, node, val)\n"
            "b = np.full(n, 1.0 / n)\n"
            "s = sg.cg(1e-10); s.set_option('coop_spin_limit', 1); s.setup(A)\n"
            "u = np.zeros(n); s.solve(A, u, b)\n"
            "s2 = sg.cg(1e-10); s2.set_option('cg_small', 0); s2.setup(A)\n"
            "u2 = np.zeros(n); s2.solve(A, u2, b)\n"
            "print('ITS', s.iterations, s2.iterations, bool(np.array_equal(u, u2)))\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    its = [ln for ln in p.stdout.splitlines() if ln.startswith("ITS")][0].split()
    assert its[1] == its[2] and its[3] == "True", p.stdout          # the fallback IS the launch loop: bit-identical to it
    assert "cooperative CG gave up waiting" in p.stderr


def test_one_xcd_variant_runs_where_it_fits_and_equals_the_all_cu_variant_bit_for_bit():
    """Systems of up to 32 workgroups' rows run on the CUs of ONE XCD (hand-offs through that XCD's L2) once the participants
    have proved their co-location; the same rows per workgroup on all CUs (solver option cg_coop_variant = 16) sum the same partials in the
    same order: bit-identical solutions and histories.  SGM_TRACE names the variant that ran."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, sigma_amd as sg, hashlib\n"
            "from sigma_amd import problems as P\n"
            "sg.init(0)\n"
            "for nx, ny in ((150, 131), (250, 240), (300, 300), (330, 330), (520, 410)):\n"
            "    n = nx * ny\n"
            "    ptr, node, val = P.poisson2d_csr(nx, ny)\n"
            "    A = sg.csr_matrix(n, n, ptr, node, val)\n"
            "    b = np.cos(0.003 * np.arange(n))\n"
            "    s = sg.cg(1e-10); s.set_option('cg_coop_variant', int(sys.argv[1])); s.set_history(100000); s.setup(A)\n"
            "    u = np.zeros(n); s.solve(A, u, b)\n"
            "    print('SOLVE', n, s.iterations, hashlib.sha1(u.tobytes() + np.array(s.history).tobytes()).hexdigest())\n" % ROOT)
    runs = {}
    for xcd in ("1", "0"):          # option cg_coop_variant: + 16 = never the one-XCD variant
        env = dict(os.environ, SGM_TRACE="1")
        p = subprocess.run([sys.executable, "-c", code, "0" if xcd == "1" else "16"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        runs[xcd] = ([ln for ln in p.stdout.splitlines() if ln.startswith("SOLVE")],
                     [ln for ln in p.stderr.splitlines() if "cooperative launch" in ln])
    on, off = runs["1"], runs["0"]
    assert len(on[0]) == 5 and len(on[1]) == 5 and len(off[1]) == 5, (on, off)
    assert ["one XCD" in ln for ln in on[1]] == [True, True, True, False, False], on[1]      # (330^2 > 3 x 32 x 1024 rows: all CUs)
    assert not any("one XCD" in ln for ln in off[1]), off[1]
    # rows per workgroup: the one-XCD variant takes 1, 2, 3 rows per thread; the all-CU variant 1 (<= 256 workgroups) --
    # different partitions of the dot products, so only the LAST TWO cases (all CUs either way) must agree bit for bit here ...
    assert on[0][3:] == off[0][3:]
    # ... and with the rows per thread pinned the two variants are the same arithmetic everywhere
    runs2 = {}
    for xcd in ("1", "0"):          # (low four bits: rows per thread pinned to 4)
        p = subprocess.run([sys.executable, "-c", code, "4" if xcd == "1" else "20"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        runs2[xcd] = [ln for ln in p.stdout.splitlines() if ln.startswith("SOLVE")]
    assert runs2["1"] == runs2["0"] and len(runs2["1"]) == 5, runs2
    # iteration counts of the default selection within +-1 of the pinned one (same statements, other summation order)
    for a, b in zip(on[0], runs2["1"]):
        assert abs(int(a.split()[2]) - int(b.split()[2])) <= 1, (a, b)


def test_wave_sum_is_the_shuffle_butterfly_bit_for_bit():
    """Every block sum of the library goes through wave_sum (sgm_internal.hpp): v_permlane32/16_swap + DPP row rotations in
    place of six ds_bpermute round trips.  The stand-alone check runs both on 524,288 random doubles (mixed magnitudes and
    signs, zeros, an infinity, a NaN) and compares every lane's bits, and block_sum<1024> with the wave sums added in order."""
    exe = os.path.join(ROOT, "tools", "probes", "wave_sum_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-I", os.path.join(ROOT, "sigma_amd", "csrc"),
                               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "probes", "wave_sum_probe.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "0 of 524288 lanes differ" in p.stdout and "block_sum<1024>: 0 differ" in p.stdout, p.stdout + p.stderr


def _nonsymmetric(ptr, node, val):
    """the 5- / 7-point stencil with an upwind-like skew -- entries right of the diagonal x 0.8, left of it x 1.2 -- and the
    diagonal x 1.05 (strictly dominant: without it the interior rows sum to zero and BiCGStab breaks down in every implementation)"""
    rows = np.repeat(np.arange(1, len(ptr)), np.diff(ptr))
    return val * (1.0 + 0.2 * np.sign(rows - node)) * np.where(rows == node, 1.05, 1.0)


@pytest.mark.parametrize("jac", [False, True])
def test_cooperative_bicgstab_vs_oracle_and_launch_loop(orc, jac):
    """k_bicg_coop: BiCGStab as one cooperative launch (five hand-offs per iteration, two of them carrying two scalars).
    Same statements as bicgstab_solve(_pc) and as the launch loop; the dots' summation order differs, so the gates are the
    launch loop's: iteration counts within max(3, 10 %) of the oracle's, true residual at the tolerance, solutions within 1e-7."""
    cases = [("2-D 150 x 131 (one XCD, one row per thread)", 150 * 131, P.poisson2d_csr(150, 131)),
             ("2-D 256 x 250 (all CUs, one row per thread)", 256 * 250, P.poisson2d_csr(256, 250)),
             ("2-D 600 x 500 (all CUs, two rows per thread)", 600 * 500, P.poisson2d_csr(600, 500)),
             ("2-D 500 x 450 (all CUs)", 500 * 450, P.poisson2d_csr(500, 450)),
             ("2-D 760 x 700 (all CUs, four rows per thread)", 760 * 700, P.poisson2d_csr(760, 700)),
             ("3-D 40 x 37 x 33", 40 * 37 * 33, P.laplace3d_csr(40, 37, 33)),
             ("1-D n = 50001", 50001, P.tridiag_csr(50001, 2.5, -1.0, -1.0))]
    for label, n, (ptr, node, val) in cases:
        val = _nonsymmetric(ptr, node, val)
        A = orc.CsrMatrix(n, n, ptr, node, val)
        H = sg.csr_matrix(n, n, ptr, node, val)
        b = np.sin(0.01 * np.arange(1, n + 1)) + 0.5
        tol = 1e-9
        pco = orc.Jacobi(A) if jac else None
        x0 = np.full(n, 0.25)
        ur, itr = orc.bicgstab(A, b, x0=x0.copy(), tol=tol, pc=pco)[:2]
        out = {}
        for mode in ("coop", "loop"):
            pc = None
            if jac:
                pc = sg.jacobi(); pc.setup(H)
            s = sg.bicgstab(tol)
            s.set_history(100000)
            if mode == "loop":
                s.set_option("bicgstab_small", 0)
            s.setup(H)
            u = x0.copy()
            s.solve(H, u, b, pc)
            Au = np.zeros(n); H.matvec(u, Au)
            out[mode] = (u, s.iterations, np.array(s.history))
            assert s.converged and np.sqrt(s.res2) <= tol, (label, mode)
            assert abs(s.iterations - itr) <= max(3, itr // 10), (label, mode, s.iterations, itr)
            assert np.abs(Au - b).max() <= 1e-7 * np.abs(b).max(), (label, mode)
            assert np.abs(u - ur).max() <= 1e-7 * np.abs(ur).max(), (label, mode)
        k = min(len(out["coop"][2]), len(out["loop"][2]), 20)
        assert np.abs(out["coop"][2][:k] - out["loop"][2][:k]).max() <= 1e-8 * out["loop"][2][:k].max(), label


def test_cooperative_bicgstab_cut_into_launches_capped_and_falling_back(tmp_path):
    """Launches of at most `bicgstab_small` iterations hand r, r0, p, v and the four scalars on: bit-identical to the uncut solve;
    set_max_iter stops it; with a spin limit of 1 the hand-offs give up at once and the launch loop returns the same answer;
    SGM_TRACE names the variant."""
    nx, ny = 200, 160
    n = nx * ny
    ptr, node, val = P.poisson2d_csr(nx, ny)
    val = _nonsymmetric(ptr, node, val)
    H = sg.csr_matrix(n, n, ptr, node, val)
    b = np.full(n, 1.0 / n)
    res = {}
    for chunk in (1, 7, 64):
        s = sg.bicgstab(1e-10)
        s.set_history(10000)
        s.set_option("bicgstab_small", chunk)
        s.setup(H)
        u = np.zeros(n)
        s.solve(H, u, b)
        res[chunk] = (u, s.iterations, np.array(s.history))
    for chunk in (7, 64):
        assert res[chunk][1] == res[1][1] and np.array_equal(res[chunk][0], res[1][0]) and np.array_equal(res[chunk][2], res[1][2])
    s = sg.bicgstab(1e-300)
    s.set_max_iter(23)
    s.setup(H)
    u = np.zeros(n)
    s.solve(H, u, b, check=False)
    assert s.last_iterations == 23 and not s.converged
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, sigma_amd as sg\n"
            "from sigma_amd import problems as P\n"
            "sg.init(0)\n"
            "n = 200 * 160\n"
            "ptr, node, val = P.poisson2d_csr(200, 160)\n"
            "rows = np.repeat(np.arange(1, n + 1), np.diff(ptr)); val = val * (1.0 + 0.2 * np.sign(rows - node)) * np.where(rows == node, 1.05, 1.0)\n"
            "A = sg.csr_matrix(n, n, ptr, node, val)\n"
            "b = np.full(n, 1.0 / n)\n"
            "s = sg.bicgstab(1e-10); s.set_option('coop_spin_limit', int(sys.argv[1])); s.setup(A)\n"
            "u = np.zeros(n); s.solve(A, u, b)\n"
            "s2 = sg.bicgstab(1e-10); s2.set_option('bicgstab_small', 0); s2.setup(A)\n"
            "u2 = np.zeros(n); s2.solve(A, u2, b)\n"
            "print('ITS', s.iterations, s2.iterations, bool(np.array_equal(u, u2)))\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code, "1"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    its = [ln for ln in p.stdout.splitlines() if ln.startswith("ITS")][0].split()
    assert its[1] == its[2] and its[3] == "True", p.stdout
    assert "cooperative BiCGStab gave up waiting" in p.stderr
    p = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=600, env=dict(os.environ, SGM_TRACE="1"))
    assert p.returncode == 0 and "bicgstab: one cooperative launch" in p.stderr and "on one XCD" in p.stderr, p.stderr[-1000:]


def test_cooperative_cg_eight_rows_per_thread_with_r_in_lds_vs_launch_loop():
    """Between 1,048,576 and 2,097,152 rows the cooperative CG kernel keeps r in LDS beside p (eight rows per thread): against
    the launch loop on a 1300 x 1100 grid with variable coefficients (too large for the oracle's single thread in a test):
    iterations within one, solutions within 1e-10, true residual at the tolerance; SGM_TRACE names 8192 rows per workgroup."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, sigma_amd as sg\n"
            "from sigma_amd import problems as P\n"
            "sg.init(0)\n"
            "nx, ny = 1300, 1100\n"
            "n = nx * ny\n"
            "ptr, node, val = P.poisson2d_csr(nx, ny)\n"
            "rows = np.repeat(np.arange(1, n + 1), np.diff(ptr))\n"
            "val = val * (1.0 + 0.1 * np.cos(0.37 * (rows + node))) * np.where(rows == node, 1.1, 1.0)\n"
            "A = sg.csr_matrix(n, n, ptr, node, val)\n"
            "b = np.sin(0.01 * np.arange(1, n + 1)) + 0.5\n"
            "out = []\n"
            "for jac in (False, True):\n"
            "    for small in (1, 0):\n"
            "        pc = None\n"
            "        if jac:\n"
            "            pc = sg.jacobi(); pc.setup(A)\n"
            "        s = sg.cg(1e-9); s.set_option('cg_small', small); s.setup(A)\n"
            "        u = np.full(n, 0.2); s.solve(A, u, b, pc)\n"
            "        Au = np.zeros(n); A.matvec(u, Au)\n"
            "        out.append((u, s.iterations, float(np.abs(Au - b).max())))\n"
            "    (uc, ic, rc), (ul, il, rl) = out[-2], out[-1]\n"
            "    print('PAIR', int(jac), ic, il, float(np.abs(uc - ul).max() / np.abs(ul).max()), rc, rl)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, SGM_TRACE="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    pairs = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("PAIR")]
    assert len(pairs) == 2, p.stdout
    for _, jac, ic, il, du, rc, rl in pairs:
        assert abs(int(ic) - int(il)) <= 1 and float(du) <= 1e-10 and float(rc) <= 1e-7 and float(rl) <= 1e-7, (jac, ic, il, du, rc, rl)
    assert "x 8192 rows" in p.stderr, p.stderr[-1500:]


def test_cooperative_kernels_take_structured_ellpack_matrices(orc):
    """An ELLPACK matrix with <= 8 slots from <= 15 offsets lives in the same sliced form as a CSR one (every slot an entry,
    padding = 0.0 x the last neighbour, like the reference's ellpack_matvec_add): the cooperative CG / BiCGStab kernels
    take it -- against the oracle's solvers on the oracle's ELLPACK restatement, and SGM_TRACE says so."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, sigma_amd as sg, oracle as orc\n"
            "from sigma_amd import problems as P\n"
            "sg.init(0)\n"
            "nx, ny = 210, 190\n"
            "n = nx * ny\n"
            "ei, ej, ev = P.poisson2d_edges(nx, ny)\n"
            "ev = np.where(ei == ej, ev * 1.05, ev * (1.0 + 0.2 * np.sign(ei - ej)))\n"
            "As = orc.EllMatrix.from_edges(n, n, ei, ej, np.where(ei == ej, 4.2, -1.0))\n"
            "An = orc.EllMatrix.from_edges(n, n, ei, ej, ev)\n"
            "b = np.sin(0.01 * np.arange(1, n + 1)) + 0.5\n"
            "for name, A, solver, ref in (('cg', As, sg.cg, orc.cg), ('bicgstab', An, sg.bicgstab, orc.bicgstab)):\n"
            "    H = sg.ellpack_matrix(n, n, A.node, A.val)\n"
            "    ur, itr = ref(A, b, x0=np.full(n, 0.25), tol=1e-9)[:2]\n"
            "    s = solver(1e-9); s.setup(H)\n"
            "    u = np.full(n, 0.25); s.solve(H, u, b)\n"
            "    print('RES', name, s.iterations, itr, float(np.abs(u - ur).max() / np.abs(ur).max()))\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, SGM_TRACE="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    res = {ln.split()[1]: ln.split()[2:] for ln in p.stdout.splitlines() if ln.startswith("RES")}
    assert abs(int(res["cg"][0]) - int(res["cg"][1])) <= 1 and float(res["cg"][2]) <= 1e-9, res
    assert abs(int(res["bicgstab"][0]) - int(res["bicgstab"][1])) <= max(3, int(res["bicgstab"][1]) // 10) and float(res["bicgstab"][2]) <= 1e-7, res
    assert "cg: one cooperative launch" in p.stderr and "bicgstab: one cooperative launch" in p.stderr, p.stderr[-1500:]


@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
def test_one_solver_handle_serves_matrices_of_one_size_and_different_stencils(orc, kind):
    """The halo window of the cooperative kernels is the reach of the dictionary of the matrix being solved (kept on the
    matrix part where the dictionary is built), not of the first matrix a solver handle met: a handle set up on a
    tridiagonal matrix (reach 1) then solves a 100 x 100 grid (reach 100) of the same n -- with and without a second
    setup, the reference only asks for matching sizes (cg_solvers.f90:52-90) -- and the other way round."""
    n = 10000
    mats = {"tridiagonal": P.tridiag_csr(n, 2.5, -1.0, -1.0), "grid 100 x 100": P.poisson2d_csr(100, 100)}
    if kind == "bicgstab":
        mats = {k: (p, c, _nonsymmetric(p, c, v)) for k, (p, c, v) in mats.items()}
    b = np.sin(0.01 * np.arange(1, n + 1)) + 0.5
    ref = orc.cg if kind == "cg" else orc.bicgstab
    for order in (("tridiagonal", "grid 100 x 100"), ("grid 100 x 100", "tridiagonal")):
        for again in (False, True):
            s = (sg.cg if kind == "cg" else sg.bicgstab)(1e-9)
            first = True
            for name in order + order[:1]:
                ptr, node, val = mats[name]
                H = sg.csr_matrix(n, n, ptr, node, val)
                assert H.kernel.startswith("k_csr_sl<"), (name, H.kernel)
                if first or again:
                    s.setup(H)
                first = False
                u = np.zeros(n)
                s.solve(H, u, b)
                ur, itr = ref(orc.CsrMatrix(n, n, ptr, node, val), b, tol=1e-9)[:2]
                slack = 1 if kind == "cg" else max(3, itr // 10)
                assert s.converged and abs(s.last_iterations - itr) <= slack, (kind, order, again, name, s.last_iterations, itr)
                assert np.abs(u - ur).max() <= 1e-8 * np.abs(ur).max(), (kind, order, again, name)
