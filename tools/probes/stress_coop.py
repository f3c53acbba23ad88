#!/usr/bin/env python3
"""One-off stress (not part of the suite): the cooperative CG / BiCGStab kernels against the launch loop on seeded random
grids -- 2-D and 3-D shapes, variable coefficients (symmetric for CG, skewed for BiCGStab), plain and Jacobi, random
iteration chunks: iteration counts within one (CG) / max(3, 10 %) (BiCGStab), solutions within 1e-8, true residuals small."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)


def case(rs):
    kind = rs.choice(["2d", "2d", "3d", "1d"])
    if kind == "2d":
        nx, ny = int(rs.randint(40, 900)), int(rs.randint(40, 900))
        while nx * ny > 1_000_000 or nx * ny < 2100:
            nx, ny = int(rs.randint(40, 900)), int(rs.randint(40, 900))
        return f"2d {nx}x{ny}", nx * ny, P.poisson2d_csr(nx, ny)
    if kind == "3d":
        nx, ny, nz = int(rs.randint(8, 90)), int(rs.randint(8, 90)), int(rs.randint(8, 90))
        while nx * ny * nz > 700_000 or nx * ny * nz < 2100:
            nx, ny, nz = int(rs.randint(8, 90)), int(rs.randint(8, 90)), int(rs.randint(8, 90))
        return f"3d {nx}x{ny}x{nz}", nx * ny * nz, P.laplace3d_csr(nx, ny, nz)
    n = int(rs.randint(2100, 900_000))
    return f"1d {n}", n, P.tridiag_csr(n, 2.2, -1.0, -1.0)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = []
    for t in range(trials):
        rs = np.random.RandomState(9000 + t)
        label, n, (ptr, node, val) = case(rs)
        rows = np.repeat(np.arange(1, n + 1), np.diff(ptr))
        sym = val * (1.0 + 0.1 * np.cos(0.37 * (rows + node))) * np.where(rows == node, 1.25, 1.0)      # (strictly dominant: SPD)
        skew = sym * (1.0 + 0.15 * np.sign(rows - node)) * np.where(rows == node, 1.03, 1.0)
        b = np.sin(0.013 * np.arange(1, n + 1)) + 0.3
        for solver, v in (("cg", sym), ("bicgstab", skew)):
            A = sg.csr_matrix(n, n, ptr, node, v)
            for jac in (False, True):
                out = {}
                for mode in ("coop", "loop"):
                    pc = None
                    if jac:
                        pc = sg.jacobi(); pc.setup(A)
                    s = getattr(sg, solver)(1e-9)
                    opt = "cg_small" if solver == "cg" else "bicgstab_small"
                    s.set_option(opt, 0 if mode == "loop" else int(rs.choice([1, 1, 37, 200])))
                    s.setup(A)
                    u = np.full(n, 0.1)
                    s.solve(A, u, b, pc)
                    Au = np.zeros(n); A.matvec(u, Au)
                    out[mode] = (u, s.iterations, float(np.abs(Au - b).max()))
                    if pc is not None:
                        pc.destroy()
                (uc, ic, rc), (ul, il, rl) = out["coop"], out["loop"]
                tol_it = 1 if solver == "cg" else max(3, il // 10)
                if abs(ic - il) > tol_it or np.abs(uc - ul).max() > 1e-8 * np.abs(ul).max() or rc > 1e-6 or not np.isfinite(rc):
                    bad.append((t, label, solver, int(jac), ic, il, float(np.abs(uc - ul).max()), rc))
            A.destroy()
        print(t, label, "ok" if not bad or bad[-1][0] != t else "MISMATCH", flush=True)
    print(json.dumps({"trials": trials, "mismatches": len(bad), "first": [str(x) for x in bad[:6]]}))


if __name__ == "__main__":
    main()
