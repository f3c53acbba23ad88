#!/usr/bin/env python3
"""One-off stress (not collected by pytest): CG on many small seeded SPD systems -- single workgroup, launch loop,
oracle -- to see how often the iteration counts differ by more than one.  python tests/stress_cg_small.py [trials]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import sigma_amd as sg
import oracle as orc
from sigma_amd import problems as P
sg.init(0)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    worst = {"small": 0, "loop": 0}
    over = {"small": 0, "loop": 0}
    for t in range(trials):
        rs = np.random.RandomState(500 + t)
        if t % 3 == 0:
            nx, ny = int(rs.randint(3, 90)), int(rs.randint(3, 90))
            n = nx * ny
            ptr, node, val = P.poisson2d_csr(nx, ny)
        elif t % 3 == 1:
            n = int(rs.randint(10, 4000))
            S = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n) + rs.rand(n) * 0.1, -np.ones(n - 1)], [-1, 0, 1]).tocsr()
            ptr, node, val = (S.indptr + 1).astype(np.int32), (S.indices + 1).astype(np.int32), S.data.copy()
        else:
            n = int(rs.randint(50, 4000))
            B = sp.random(n, n, density=min(0.5, 6.0 / n), random_state=rs, format="csr")
            S = (B + B.T).tocsr()
            S = (S + sp.diags(np.abs(S).sum(axis=1).A1 * rs.uniform(1.0, 1.3) + 0.01)).tocsr()
            S.sort_indices()
            ptr, node, val = (S.indptr + 1).astype(np.int32), (S.indices + 1).astype(np.int32), S.data.copy()
        A = orc.CsrMatrix(n, n, ptr, node, val)
        H = sg.csr_matrix(n, n, ptr, node, val)
        b = rs.standard_normal(n)
        tol = 10.0 ** rs.uniform(-13, -8)
        jac = bool(t % 2)
        ur, itr, _, _ = orc.cg(A, b, tol=tol, pc=orc.Jacobi(A) if jac else None)
        for key, small in (("small", 1), ("loop", 0)):
            sg.set_option("cg_small", small)
            pc = None
            if jac:
                pc = sg.jacobi(); pc.setup(H)
            sv = sg.cg(tol); sv.setup(H)
            u = np.zeros(n); sv.solve(H, u, b, pc)
            d = abs(sv.iterations - itr)
            worst[key] = max(worst[key], d)
            if d > 1:
                over[key] += 1
                print("count differs", key, t, n, "jacobi" if jac else "plain", "tol %.1e" % tol, sv.iterations, itr)
        sg.set_option("cg_small", 1)
        H.destroy()
    print(json.dumps({"trials": trials, "worst_iteration_difference": worst, "solves_more_than_one_off": over}))


if __name__ == "__main__":
    main()
