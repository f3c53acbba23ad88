#!/usr/bin/env python3
"""Where the default (tree-order) dots leave SURVEY 8d's "iterations +-1" gate: CG on tridiag(-1, 2, -1),
f = 2 dx^2 (config C1's system: cond ~ 0.4 n^2) over n x tolerance, default dot order against dot_order = 1 (= the reference's
count, bit for bit) -- one markdown table on stdout, pasted into INTEGRATION.md.  GPU box only."""
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sigma_amd as sg          # noqa: E402
from sigma_amd import problems as P          # noqa: E402

sg.init(0)
rows = []
for n in (127, 1000, 2000, 5000, 10000, 20000):
    edges, f, v = P.diffusion_1d(n)
    A = sg.csr_matrix.from_edges(n, n, *edges)
    cond = 4.0 / (np.pi / (n + 1)) ** 2
    for tol in (1e-8, 1e-10, 1e-12, 1e-14, 1e-16):
        its = {}
        for order in (1, 0):
            s = sg.cg(tol)
            s.set_option("dot_order", order)
            s.set_max_iter(40 * n)
            s.setup(A)
            u = np.zeros(n)
            s.solve(A, u, f, check=False)
            its[order] = (s.iterations, float(np.abs(u - v).max()))
            s.destroy()
        rows.append({"n": n, "cond": cond, "tol": tol, "cond_x_tol": cond * tol, "ref_order_iters": its[1][0], "tree_iters": its[0][0],
                     "ref_err": its[1][1], "tree_err": its[0][1]})
    A.destroy()
print("| n | cond(A) | tolerance | cond x tol | iterations, reference order (= the CPU build) | iterations, tree order | within +-1 | max error vs analytic (ref / tree) |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    ok = abs(r["ref_order_iters"] - r["tree_iters"]) <= 1
    print(f"| {r['n']} | {r['cond']:.1e} | {r['tol']:.0e} | {r['cond_x_tol']:.1e} | {r['ref_order_iters']} | {r['tree_iters']} | {'yes' if ok else '**no**'} | "
          f"{r['ref_err']:.1e} / {r['tree_err']:.1e} |")
print()
print(json.dumps(rows))
