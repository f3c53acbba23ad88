#!/usr/bin/env python3
"""Diagnose a fuzz_solvers seed: our default-order solve several times, with and without replayed groups, history tails."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import sigma_amd as sg
import oracle as orc
import fuzz_solvers as F
sg.init(0)
for seed in [int(a) for a in sys.argv[1:]]:
    rs = np.random.RandomState(seed)
    kind, n, (ptr, node, val) = F.make(rs)
    A = orc.CsrMatrix(n, n, ptr, node, val)
    nparts = int(rs.choice([1, 1, 1, 2, 3, 5])) if n >= 64 else 1
    pck = ["none", "jacobi", "ildu", "ildu"][int(rs.randint(0, 4))]
    solver = ["cg", "cg", "bicgstab", "gmres"][int(rs.randint(0, 4))]
    b = rs.standard_normal(n)
    cuts = np.sort(rs.choice(np.arange(1, n // 2), size=nparts - 1, replace=False)) * 2
    starts = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    print(seed, kind, n, nparts, pck, solver, starts)
    opc = orc.Jacobi(A) if pck == "jacobi" else orc.Ildu(F.block_diagonal(A, starts))
    ur, itr, res, hist = orc.bicgstab(A, b, tol=1e-8, pc=opc, max_iter=600, history=600)
    print(" oracle", itr, hist[max(0, itr - 6):itr])
    for parts in (nparts, 1):
        for graph in (1, 0):
            for rep in range(2):
                H = sg.partitioned_csr_matrix(n, n, ptr, node, val, starts) if parts > 1 else sg.csr_matrix(n, n, ptr, node, val)
                pc = sg.jacobi() if pck == "jacobi" else sg.ldu()
                pc.setup(H)
                s = sg.bicgstab(1e-8)
                s.set_option("krylov_graph", graph)
                s.set_max_iter(600)
                s.set_history(700)
                s.setup(H)
                u = np.zeros(n)
                s.solve(H, u, b, pc, check=False)
                h = s.history.copy()
                it = s.last_iterations
                firstnan = int(np.argmax(~np.isfinite(h))) if (~np.isfinite(h)).any() else -1
                print(f" parts={parts} graph={graph} rep={rep}: it={it} nan_in_u={int(np.isnan(u).sum())} first_nonfinite_hist={firstnan} tail={h[max(0, it - 5):it]}")
                if firstnan >= 0:
                    print("   around:", h[max(0, firstnan - 6):firstnan + 2])
                m = min(len(h), itr, it) - 1
                rel = np.abs(h[:m] - hist[:m]) / hist[:m]
                print("   rel diff of res2 history vs oracle at its 0,5,10,...:", " ".join(f"{v:.1e}" for v in rel[::5][:30]))
                print("   res2 every 10:", " ".join(f"{v:.1e}" for v in h[:it:10]))
                s.destroy(); pc.destroy(); H.destroy()
