#!/usr/bin/env python3
"""CG time per iteration on small grids (launch-bound regime): 2-D Poisson nx^2, fixed iteration count."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0); sg.use_torch_stream()
dev = torch.device("cuda", 0)
for nx in (32, 100, 316, 1000, 2000):
    n = nx * nx
    ptr, node, val = P.poisson2d_csr(nx, nx)
    A = sg.csr_matrix(n, n, torch.from_numpy(ptr).to(dev), torch.from_numpy(node).to(dev), torch.from_numpy(val).to(dev))
    b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
    for kind in ("cg", "bicgstab"):
        s = sg.cg(1e-300) if kind == "cg" else sg.bicgstab(1e-300)
        s.setup(A)
        iters = 2000 if kind == "cg" else 1000
        s.set_max_iter(iters)
        u = torch.zeros(n, dtype=torch.float64, device=dev)
        s.solve(A, u, b, check=False)          # warm-up
        torch.cuda.synchronize()
        u.zero_()
        t0 = time.perf_counter()
        s.solve(A, u, b, check=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"nx": nx, "n": n, "solver": kind, "iterations": s.last_iterations, "us_per_iter": round(dt / max(1, s.last_iterations) * 1e6, 2)}), flush=True)
