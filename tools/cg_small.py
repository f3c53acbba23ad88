#!/usr/bin/env python3
"""CG time per iteration on small grids (launch-bound regime): 2-D Poisson nx^2, fixed iteration count."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0); sg.use_torch_stream()
dev = torch.device("cuda", 0)
GRAPH = [int(v) for v in os.environ.get("KRYLOV_GRAPH", "1,0").split(",")]
NXS = [int(v) for v in os.environ.get("NXS", "32,100,316,1000,2000").split(",")]
SOLVERS = os.environ.get("SOLVERS", "cg,bicgstab").split(",")
LAUNCH_LOOP = bool(os.environ.get("LAUNCH_LOOP"))      # solver options cg_small / bicgstab_small = 0: the launch loops alone
for nx in NXS:
    n = nx * nx
    ptr, node, val = P.poisson2d_csr(nx, nx)
    A = sg.csr_matrix(n, n, torch.from_numpy(ptr).to(dev), torch.from_numpy(node).to(dev), torch.from_numpy(val).to(dev))
    b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
    for kind, graph in [(k, g) for k in SOLVERS for g in GRAPH]:
        sg.set_option("krylov_graph", graph)
        s = sg.cg(1e-300) if kind == "cg" else sg.bicgstab(1e-300)
        if LAUNCH_LOOP:
            s.set_option("cg_small" if kind == "cg" else "bicgstab_small", 0)
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
        print(json.dumps({"nx": nx, "n": n, "solver": kind, "krylov_graph": graph, "iterations": s.last_iterations, "us_per_iter": round(dt / max(1, s.last_iterations) * 1e6, 2)}), flush=True)
# C1 of BASELINE.json: tridiagonal (-1, 2, -1), n = 10,000, CG
if os.environ.get("NO_C1"):
    sys.exit(0)
n = 10000
ptr = np.zeros(n + 1, np.int32); deg = np.full(n, 3); deg[0] = deg[-1] = 2
ptr[0] = 1; ptr[1:] = 1 + np.cumsum(deg)
rows = np.repeat(np.arange(n), deg)
cols = np.concatenate([[0, 1]] + [[i - 1, i, i + 1] for i in range(1, n - 1)] + [[n - 2, n - 1]]).astype(np.int64)
vals = np.where(cols == rows, 2.0, -1.0)
A = sg.csr_matrix(n, n, torch.from_numpy(ptr).to(dev), torch.from_numpy((cols + 1).astype(np.int32)).to(dev), torch.from_numpy(vals).to(dev))
dx = 1.0 / (n + 1)
b = torch.full((n,), 2 * dx * dx, dtype=torch.float64, device=dev)
for small in (1, 0):
    sg.set_option("cg_small", small)
    s = sg.cg(1e-16); s.setup(A)
    u = torch.zeros(n, dtype=torch.float64, device=dev)
    s.solve(A, u, b); torch.cuda.synchronize()
    u.zero_()
    t0 = time.perf_counter(); s.solve(A, u, b); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    i = torch.arange(1, n + 1, dtype=torch.float64, device=dev) * dx
    err = float((u - i * (1 - i)).abs().max())
    print(json.dumps({"C1": "tridiagonal n=10000 CG tol 1e-16", "cg_small": small, "iterations": s.last_iterations, "solve_ms": round(dt * 1e3, 2),
                      "us_per_iter": round(dt / s.last_iterations * 1e6, 2), "max_err_vs_analytic": err}), flush=True)
