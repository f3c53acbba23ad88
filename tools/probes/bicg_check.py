#!/usr/bin/env python3
"""BiCGStab on skewed 5-point grids: iterations and TRUE residual per size (variant chosen by the environment:
SGM_CG_COOP=0 launch loop, SGM_CG_COOP_XCD=0 all CUs, SGM_CG_COOP_STREAM=1 streamed matrix, SGM_CG_COOP_RMAX=n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
for nx, ny in [(int(a), int(b)) for a, b in (t.split("x") for t in os.environ.get("GRIDS", "150x131,256x250,400x300,760x700").split(","))]:
    n = nx * ny
    ptr, node, val = P.poisson2d_csr(nx, ny)
    rows = np.repeat(np.arange(1, n + 1), np.diff(ptr))
    val = val * (1.0 + 0.2 * np.sign(rows - node)) * np.where(rows == node, 1.05, 1.0)
    A = sg.csr_matrix(n, n, ptr, node, val)
    b = np.sin(0.01 * np.arange(1, n + 1)) + 0.5
    for jac in (False, True):
        pc = None
        if jac:
            pc = sg.jacobi(); pc.setup(A)
        s = sg.bicgstab(1e-9); s.setup(A)
        u = np.full(n, 0.25)
        s.solve(A, u, b, pc)
        Au = np.zeros(n); A.matvec(u, Au)
        print(f"{nx}x{ny} jacobi={int(jac)} iterations {s.iterations} sqrt(res2) {np.sqrt(s.res2):.3e} true residual {np.abs(Au - b).max():.3e}", flush=True)
