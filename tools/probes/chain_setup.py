#!/usr/bin/env python3
"""ILDU(0) setup + apply on a factor that is one chain (tridiagonal-like band), with SGM_PC_TIMING=1: where the setup goes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import scipy.sparse as sp
import sigma_amd as sg
sg.init(0)
for n in (100000, 400000):
    S = sp.diags([-np.ones(n - 7), -np.ones(n - 1), 4.0 * np.ones(n), -np.ones(n - 1), -np.ones(n - 7)], [-7, -1, 0, 1, 7]).tocsr()
    ptr, node, val = (S.indptr + 1).astype(np.int32), (S.indices + 1).astype(np.int32), S.data.copy()
    H = sg.csr_matrix(n, n, ptr, node, val)
    pc = sg.ldu()
    t0 = time.perf_counter()
    pc.setup(H)
    sg.synchronize()
    print(f"n={n}: setup {time.perf_counter() - t0:.3f} s", pc.info(0), flush=True)
    r = np.ones(n); z = np.zeros(n)
    t0 = time.perf_counter(); pc.solve(H, z, r); print(f"   first apply {time.perf_counter() - t0:.3f} s", flush=True)
    t0 = time.perf_counter(); pc.solve(H, z, r); print(f"   second apply {time.perf_counter() - t0:.3f} s", flush=True)
    pc.destroy(); H.destroy()
