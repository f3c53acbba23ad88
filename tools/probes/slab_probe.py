#!/usr/bin/env python3
"""Per-step cost and workgroup-to-workgroup lag of k_trsv_slab: ILDU(0) applies on nx x ny x nz 7-point grids.
  python tools/probes/slab_probe.py [nx,ny,nz ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(64, 8, 400), (64, 64, 100), (100, 100, 100), (128, 128, 128)]
for nx, ny, nz in shapes:
    n = nx * ny * nz
    ptr, node, val = P.laplace3d_csr(nx, ny, nz)
    A = sg.csr_matrix(n, n, ptr, node, val)
    pc = sg.ldu()
    pc.setup(A)
    r = torch.ones(n, dtype=torch.float64, device=dev)
    z = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        pc.solve(A, z, r)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        pc.solve(A, z, r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = pc.get("slabs", np.int32)
    out = {"grid": [nx, ny, nz], "slabs": st.tolist(), "apply_us": dt * 1e6}
    if st[0]:
        ck = pc.get("slab_clocks", np.int64)[:2 * int(st[0]) * int(st[1])].reshape(-1, 2)      # chain start / end per (group, strip)
        t0c = ck[:, 0].min()
        rows = [(round((a - t0c) / 100.0, 1), round((b - t0c) / 100.0, 1)) for a, b in ck]
        print("   chain start/end us per (group, strip):", rows[:8], "...", rows[-4:])
        out["chain_ns_per_step"] = float(np.median((ck[:, 1] - ck[:, 0]) * 10.0 / st[3]))
        out["sweep_us_by_clocks"] = float((ck[:, 1].max() - t0c) / 100.0)
    pc.set_option("ildu_strips", 0)
    for _ in range(2):
        pc.solve(A, z, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pc.solve(A, z, r)
    torch.cuda.synchronize()
    out["apply_us_level_walkers"] = (time.perf_counter() - t0) / 5 * 1e6
    pc.set_option("ildu_strips", 1)
    print(json.dumps(out), flush=True)
