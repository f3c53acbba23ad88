#!/usr/bin/env python3
"""Per-step cost and inter-strip lag of k_trsv_strip: ILDU(0) applies on nx x ny 5-point grids with 1, 2, 4 strips."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
dev = torch.device("cuda", 0)
CASES = ((64, 16000), (128, 16000), (256, 16000), (1024, 4000))
if len(sys.argv) > 1:
    CASES = CASES[:int(sys.argv[1])]
for nx, ny in CASES:
    n = nx * ny
    ptr, node, val = P.poisson2d_csr(nx, ny)
    A = sg.csr_matrix(n, n, ptr, node, val)
    pc = sg.ldu()
    pc.setup(A)
    r = torch.ones(n, dtype=torch.float64, device=dev)
    z = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        pc.solve(A, z, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        pc.solve(A, z, r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = pc.get("strips", np.int32)
    ck = pc.get("strip_clocks", np.int64).reshape(-1, 2)
    t0c = ck[:, 0].min()
    print("   strip start/end us:", [(round((a - t0c) / 100.0, 1), round((b - t0c) / 100.0, 1)) for a, b in ck][:16])
    print(json.dumps({"nx": nx, "ny": ny, "strips": int(st[0]), "steps": int(st[1]), "apply_us": dt * 1e6,
                      "ns_per_step_if_no_lag": dt * 1e9 / 2 / max(int(st[1]), 1)}), flush=True)
