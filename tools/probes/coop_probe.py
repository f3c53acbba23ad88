#!/usr/bin/env python3
"""Where an iteration of the cooperative CG kernel spends its time (build tools/probes/libsigma_hip_probe.so first:
tools/probes/build_coop_probe.sh).  Thread 0 of workgroup 0 sums 10 ns ticks per phase over a 2000-iteration solve."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sigma_amd as sg
sg.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libsigma_hip_probe.so")
from sigma_amd import problems as P
sg.init(0); sg.use_torch_stream()
dev = torch.device("cuda", 0)
NAMES = ["row_sums", "dot1", "-", "update+dot2", "publish", "halo_load", "end_barrier", "-",
         "h:drain+barrier", "h:store+poll", "h:barrier", "h:block_sum"]
for nx in [int(v) for v in os.environ.get("NXS", "100,256,300,500,1000").split(",")]:
    n = nx * nx
    ptr, node, val = P.poisson2d_csr(nx, nx)
    A = sg.csr_matrix(n, n, torch.from_numpy(ptr).to(dev), torch.from_numpy(node).to(dev), torch.from_numpy(val).to(dev))
    b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
    s = sg.cg(1e-300); s.setup(A); s.set_max_iter(2000)
    u = torch.zeros(n, dtype=torch.float64, device=dev)
    s.solve(A, u, b, check=False); u.zero_(); s.solve(A, u, b, check=False); torch.cuda.synchronize()
    out = (C.c_longlong * 16)()
    assert sg.lib().sgm_debug_coop_probe(out) == 0
    its = max(1, out[15])
    print(json.dumps({"nx": nx, "n": n, "iterations_timed": its,
                      "ns_per_iteration": {NAMES[k]: round(out[k] * 10.0 / its, 1) for k in range(12) if NAMES[k] != "-"},
                      "sum_ns": round(sum(out[k] for k in (0, 1, 3, 4, 5, 6, 8, 9, 10, 11)) * 10.0 / its, 1)}), flush=True)
