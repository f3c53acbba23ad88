#!/usr/bin/env python3
"""What does ONE long row cost?  n rows of 8 scattered entries + `count` rows of L entries each (in different row blocks):
product time against L and count.  The reference's row sum is a serial chain (cs_matrices.f90:611-620), 4.2 ns per dependent
fp64 add on this GPU: the floor for a row of L entries is L x 4.2 ns.    python tools/probes/long_row_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import sigma_amd as sg

sg.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
sg.use_torch_stream()
n = 250000
rs = np.random.RandomState(1)
for L, count in ((0, 0), (1000, 1), (5000, 1), (20000, 1), (5000, 16), (5000, 256), (100000, 1)):
    deg = np.full(n, 8)
    if count:
        deg[(np.arange(count) * (n // count) + 7) % n] = L
    rows = np.repeat(np.arange(n), deg)
    cols = rs.randint(0, n, size=rows.size)
    ptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    H = sg.csr_matrix(n, n, ptr, (cols + 1).astype(np.int32), rs.standard_normal(rows.size))
    x = torch.randn(n, dtype=torch.float64, device=dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        H.matvec(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        H.matvec(x, y)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"n": n, "long_rows": count, "L": L, "nnz": int(rows.size), "kernel": H.kernel, "us": 1e3 * ms,
                      "ns_per_entry_of_the_longest_row": (1e6 * ms / L) if L else None}), flush=True)
    H.destroy()
