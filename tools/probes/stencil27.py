#!/usr/bin/env python3
"""27-point stencil on an m^3 grid (27 entries per interior row): the default kernel and, with the structured-matrix
options off, the general ones.  us per product, fraction of 8 TB/s on reference-layout bytes (12 B/entry + 20 B/row)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sigma_amd as sg
sg.init(0); sg.use_torch_stream()
dev = torch.device("cuda", 0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 160
n = m ** 3
idx = torch.arange(n, device=dev)
i, j, k = idx % m, (idx // m) % m, idx // (m * m)
cols, vals, valid = [], [], []
for dk in (-1, 0, 1):
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            ok = (i + di >= 0) & (i + di < m) & (j + dj >= 0) & (j + dj < m) & (k + dk >= 0) & (k + dk < m)
            cols.append(idx + di + dj * m + dk * m * m); valid.append(ok)
            vals.append(torch.full((n,), 26.0 if (di, dj, dk) == (0, 0, 0) else -1.0 + 0.01 * (di + 3 * dj + 9 * dk), device=dev, dtype=torch.float64))
C, V, M = torch.stack(cols, 1), torch.stack(vals, 1), torch.stack(valid, 1)
deg = M.sum(1)
ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev); ptr[1:] = torch.cumsum(deg, 0)
node = (C[M] + 1).to(torch.int32); val = V[M].contiguous()
nnz = int(ptr[-1])
x = torch.rand(n, device=dev, dtype=torch.float64)
for opts in ({}, {"csr_offset_dict": 0, "csr_sliced": 0}, {"csr_offset_dict": 0, "csr_sliced": 0, "csr_row_owner": 0},
             {"csr_offset_dict": 0, "csr_sliced": 0, "csr_row_owner": 0, "csr_row_lines": 0}):
    for o in ("csr_offset_dict", "csr_sliced", "csr_row_owner", "csr_row_lines"):
        sg.set_option(o, opts.get(o, 1))
    A = sg.csr_matrix(n, n, (ptr + 1).to(torch.int32), node, val)
    y = torch.zeros_like(x)
    for _ in range(5): A.matvec(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): A.matvec(x, y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20.0
    ref = 12 * nnz + 20 * n
    print(json.dumps({"grid": m, "n": n, "nnz": nnz, "kernel": A.kernel, "us": us, "ref_layout_GB": ref / 1e9, "frac_of_8TBs_on_ref_bytes": ref / us / 8e6}), flush=True)
    A.destroy()
