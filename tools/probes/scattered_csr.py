#!/usr/bin/env python3
"""A CSR matrix with scattered columns: n = 5e6, 8..32 entries per row at uniformly random columns,
seeded.  Time per product of the column-blocked two-phase form it now gets at creation (k_ellcb<..,csr>) against the row
kernels the same handle runs with option ell_colblock = 0; the two results bit for bit; a sample of rows against their
sums evaluated with torch in stored order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import sigma_amd as sg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
dev = torch.device("cuda", 0)
sg.init(0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
sg.use_torch_stream()
g = torch.Generator(device=dev)
g.manual_seed(12345)
deg = torch.randint(8, 33, (n,), device=dev, generator=g, dtype=torch.int64)
ptr = torch.ones(n + 1, dtype=torch.int64, device=dev)
ptr[1:] += torch.cumsum(deg, 0)
nnz = int(ptr[-1].item()) - 1
node = torch.randint(1, n + 1, (nnz,), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
val = torch.rand(nnz, device=dev, generator=g, dtype=torch.float64) - 0.5
x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
torch.cuda.synchronize()
A = sg.csr_matrix(n, n, ptr.to(torch.int32), node, val)
out = {"n": n, "nnz": nnz, "kernel_default": A.kernel}
y1 = torch.zeros(n, dtype=torch.float64, device=dev)
y0 = torch.zeros(n, dtype=torch.float64, device=dev)
sg.set_async(True)
t1 = bench.timed_launches(torch, lambda: A.matvec(x, y1), 20)
A.set_option("ell_colblock", 0)
out["kernel_rows"] = A.kernel
t0 = bench.timed_launches(torch, lambda: A.matvec(x, y0), 10)
A.set_option("ell_colblock", 1)
torch.cuda.synchronize()
_, moved = A.footprint()
# a sample of rows in stored order: products rounded one by one, added left to right
rows = torch.arange(0, n, max(1, n // 200000), device=dev)
ok = True
z = torch.zeros(len(rows), dtype=torch.float64, device=dev)
p0 = ptr[rows] - 1
for k in range(32):
    m = deg[rows] > k
    idx = torch.where(m, p0 + k, torch.zeros_like(p0))
    term = val[idx] * x[(node[idx] - 1).to(torch.int64)]
    z = torch.where(m, z + term, z)
out.update({"ms_column_blocked": 1e3 * t1, "ms_row_kernels": 1e3 * t0, "speedup": t0 / t1,
            "bit_identical_both_forms": bool(torch.equal(y0, y1)), "sample_rows_equal_stored_order_sums": bool(torch.equal(0.0 + z, y1[rows])),
            "moved_bytes": moved, "frac_moved": moved / t1 / 1e9 / 8000.0,
            "frac_reference_bytes": (12 * nnz + 4 * (n + 1) + 16 * n) / t1 / 1e9 / 8000.0})
print(json.dumps(out))
