#!/usr/bin/env python3
"""The any-row-length CSR kernel (k_csr_spmv) where it belongs: banded matrices with long rows of varying length.
Prints us per product and the fraction of 8 TB/s on the bytes it moves (12 B per entry + 4 B per row + x + y)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sigma_amd as sg
sg.init(0)
sg.use_torch_stream()
dev = torch.device("cuda", 0)
CASES = ((4_000_000, 33, 64, 4096), (2_000_000, 64, 128, 8192), (8_000_000, 20, 40, 2048), (6_000_000, 8, 60, 3000), (1_000_000, 150, 300, 20000))
if os.environ.get("GR_CASES"):          # e.g. GR_CASES=0,2 GR_KERNELS=1 GR_REPS=3 under a profiler
    CASES = tuple(CASES[int(t)] for t in os.environ["GR_CASES"].split(","))
if os.environ.get("GR_OPTS") == "row_owner0": sg.set_option("csr_row_owner", 0)
# kernels to time on every matrix: (csr_sell, csr_row_lines) -- SELL-128-512 whenever its padding allows, then the CSR kernels
# (row owner up to 64 entries per row / line-staged beyond; streaming)
# (a third field: csr_xwindow, the SELL kernel's LDS-staged window of x -- "2:1:0" = SELL without it)
KERNELS = tuple(tuple(int(v) for v in t.split(":")) for t in os.environ.get("GR_KERNELS", "2:1:1,2:1:0,0:1:1,0:0:1").split(","))
REPS = int(os.environ.get("GR_REPS", "50"))
for n, lo, hi, band in CASES:
    g = torch.Generator(device=dev); g.manual_seed(1)
    deg = torch.randint(lo, hi + 1, (n,), device=dev, generator=g, dtype=torch.int64)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev); ptr[1:] = torch.cumsum(deg, 0)
    nnz = int(ptr[-1])
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    slot = torch.arange(nnz, device=dev) - ptr[rows]
    # ascending columns inside a band around the diagonal: start + slot * stride (+ jitter below the stride)
    stride = max(1, (2 * band) // hi)
    col = rows - band + slot * stride + torch.randint(0, stride, (nnz,), device=dev, generator=g)
    col = col.clamp_(0, n - 1)
    val = torch.rand(nnz, device=dev, generator=g, dtype=torch.float64)
    sg.set_option("csr_sell", 2)
    A = sg.csr_matrix(n, n, (ptr + 1).to(torch.int32), (col + 1).to(torch.int32), val)
    sg.set_option("csr_sell", 1)
    x = torch.rand(n, device=dev, dtype=torch.float64)
    moved = 12 * nnz + 4 * n + 16 * n
    ys = []
    for sell, rowline, xw in KERNELS:
        A.set_option("csr_sell", sell)
        A.set_option("csr_row_lines", rowline)
        A.set_option("csr_xwindow", xw)
        y = torch.zeros_like(x)
        for _ in range(min(5, REPS)): A.matvec(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS): A.matvec(x, y)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / REPS
        ys.append(y)
        # check a few rows against a host loop in stored order
        idx = torch.randint(0, n, (200,), device=dev, generator=g)
        ok = True
        for i in idx.tolist():
            k0, k1 = int(ptr[i]), int(ptr[i + 1])
            z = 0.0
            vv = val[k0:k1].cpu().numpy(); xx = x[col[k0:k1]].cpu().numpy()
            for a, b in zip(vv, xx): z = z + a * b
            ok = ok and (0.0 + z == float(y[i]))
        print(json.dumps({"n": n, "nnz_per_row": [lo, hi], "kernel": A.kernel, "us": round(us, 1), "moved_GB": round(moved / 1e9, 3),
                          "TBs": round(moved / us / 1e6, 3), "frac_of_8TBs": round(moved / us / 8e6, 3), "rows_bit_exact": bool(ok)}), flush=True)
    if len(ys) > 1: print(json.dumps({"kernels_bit_identical": bool(all(torch.equal(ys[0], yy) for yy in ys[1:]))}), flush=True)
    A.destroy()
