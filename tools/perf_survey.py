#!/usr/bin/env python3
"""Which kernel does a matrix get, and how far from the roofline does it run?  The seeded generator of tests/fuzz_formats.py
(bands, bands with outliers, scattered, power-law rows, perturbed stencils, rectangular, blocks) at sizes where a product is
HBM-bound (>= 2e6 stored entries), default options: kernel name and the fraction of 8 TB/s on the reference layout's bytes
(12 nnz + 4 (n + 1) + 8 m + 8 n).  One JSON line per matrix, then the fractions by kind and kernel.

    python tools/perf_survey.py [seconds] [first_seed] [ell]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import sigma_amd as sg

# the generator only (the oracle is not used here: nothing is checked, only timed)
_src = open(os.path.join(ROOT, "tests", "fuzz_formats.py")).read()
_ns = {"__name__": "gen", "__file__": os.path.join(ROOT, "tests", "fuzz_formats.py")}
exec(compile(_src.split("OPTIONS = (")[0].replace("import oracle as orc\n", ""), "gen", "exec"), _ns)
make = _ns["make"]


def make_ell(rs):
    """ELLPACK matrices with full rows (no padding): stencil-like, banded, scattered columns."""
    kind = ["ell_stencil", "ell_band", "ell_scattered"][int(rs.randint(0, 3))]
    md = int(rs.choice([3, 5, 7, 8, 9, 12, 16, 20, 27, 32, 40]))
    n = int(10 ** rs.uniform(5.0, 6.6))
    n = max(1000, min(n, 12_000_000 // md))
    rows = np.arange(n)[:, None]
    if kind == "ell_stencil":
        offs = np.sort(rs.choice(np.arange(-3 * md, 3 * md + 1), size=md, replace=False)) * int(rs.choice([1, 1, 37, 1000]))
        node = np.clip(rows + offs[None, :], 0, n - 1)
    elif kind == "ell_band":
        bw = int(10 ** rs.uniform(0.5, 4.0))
        node = np.clip(rows + rs.randint(-bw, bw + 1, size=(n, md)), 0, n - 1)
    else:
        node = rs.randint(0, n, size=(n, md))
    return kind, n, md, (node + 1).astype(np.int32), rs.standard_normal((n, md))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 900000
    ell = len(sys.argv) > 3 and sys.argv[3] == "ell"
    sg.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    sg.use_torch_stream()
    t0 = time.time()
    rows = []
    while time.time() - t0 < seconds:
        rs = np.random.RandomState(seed)
        seed += 1
        if ell:
            kind, n, md, node, val = make_ell(rs)
            m, ptr = n, np.arange(0, n * md + 1, md) + 1
        else:
            kind, n, m, ptr, node, val = make(rs)
        if val.size < 2_000_000:
            continue
        tc = time.perf_counter()
        H = sg.ellpack_matrix(n, m, node, val) if ell else sg.csr_matrix(n, m, ptr, node, val)
        sg.synchronize()
        create_s = time.perf_counter() - tc
        x = torch.randn(m, dtype=torch.float64, device=dev)
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        for _ in range(3):
            H.matvec(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(st)
        for _ in range(reps):
            H.matvec(x, y)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        bytes_ref = 12 * val.size + (0 if ell else 4 * (n + 1)) + 8 * m + 8 * n          # (B_ell = 12 n max_d + 8 m + 8 n)
        deg = np.diff(ptr)
        row = {"seed": seed - 1, "kind": kind, "n": n, "m": m, "nnz": int(val.size), "mean_row": float(deg.mean()), "max_row": int(deg.max()),
               "kernel": H.kernel, "ms": ms, "frac_reference_bytes": bytes_ref / (ms * 1e-3) / 8e12, "create_s": create_s}
        rows.append(row)
        print(json.dumps(row), flush=True)
        H.destroy()
    by = {}
    for r in rows:
        by.setdefault((r["kind"], r["kernel"].split("<")[0]), []).append(r["frac_reference_bytes"])
    for k in sorted(by):
        v = np.array(by[k])
        print(f"# {k[0]:18s} {k[1]:14s} {len(v):4d} matrices  frac of 8 TB/s on reference bytes: min {v.min():.2f}  median {np.median(v):.2f}  max {v.max():.2f}")


if __name__ == "__main__":
    main()
