#!/usr/bin/env python3
"""Where does the column-blocked two-phase form start to pay for a CSR matrix with scattered columns?  n from 2.5e5 to 4e6 (x of
2 ... 32 MB), 16 and 32 entries per row, uniformly random columns: the default layout against option ell_colblock = 2 (forced).
    python tools/probes/colblock_threshold.py"""
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
rs = np.random.RandomState(3)
for d in (16, 32):
    for n in (250_000, 500_000, 1_000_000, 1_500_000, 2_000_000, 4_000_000):
        if n * d > 70_000_000:
            continue
        rows = np.repeat(np.arange(n), d)
        cols = rs.randint(0, n, size=rows.size)
        ptr = (np.arange(n + 1) * d + 1).astype(np.int32)
        node, val = (cols + 1).astype(np.int32), rs.standard_normal(rows.size)
        out = {"n": n, "per_row": d, "x_MB": n * 8 / 2 ** 20}
        for forced in (0, 1):
            sg.set_option("ell_colblock", 2 if forced else 1)
            H = sg.csr_matrix(n, n, ptr, node, val)
            sg.set_option("ell_colblock", 1)
            x = torch.randn(n, dtype=torch.float64, device=dev)
            y = torch.zeros(n, dtype=torch.float64, device=dev)
            for _ in range(3):
                H.matvec(x, y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                H.matvec(x, y)
            e1.record(st)
            torch.cuda.synchronize()
            out["forced" if forced else "default"] = {"kernel": H.kernel, "us": 1e3 * e0.elapsed_time(e1) / 10}
            H.destroy()
        out["default_over_forced"] = out["default"]["us"] / out["forced"]["us"]
        print(json.dumps(out), flush=True)
