#!/usr/bin/env python3
"""C4 (ELLPACK random digraph, degree 32, n = 5e6): the column-blocked two-phase product with and without row bands
(options ell_colblock_band / _pieces / _nt).  Run on the GPU box; prints one JSON line per setting."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sigma_amd as sg
from sigma_amd import problems as P

sg.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); sg.use_torch_stream(); sg.set_async(True)
n = int(os.environ.get("C4_N", 5_000_000))
ei, ej, ev = P.random_regular_ell(n, 32, 12345)
node, val = ej.reshape(n, 32), ev.reshape(n, 32)
x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
y = torch.zeros(n, dtype=torch.float64, device=dev)
ref = None


def timed(A, reps=20):
    for _ in range(3):
        A.matvec(x, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        A.matvec(x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


settings = [(-1, 512, 0), (0, 512, 0), (0, 512, 1), (0, 256, 0), (0, 1024, 0), (131072, 512, 0), (524288, 512, 0), (1048576, 512, 0)]
if len(sys.argv) > 1:
    settings = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for band, pieces, nt in settings:
    sg.set_option("ell_colblock_band", band)
    sg.set_option("ell_colblock_pieces", pieces)
    sg.set_option("ell_colblock_nt", nt)
    A = sg.ellpack_matrix(n, n, node, val)
    us = timed(A)
    A.matvec(x, y); torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    same = bool(torch.equal(y, ref))
    res, mv = A.footprint()
    print(json.dumps({"band": band, "pieces": pieces, "nt": nt, "us": round(us, 1), "frac_of_8TBs_on_2.0GB": round(2.0e9 / (us * 1e-6) / 8e12, 3),
                      "moved_GB": round(mv / 1e9, 2), "resident_GB": round(res / 1e9, 2), "bit_identical_to_first": same, "kernel": A.kernel}), flush=True)
    A.destroy()
