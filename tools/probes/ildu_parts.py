#!/usr/bin/env python3
"""Colour-ordered ILDU(0)-PCG on a row partition: the 2-D Poisson nx^2 matrix as P in-process parts,
sg.ldu(reorder="colour") -- every part orders and factors its own diagonal block -- against the same solve on one part.
  python tools/probes/ildu_parts.py 3162 8 [iterations]
Fixed iteration count (device vectors: no host staging in the timed region); microseconds per iteration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sigma_amd as sg  # noqa: E402
from sigma_amd import problems as P  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 3162
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
only = sys.argv[4] if len(sys.argv) > 4 else ""           # "one" / "parts": one matrix form; + ":none" / ":ildu": one preconditioner
graph = not (len(sys.argv) > 5 and sys.argv[5] == "nograph")      # "nograph": the launch loop without the replayed groups
n = nx * nx
dev = torch.device("cuda", 0)
sg.init(0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
sg.use_torch_stream()
ptr, node, val = P.poisson2d_csr(nx, nx)
b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
for label, np_ in (("one part", 1), (f"{parts} in-process parts", parts)):
    if only and only.split(":")[0] != ("one" if np_ == 1 else "parts"):
        continue
    if np_ == 1:
        A = sg.csr_matrix(n, n, ptr, node, val)
    else:
        starts = (np.arange(np_ + 1) * (nx // np_) * nx).astype(np.int64)       # whole grid lines per part
        starts[-1] = n
        starts = starts // 2 * 2
        starts[-1] = n
        A = sg.partitioned_csr_matrix(n, n, ptr, node, val, starts)
    for pcname, mk in (("none", None), ("ildu0 colour", lambda: sg.ldu(reorder="colour"))):
        if ":" in only and only.split(":")[1] != ("none" if mk is None else "ildu"):
            continue
        pc = mk() if mk else None
        t0 = time.perf_counter()
        if pc:
            pc.setup(A)
        sg.synchronize()
        tset = time.perf_counter() - t0
        s = sg.cg(1e-300)
        s.set_max_iter(iters)
        if not graph:
            s.set_option("krylov_graph", 0)
        s.setup(A)
        u = torch.zeros(n, dtype=torch.float64, device=dev)
        sg.set_async(True)
        s.solve(A, u, b, pc, check=False)
        u.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve(A, u, b, pc, check=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sg.set_async(False)
        out = {"grid": nx, "matrix": label, "pc": pcname, "setup_s": tset, "iterations": s.last_iterations,
               "us_per_iter": 1e6 * dt / max(1, s.last_iterations), "res2": s.res2, "krylov_graph": graph}
        if pc:
            out["pc_info_part0"] = pc.info(0)
        print(json.dumps(out), flush=True)
        s.destroy()
        if pc:
            pc.destroy()
    A.destroy()
