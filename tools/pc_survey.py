#!/usr/bin/env python3
"""Which sweep does an ILDU(0) apply get, and what does it cost?  SPD systems of tests/fuzz_solvers.py's generator (2-D / 3-D
grids with random coefficients, bands, random symmetric graphs) at n >= 5e4, natural order and reorder="colour": setup time,
the path sgm_pc_info names (levels, estimate), the measured apply, nanoseconds per row.  One JSON line per system.

    python tools/pc_survey.py [seconds] [first_seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp   # noqa: F401  (the generator uses it)
import torch

import sigma_amd as sg

_src = open(os.path.join(ROOT, "tests", "fuzz_solvers.py")).read()
_ns = {"__name__": "gen", "__file__": os.path.join(ROOT, "tests", "fuzz_solvers.py")}
exec(compile(_src.split("def block_diagonal")[0].replace("import oracle as orc\n", ""), "gen", "exec"), _ns)
make = _ns["make"]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 980000
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
        kind, n, (ptr, node, val) = make(rs)
        if n < 50000:
            continue
        H = sg.csr_matrix(n, n, ptr, node, val)
        r = torch.randn(n, dtype=torch.float64, device=dev)
        z = torch.zeros(n, dtype=torch.float64, device=dev)
        for order in ("natural", "colour"):
            pc = sg.ldu(reorder="colour") if order == "colour" else sg.ldu()
            tc = time.perf_counter()
            try:
                pc.setup(H)
            except sg.SigmaError as e:
                print(json.dumps({"seed": seed - 1, "kind": kind, "n": n, "order": order, "refused": str(e)[:80]}), flush=True)
                pc.destroy()
                continue
            sg.synchronize()
            setup_s = time.perf_counter() - tc
            info = pc.info(0)
            for _ in range(2):
                pc.solve(H, z, r)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record(st)
            for _ in range(reps):
                pc.solve(H, z, r)
            e1.record(st)
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            row = {"seed": seed - 1, "kind": kind, "n": n, "nnz": int(val.size), "order": order, "path": info["name"], "levels": info["levels"],
                   "colours": info["colours"], "est_us": info["est_us"], "apply_us": us, "ns_per_row": 1e3 * us / n, "setup_s": setup_s}
            rows.append(row)
            print(json.dumps(row), flush=True)
            pc.destroy()
        H.destroy()
    by = {}
    for r in rows:
        by.setdefault((r["kind"], r["order"], r["path"].split(",")[0][:28]), []).append(r["ns_per_row"])
    for k in sorted(by):
        v = np.array(by[k])
        print(f"# {k[0]:11s} {k[1]:8s} {k[2]:28s} {len(v):4d} systems  apply ns per row: min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():8.2f}")


if __name__ == "__main__":
    main()
