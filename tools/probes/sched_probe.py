#!/usr/bin/env python3
"""Slice schedule A/B (one process per configuration; SGM_SLICE_SCHED / SGM_SPMV_CFG are read at init):
SpMV time of the sliced kernel on 3-D grids, and a bit-identity check against the 1-byte-code kernel.
  SGM_SLICE_SCHED=1,64 python tools/probes/sched_probe.py 3d:464 3d:300"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_TOOLS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TOOLS)
sys.path.insert(0, os.path.join(_TOOLS, "probes"))
import torch  # noqa: E402

import sigma_amd as sg  # noqa: E402
from bench_configs import timed  # noqa: E402
from size_sweep import build  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    sg.init(0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    sg.use_torch_stream()
    sg.set_async(True)
    reps = int(os.environ.get("PROBE_REPS", "20"))
    for spec in sys.argv[1:]:
        n, (ptr, node, val) = build(spec, dev)
        nnz = int(val.numel())
        A = sg.csr_matrix(n, n, ptr, node, val)
        del ptr, node, val
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        t = timed(lambda: A.matvec(x, y), reps)
        _res, moved = A.footprint()
        out = {"spec": spec, "n": n, "nnz": nnz, "kernel": A.kernel,
               "sched": os.environ.get("SGM_SLICE_SCHED", "default"), "cfg": os.environ.get("SGM_SPMV_CFG", "default"),
               "us": round(t * 1e6, 1), "moved_TBs": round(moved / t / 1e12, 3), "frac": round(moved / t / 8e12, 3)}
        if os.environ.get("PROBE_CHECK", "1") == "1":
            A.set_option("csr_sliced", 0)
            y2 = torch.zeros(n, dtype=torch.float64, device=dev)
            A.matvec(x, y2)
            sg.synchronize()
            out["bit_identical_to_dict_kernel"] = bool(torch.equal(y, y2))
            A.set_option("csr_sliced", 1)
            del y2
        print(json.dumps(out), flush=True)
        del A, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
