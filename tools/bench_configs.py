#!/usr/bin/env python3
"""The single-GPU configs of BASELINE.json other than C2, one JSON line each -- the SAME legs bench.py puts into its
line (bench.c3_leg, bench.c4_leg and the C5-on-one-GPU product + CG), callable on their own for profiling runs
(`rocprofv3 ... -- python3 tools/bench_configs.py --configs c4`):
  C3  1-D advection-diffusion CSR n=1e7: SpMV, BiCGStab and GMRES(30), fixed 300 iterations
  C4  ELLPACK random digraph, degree 32, n=5e6: SpMV (generated on the device)
  C5  3-D 7-point Laplacian 464^3 (n=99,897,344): SpMV + CG 200 iterations on ONE GPU
  asm device-side assembly of the C2 matrix from its edge list
Every fraction is named after the bytes it is made of: `frac_moved` (what the kernel moves by construction, never above
1) and `frac_survey_bytes` (SURVEY 8d's reference-layout bytes; null where the kernel reads a compressed layout) --
bench.spmv_fracs."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import sigma_amd as sg  # noqa: E402
from sigma_amd import problems as P  # noqa: E402

# kept under their old names for the probes that import them from here
timed = lambda fn, reps: bench.timed_launches(torch, fn, reps)  # noqa: E731


def stencil_csr_torch(n, offs_masks_vals, dev):
    return P.stencil_csr_torch(torch.arange(n, device=dev, dtype=torch.int64), offs_masks_vals, dev)


def laplace3d_torch(nx, ny, nz, dev):
    return P.laplace3d_rows_torch(nx, ny, nz, dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c3,c4,c5")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the problems (testing)")
    ap.add_argument("--gmres-mgs", action="store_true", help="C3: GMRES(30) with modified Gram-Schmidt (solver option gmres_cgs2 = 0)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sg.init(0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    sg.use_torch_stream()
    sg.set_async(True)
    todo = args.configs.split(",")

    if "asm" in todo:
        # device-side assembly from the edge list (reference: 12.7 s on one core at n ~ 1e7)
        nx = int(3162 * args.scale ** 0.5)
        n = nx * nx
        ptr, node, val = P.poisson2d_csr(nx, nx)
        ei = np.repeat(np.arange(1, n + 1, dtype=np.int32), np.diff(ptr))
        d_ei, d_ej, d_ev = torch.from_numpy(ei).to(dev), torch.from_numpy(node).to(dev), torch.from_numpy(val).to(dev)
        torch.cuda.synchronize()
        sg.set_async(False)
        for rep in range(2):
            t0 = time.perf_counter()
            A = sg.csr_matrix.from_edges(n, n, d_ei, d_ej, d_ev)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        ok = bool(np.array_equal(A.get("node", np.int32), node) and np.array_equal(A.get("ptr", np.int32), ptr))
        sg.set_async(True)
        print(json.dumps({"config": "assembly: edge list -> CSR on the device (incl. the offset-dictionary and sliced-code build, also on the device)",
                          "n": n, "edges": int(len(ei)), "seconds": dt, "arrays_equal_generator": ok}), flush=True)
        del A, d_ei, d_ej, d_ev

    if "c3" in todo:
        print(json.dumps(bench.c3_leg(sg, P, torch, dev, n=int(1e7 * args.scale), gmres_orth=0 if args.gmres_mgs else 1)), flush=True)

    if "c4" in todo:
        print(json.dumps(bench.c4_leg(sg, P, torch, dev, n=int(5e6 * args.scale))), flush=True)

    if "c5" in todo:
        m = max(8, int(round(464 * args.scale ** (1 / 3))))
        n = m ** 3
        ptr, node, val = P.laplace3d_rows_torch(m, m, m, dev)
        nnz = int(val.numel())
        torch.cuda.synchronize()
        A = sg.csr_matrix(n, n, ptr, node, val)
        del ptr, node, val
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        t = bench.timed_launches(torch, lambda: A.matvec(x, y), 20)
        _, moved = A.footprint()
        out = {"workload": f"C5 3-D 7-point Laplacian {m}^3 on ONE GPU, n={n}, nnz={nnz}", "kernel": A.kernel}
        out.update(bench.spmv_fracs(t, moved, bench.spmv_bytes(n, n, nnz)))
        b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
        its, dt, res2 = bench.fixed_iterations(sg, torch, lambda: sg.cg(1e-300), A, n, b, 200)
        per_it = moved + 64 * n
        out["cg"] = {"iterations": its, "iters_per_s": its / dt, "ms_per_iter": 1e3 * dt / its, "final_res2": res2,
                     "moved_bytes_per_iter": per_it, "frac_moved": per_it * its / dt / 1e9 / bench.HBM_PEAK_GBS}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
