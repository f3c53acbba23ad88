#!/usr/bin/env python3
"""Secondary configs of BASELINE.json on one MI355X (not the bench.py contract line):
  C3  1-D advection-diffusion CSR n=1e7: GMRES(30) and BiCGStab, fixed 300 iterations
  C4  ELLPACK random digraph, degree 32, n=5e6: SpMV
  C5  3-D 7-point Laplacian 464^3 (n=99,897,344): SpMV + CG 200 iterations on ONE GPU
Prints one JSON line per config.  Matrices are generated on the device with torch (plumbing)
in the same entry order as sigma_amd.problems (checked at small size by the tests)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sigma_amd as sg  # noqa: E402
from sigma_amd import problems as P  # noqa: E402

PEAK = 8000.0


def stencil_csr_torch(n, offs_masks_vals, dev):
    k = torch.arange(n, device=dev, dtype=torch.int64)
    cols = torch.stack([k + 1 + o for o, _, _ in offs_masks_vals], dim=1)
    mask = torch.stack([m for _, m, _ in offs_masks_vals], dim=1)
    vals = torch.tensor([v for _, _, v in offs_masks_vals], dtype=torch.float64, device=dev).expand(n, -1)
    ptr = torch.ones(n + 1, dtype=torch.int64, device=dev)
    ptr[1:] += torch.cumsum(mask.sum(dim=1), 0)
    return ptr.to(torch.int32), cols[mask].to(torch.int32), vals[mask].contiguous()


def laplace3d_torch(nx, ny, nz, dev):
    n = nx * ny * nz
    k = torch.arange(n, device=dev, dtype=torch.int64)
    i, j, l = k % nx, (k // nx) % ny, k // (nx * ny)
    one = torch.ones(n, dtype=torch.bool, device=dev)
    return stencil_csr_torch(n, [(-nx * ny, l > 0, -1.0), (-nx, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 6.0),
                                 (1, i < nx - 1, -1.0), (nx, j < ny - 1, -1.0), (nx * ny, l < nz - 1, -1.0)], dev)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def solver_run(mk, A, n, b, its):
    s = mk()
    s.set_max_iter(its)
    s.setup(A)
    u = torch.zeros(n, dtype=torch.float64, device=b.device)
    s.solve(A, u, b, check=False)
    u.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.solve(A, u, b, check=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return s.last_iterations, dt, s.res2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c3,c4,c5")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the problems (testing)")
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
        n = int(1e7 * args.scale)
        dx = 1.0 / (n + 1)
        c = 0.5
        ptr, node, val = P.tridiag_csr(n, 2.0, -1.0 + c * dx / 2, -1.0 - c * dx / 2)
        A = sg.csr_matrix(n, n, torch.from_numpy(ptr).to(dev), torch.from_numpy(node).to(dev),
                          torch.from_numpy(val).to(dev))
        nnz = len(val)
        bcsr = 12 * nnz + 4 * (n + 1) + 16 * n
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        t = timed(lambda: A.matvec(x, y), 50)
        b = torch.full((n,), 2.0 * dx * dx, dtype=torch.float64, device=dev)
        out = {"config": "C3 1-D advection-diffusion CSR", "n": n, "nnz": nnz,
               "spmv_us": t * 1e6, "spmv_GBs": bcsr / t / 1e9, "spmv_frac": bcsr / t / 1e9 / PEAK}
        for name, mk, bytes_it in (("bicgstab", lambda: sg.bicgstab(1e-300), 2 * bcsr + 112 * n),
                                   ("gmres30", lambda: sg.gmres(1e-300, 30), None)):
            its, dt, res2 = solver_run(mk, A, n, b, 300)
            out[name] = {"iterations": its, "iters_per_s": its / dt, "ms_per_iter": 1e3 * dt / its, "res2": res2}
            if bytes_it:
                out[name]["GBs_on_fused_floor"] = bytes_it * its / dt / 1e9
            else:   # MGS: B_csr + 8n(5j+3) at inner step j=1..30
                cyc = sum(bcsr + 8 * n * (5 * j + 3) for j in range(1, 31)) + 8 * n * 32 + bcsr
                out[name]["GBs_on_survey_formula"] = cyc * (its / 30.0) / dt / 1e9
        print(json.dumps(out), flush=True)
        del A

    if "c4" in todo:
        n = int(5e6 * args.scale)
        t0 = time.time()
        ei, ej, ev = P.random_regular_ell(n, 32, 12345)
        node = ej.reshape(n, 32)        # no padding: every row has exactly 32 slots, insertion order
        val = ev.reshape(n, 32)
        gen_s = time.time() - t0
        torch.cuda.synchronize()
        t0 = time.time()
        A = sg.ellpack_matrix(n, n, node, val)
        torch.cuda.synchronize()
        create_s = time.time() - t0
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        bell = 12 * n * 32 + 16 * n
        out = {"config": "C4 ELLPACK random digraph degree 32", "n": n, "max_d": 32, "gen_s": gen_s, "create_s": create_s,
               "algorithmic_bytes": bell, "resident_bytes": A.footprint()[0]}
        for opt, label in ((1, "default"), (0, "ell_colblock=0")):
            sg.set_option("ell_colblock", opt)
            t = timed(lambda: A.matvec(x, y), 30)
            mv = A.footprint()[1]
            out[label] = {"kernel": A.kernel, "spmv_us": t * 1e6, "moved_bytes": mv, "GBs_moved": mv / t / 1e9,
                          "frac_of_hbm_peak_moved": mv / t / 1e9 / PEAK, "effective_GBs_on_reference_bytes": bell / t / 1e9,
                          "effective_frac": bell / t / 1e9 / PEAK}
        sg.set_option("ell_colblock", 1)
        print(json.dumps(out), flush=True)
        del A

    if "c5" in todo:
        m = max(8, int(round(464 * args.scale ** (1 / 3))))
        n = m ** 3
        ptr, node, val = laplace3d_torch(m, m, m, dev)
        nnz = int(val.numel())
        torch.cuda.synchronize()
        A = sg.csr_matrix(n, n, ptr, node, val)
        del ptr, node, val
        bcsr = 12 * nnz + 4 * (n + 1) + 16 * n
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        t = timed(lambda: A.matvec(x, y), 20)
        b = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
        its, dt, res2 = solver_run(lambda: sg.cg(1e-300), A, n, b, 200)
        print(json.dumps({"config": f"C5 3-D 7-point Laplacian {m}^3 on ONE GPU", "n": n, "nnz": nnz,
                          "spmv_us": t * 1e6, "spmv_GBs": bcsr / t / 1e9, "spmv_frac": bcsr / t / 1e9 / PEAK,
                          "cg": {"iterations": its, "iters_per_s": its / dt, "ms_per_iter": 1e3 * dt / its,
                                 "GBs_on_floor": (bcsr + 72 * n) * its / dt / 1e9,
                                 "frac": (bcsr + 72 * n) * its / dt / 1e9 / PEAK, "res2": res2}}), flush=True)


if __name__ == "__main__":
    main()
