#!/usr/bin/env python3
"""SpMV time vs problem size and stencil shape (device-generated matrices): separates the
size effect (Infinity Cache residency between launches) from the stencil's x reach.
  python tools/probes/size_sweep.py 2d:3162x3162 2d:10000x10000 3d:215 3d:464 1d:100000000"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_TOOLS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TOOLS)
sys.path.insert(0, os.path.join(_TOOLS, "probes"))
import torch  # noqa: E402

import sigma_amd as sg  # noqa: E402
from bench_configs import laplace3d_torch, stencil_csr_torch, timed  # noqa: E402


def build(spec, dev):
    kind, dims = spec.split(":")
    if kind == "1d":
        n = int(dims)
        k = torch.arange(n, device=dev)
        one = torch.ones(n, dtype=torch.bool, device=dev)
        return n, stencil_csr_torch(n, [(-1, k > 0, -1.0), (0, one, 2.0), (1, k < n - 1, -1.0)], dev)
    if kind == "1d7":       # 7 entries per row like the 3-D stencil, but every x entry next door: the footprint without the reach
        n = int(dims)
        k = torch.arange(n, device=dev)
        return n, stencil_csr_torch(n, [(d, (k + d >= 0) & (k + d < n), 6.0 if d == 0 else -1.0) for d in (-3, -2, -1, 0, 1, 2, 3)], dev)
    if kind == "2d":
        nx, ny = (int(t) for t in dims.split("x"))
        n = nx * ny
        k = torch.arange(n, device=dev)
        i, j = k % nx, k // nx
        one = torch.ones(n, dtype=torch.bool, device=dev)
        return n, stencil_csr_torch(n, [(-nx, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 4.0),
                                        (1, i < nx - 1, -1.0), (nx, j < ny - 1, -1.0)], dev)
    if "x" in dims:       # 3d:464x464x58 = one of eight ranks' z-slab of the 464^3 grid (interior planes only: no halo)
        nx, ny, nz = (int(t) for t in dims.split("x"))
        return nx * ny * nz, laplace3d_torch(nx, ny, nz, dev)
    m = int(dims)
    return m ** 3, laplace3d_torch(m, m, m, dev)


def main():
    dev = torch.device("cuda", 0)
    sg.init(0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    sg.use_torch_stream()
    sg.set_async(True)
    for spec in sys.argv[1:]:
        n, (ptr, node, val) = build(spec, dev)
        nnz = int(val.numel())
        A = sg.csr_matrix(n, n, ptr, node, val)
        del ptr, node, val
        x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        out = {"spec": spec, "n": n, "nnz": nnz}
        alg = 12 * nnz + 4 * (n + 1) + 16 * n
        w = next(k for k in (3, 5, 7, 8) if k >= (nnz + n - 1) // n)
        for name, opt, sl in (("sliced", 1, 1), ("dict", 1, 0), ("int32", 0, 0)):
            A.set_option("csr_offset_dict", opt)
            A.set_option("csr_sliced", sl)
            t = timed(lambda: A.matvec(x, y), 20)
            stored = ((8 * w + 4) * n + 16 * n) if sl else (9 if opt else 12) * nnz + 4 * (n + 1) + 16 * n
            out[name] = {"us": round(t * 1e6, 1), "frac_alg": round(alg / t / 8e12, 3),
                         "stored_TBs": round(stored / t / 1e12, 2)}
        A.set_option("csr_offset_dict", 1)
        A.set_option("csr_sliced", 1)
        print(json.dumps(out), flush=True)
        del A, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
