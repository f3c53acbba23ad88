#!/usr/bin/env python3
"""An unstructured-mesh-like matrix through the library's own re-ordering: the P1 graph of a triangulated
nx x ny grid (6 neighbours + diagonal), vertices renumbered at random (what a mesh generator's output looks like to
SpMV: no locality at all), then breadth_first_search + left/right permute on the device (permutations.f90:22-78,
cs_matrices.f90:471-490) and the product again.  Prints us per product and the fraction of 8 TB/s on CSR bytes; the
re-ordered product must equal the first one entry for entry (y2[p[i]] == y1[i]: rows keep their stored order).
  python tools/probes/fem_like.py [nx]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
_TOOLS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _TOOLS)
sys.path.insert(0, os.path.join(_TOOLS, "probes"))
import torch  # noqa: E402

import sigma_amd as sg  # noqa: E402
from bench_configs import timed  # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 3162
    ny = nx
    n = nx * ny
    dev = torch.device("cuda", 0)
    sg.init(0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    sg.use_torch_stream()
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    k = torch.arange(n, device=dev)
    i, j = k % nx, k // nx
    # triangulated grid: W, E, S, N and the two diagonal neighbours SW, NE; values like a stiffness matrix with jittered weights
    nbrs = ((-nx - 1, (i > 0) & (j > 0)), (-nx, j > 0), (-1, i > 0), (0, torch.ones_like(k, dtype=torch.bool)), (1, i < nx - 1),
            (nx, j < ny - 1), (nx + 1, (i < nx - 1) & (j < ny - 1)))
    perm = torch.randperm(n, device=dev, generator=g)              # new number of vertex v
    M = torch.stack([m for _o, m in nbrs], 1)
    Ccol = torch.stack([k + o for o, _m in nbrs], 1).clamp_(0, n - 1)
    V = torch.rand((n, 7), device=dev, dtype=torch.float64, generator=g) * -1.0
    V[:, 3] = 7.0
    deg = M.sum(1)
    # rows in the NEW numbering: row perm[v] holds v's entries with columns perm[neighbour]
    inv = torch.empty_like(perm)
    inv[perm] = k
    Mp, Cp, Vp = M[inv], perm[Ccol[inv]], V[inv]
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(Mp.sum(1), 0)
    node, val = (Cp[Mp] + 1).to(torch.int32), Vp[Mp].contiguous()
    nnz = int(val.numel())
    A = sg.csr_matrix(n, n, (ptr + 1).to(torch.int32), node, val)
    del M, Ccol, V, Mp, Cp, Vp, deg
    x = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    y1 = torch.zeros_like(x)
    moved = 12 * nnz + 4 * (n + 1) + 16 * n
    t1 = timed(lambda: A.matvec(x, y1), 20)
    print(json.dumps({"order": "random vertex numbers", "n": n, "nnz": nnz, "kernel": A.kernel, "us": round(t1 * 1e6, 1),
                      "frac_of_8TBs_csr_bytes": round(moved / t1 / 8e12, 3)}), flush=True)
    t0 = time.time()
    p = A.bfs_order()                                           # host int32, 1-based visiting numbers
    t_bfs = time.time() - t0
    assert p.min() >= 1, "graph not connected"
    pd = torch.from_numpy(p).to(dev)
    t0 = time.time()
    A.left_permute(pd)
    A.right_permute(pd)
    sg.synchronize()
    t_perm = time.time() - t0
    x2 = torch.empty_like(x)
    x2[(pd - 1).long()] = x
    y2 = torch.zeros_like(x)
    t2 = timed(lambda: A.matvec(x2, y2), 20)
    same = bool(torch.equal(y2[(pd - 1).long()], y1))
    print(json.dumps({"order": "after bfs_order + left/right permute", "kernel": A.kernel, "us": round(t2 * 1e6, 1),
                      "frac_of_8TBs_csr_bytes": round(moved / t2 / 8e12, 3), "bfs_s": round(t_bfs, 2), "permute_s": round(t_perm, 3),
                      "products_equal_entry_for_entry": same}), flush=True)


if __name__ == "__main__":
    main()
