#!/usr/bin/env python3
"""ILDU(0)-PCG vs Jacobi-PCG vs CG on a 2-D Poisson grid: iterations and time (one GPU).
  python tools/ildu_bench.py <nx> [cg,jacobi,ildu0,ildu0_reorder] [colour]
`colour`: the matrix is first re-ordered by the reference's own greedy_color_ordering
(permutations.f90) and permuted symmetrically -- ILDU(0) then has as many dependency levels as colours.
`ildu0_reorder`: sg.ldu(reorder="colour") on the matrix AS IT IS -- the colour ordering inside the preconditioner."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sigma_amd as sg
from sigma_amd import problems as P
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000       # negative: |nx|^3 7-point grid instead of nx^2 5-point
only = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cg", "jacobi", "ildu0"]
if nx < 0:
    m3 = -nx
    n = m3 ** 3
    ptr, node, val = P.laplace3d_csr(m3, m3, m3)
else:
    n = nx * nx
    ptr, node, val = P.poisson2d_csr(nx, nx)
sg.init(0)
A = sg.csr_matrix(n, n, ptr, node, val)
b = np.full(n, 1.0 / n)
colour = len(sys.argv) > 3 and sys.argv[3] == "colour"
if colour:
    t0 = time.time()
    p, ptrs, nc = A.greedy_color_ordering()
    t1 = time.time()
    A.left_permute(p)
    A.right_permute(p)
    sg.synchronize()
    t2 = time.time()
    bp = np.empty(n)
    bp[p - 1] = b
    b = bp
    print(json.dumps({"grid": nx, "reordering": "greedy_color_ordering", "colours": nc, "ordering_s": t1 - t0,
                      "permute_s": t2 - t1}), flush=True)
for name, mk in (("cg", None), ("jacobi", sg.jacobi), ("ildu0", sg.ldu), ("ildu0_reorder", lambda: sg.ldu(reorder="colour"))):
    if name not in only:
        continue
    pc = mk() if mk else None
    t0 = time.time()
    if pc:
        pc.setup(A)
    tset = time.time() - t0
    s = sg.cg(1e-8)
    s.setup(A)
    u = np.zeros(n)
    s.solve(A, u, b, pc)          # warm-up: code objects, function attributes, staging buffers
    it0 = s.iterations            # (the solver object counts across solves, as the reference's does)
    u = np.zeros(n)
    t0 = time.time()
    s.solve(A, u, b, pc)
    dt = time.time() - t0
    its = s.iterations - it0
    extra = {}
    if name == "ildu0_reorder":
        extra["reorder_ms"] = dict(zip(("ordering", "permuted_copy", "setup_on_copy", "colours"), pc.get("reorder_ms", np.float64).tolist()))
    if name.startswith("ildu0"):
        extra["levels"] = pc.get("levels", np.int32).tolist()
        extra["strips"] = pc.get("strips", np.int32).tolist()
        extra["slabs"] = pc.get("slabs", np.int32).tolist()
        extra["row_levels"] = pc.get("row_levels", np.int32).tolist()
    extra["spmv_kernel"] = A.kernel.split("<")[0]
    print(json.dumps({"grid": nx, "pc": name, "setup_s": tset, "iterations": its, "solve_s": dt,
                      "ms_per_iter_incl_host_staging": 1e3 * dt / max(its, 1), **extra}), flush=True)
