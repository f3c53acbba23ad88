#!/usr/bin/env python3
"""Repeat ILDU(0) applies through the slab pipeline and compare every result with the level walkers' (bit for bit).
  python tools/probes/slab_stress.py w,h,nk[,reps] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
for arg in sys.argv[1:]:
    v = [int(x) for x in arg.split(",")]
    w, h, nk = v[:3]
    reps = v[3] if len(v) > 3 else 200
    n = w * h * nk
    ptr, node, val = P.laplace3d_csr(w, h, nk)
    rs = np.random.RandomState(1)
    val = val * (1.0 + 0.1 * rs.rand(len(val)))          # (ILDU of a nonsymmetric-valued matrix: fine for an apply test)
    A = sg.csr_matrix(n, n, ptr, node, val)
    pc = sg.ldu(); pc.setup(A)
    st = pc.get("slabs", np.int32).tolist()
    r = torch.from_numpy(rs.standard_normal(n)).cuda()
    z = torch.zeros_like(r); zref = torch.zeros_like(r)
    pc.set_option("ildu_strips", 0); pc.solve(A, zref, r); pc.set_option("ildu_strips", 1)
    bad = 0; first = None
    for k in range(reps):
        z.zero_()
        pc.solve(A, z, r)
        if not torch.equal(z, zref):
            bad += 1
            if first is None:
                d = (z != zref).nonzero().flatten().cpu().numpy()
                first = (k, len(d), d[:6].tolist(), [(int(x) % w, (int(x) // w) % h, int(x) // (w * h)) for x in d[:6]])
    print(arg, "slabs", st, "mismatching applies:", bad, "/", reps, "first:", first, flush=True)
# irregular patterns (the generic kernels: presence codes, per-row orders), generated as tests/test_gpu_parity.py does
if os.environ.get("SLAB_STRESS_MIXED"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    from test_gpu_parity import _grid3_like_matrix
    for w, h, nk, order, holes in ((200, 12, 9, "mixed", 0.0), (200, 12, 9, "asc", 0.2), (256, 20, 12, "mixed", 0.1), (130, 24, 10, "mixed", 0.0), (100, 20, 12, "mixed", 0.1)):
        n = w * h * nk
        ei, ej, ev = _grid3_like_matrix(n, w, h, order, seed=3, holes=holes)
        A = sg.csr_matrix.from_edges(n, n, ei, ej, ev)
        pc = sg.ldu(); pc.setup(A)
        st = pc.get("slabs", np.int32).tolist()
        rs = np.random.RandomState(2)
        r = torch.from_numpy(rs.standard_normal(n)).cuda()
        z = torch.zeros_like(r); zref = torch.zeros_like(r)
        pc.set_option("ildu_strips", 0); pc.solve(A, zref, r); pc.set_option("ildu_strips", 1)
        bad = 0; first = None
        for k in range(300):
            z.zero_()
            pc.solve(A, z, r)
            if not torch.equal(z, zref):
                bad += 1
                if first is None:
                    d = (z != zref).nonzero().flatten().cpu().numpy()
                    first = (k, len(d), [(int(x) % w, (int(x) // w) % h, int(x) // (w * h)) for x in d[:8]])
        print((w, h, nk, order, holes), "slabs", st, "mismatching applies:", bad, "/ 300 first:", first, flush=True)
