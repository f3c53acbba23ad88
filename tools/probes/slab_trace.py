#!/usr/bin/env python3
"""Chain progress over time inside k_trsv_slab (clock samples every 16 steps): python tools/probes/slab_trace.py nx ny nz [groups...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
nx, ny, nz = (int(v) for v in sys.argv[1:4])
n = nx * ny * nz
ptr, node, val = P.laplace3d_csr(nx, ny, nz)
A = sg.csr_matrix(n, n, ptr, node, val); pc = sg.ldu(); pc.setup(A)
r = torch.ones(n, dtype=torch.float64, device='cuda'); z = torch.zeros_like(r)
for _ in range(3): pc.solve(A, z, r)
torch.cuda.synchronize()
st = pc.get("slabs", np.int32); NI, NB, HB, S = [int(v) for v in st[:4]]
ck = pc.get("slab_clocks", np.int64)
se = ck[:2 * NB * NI].reshape(-1, 2); tr = ck[2 * NB * NI:NB * NI * (2 + S // 16)].reshape(NB * NI, S // 16)
ht = ck[NB * NI * (2 + S // 16):].reshape(NB, 2, 128, 2)
t0 = se[:, 0].min()
np.set_printoptions(linewidth=250, precision=0, suppress=True)
print("slabs", st.tolist())
groups = [int(v) for v in sys.argv[4:]] or [0, 1, 2, NB - 1]
for b in groups:
    for a in range(NI):
        print("group", b, "strip", a, "us at every 64th step:", ((tr[b * NI + a][::4] - t0) / 100.0).round(0), "end", round((se[b * NI + a][1] - t0) / 100.0))
for b in groups:
    for k, nm in ((0, "forwarder"), (1, "fetcher  ")):
        h = ht[b, k]; h = h[h[:, 0] > 0]
        print("group", b, nm, "first passes (us, steps):", [(round((c - t0) / 100.0, 1), int(v)) for c, v in h[:40]])
