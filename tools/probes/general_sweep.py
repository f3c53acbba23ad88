#!/usr/bin/env python3
"""k_csr_spmv (the any-row-length kernel) on the C2 matrix under one SGM_SPMV_CFG: us per product.
  SGM_SPMV_CFG="block,vpt,nt,maxgrid,remap" python tools/probes/general_sweep.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
nx = 3162; n = nx * nx
ptr, node, val = P.poisson2d_csr(nx, nx)
for o in ("csr_offset_dict", "csr_row_owner", "csr_row_lines", "csr_sliced"):
    sg.set_option(o, 0)
A = sg.csr_matrix(n, n, torch.from_numpy(ptr).cuda(), torch.from_numpy(node).cuda(), torch.from_numpy(val).cuda())
x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
sg.use_torch_stream()
for _ in range(5): A.matvec(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): A.matvec(x, y)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"cfg": os.environ.get("SGM_SPMV_CFG", "default"), "kernel": A.kernel, "us": e0.elapsed_time(e1) * 10.0}))
