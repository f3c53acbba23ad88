#!/usr/bin/env python3
"""The C5 product alone (7-point 464^3 on one GPU): a few launches of k_csr_sl<7>, for counter passes
(`rocprofv3 --pmc ... -- python3 tools/probes/c5_product.py`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sigma_amd as sg  # noqa: E402
from sigma_amd import problems as P  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 464
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
sg.init(0)
n = m ** 3
ptr, node, val = P.laplace3d_rows_torch(m, m, m, dev)
A = sg.csr_matrix(n, n, ptr, node, val)
del ptr, node, val
x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
y = torch.zeros(n, dtype=torch.float64, device=dev)
for _ in range(reps):
    A.matvec(x, y)
torch.cuda.synchronize()
print("kernel", A.kernel, "n", n, flush=True)
