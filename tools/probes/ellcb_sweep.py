#!/usr/bin/env python3
"""C4 sweep of the column-blocked ELLPACK kernel's launch parameters (run on the GPU box)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, json, time, numpy as np, torch
sys.path.insert(0, %r)
import sigma_amd as sg
from sigma_amd import problems as P
sg.init(0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); sg.use_torch_stream(); sg.set_async(True)
n = 5_000_000
node, val = P.random_regular_ell_torch(n, 32, 12345, dev)
x = torch.sin(0.001 * torch.arange(1, n + 1, dtype=torch.float64, device=dev))
y = torch.zeros(n, dtype=torch.float64, device=dev)
def timed(A, reps=20):
    for _ in range(3): A.matvec(x, y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): A.matvec(x, y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for cols in (20480, 16384, 8192):
    sg.set_option("ell_colblock_cols", cols)
    A = sg.ellpack_matrix(n, n, node, val)
    for chunks in (4, 8, 16):
        __import__("os").environ["SGM_ELLCB_CHUNKS"] = str(chunks); A.set_option("ell_colblock", 0); A.set_option("ell_colblock", 1)
        print(json.dumps({"cols": cols, "chunks": chunks, "grid2": int(__import__("os").environ.get("SGM_ELLCB_GRID", 2048)), "us": timed(A), "kernel": A.kernel}), flush=True)
    A.destroy()
''' % ROOT
for g in (512, 768, 1024, 2048, 4096):
    env = dict(os.environ, SGM_ELLCB_GRID=str(g))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(p.stdout, end="")
    if p.returncode:
        print(p.stderr[-500:])
