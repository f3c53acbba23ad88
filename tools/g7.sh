cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout=600 --durations=5 -k "full_size_properties or default_dot_order or ell_column_blocked" 2>&1 | tail -25
cat > /tmp/c4sweep.py <<'PY'
import sys, json, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
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
A = sg.ellpack_matrix(n, n, node, val)
y0 = None
for cols, rows in ((16384, 512), (20480, 512), (16384, 608), (20480, 608), (18432, 608)):
    A.set_option("ell_colblock_cols", cols); A.set_option("ell_colblock_rows", rows)
    us = timed(A)
    if y0 is None: y0 = y.clone()
    print(json.dumps({"cols": cols, "rows": rows, "kernel": A.kernel, "us": round(us, 1), "moved_GB": round(A.footprint()[1] / 1e9, 3), "same_bits": bool(torch.equal(y, y0))}), flush=True)
PY
timeout 600 python /tmp/c4sweep.py 2>&1 | grep '^{' | tee gpurun_out/r04/c4_cols_rows_sweep.jsonl
