cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06/flaky2
# (a) the parent pytest process holds a GPU context (one boundary test first), then the 8-rank case
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_multirank.py -q --timeout=600 -k "validated or (share_one_gpu and 8-laplace3d)" --basetemp=/tmp/fl_a > gpurun_out/r06/flaky2/a.log 2>&1; echo a=$?
grep -h "passed\|failed\|AssertionError: (" gpurun_out/r06/flaky2/a.log | cut -c1-200
# (b) the same, the parent idle with a context AND 2 GiB allocated by torch
timeout 900 python - <<'P' > gpurun_out/r06/flaky2/b.log 2>&1
import torch, subprocess, sys
x = torch.zeros(1 << 28, device="cuda")          # 2 GiB held by the parent
torch.cuda.synchronize()
rc = subprocess.call([sys.executable, "-m", "pytest", "tests/test_gpu_multirank.py", "-q", "--timeout=600", "-k", "share_one_gpu and 8-laplace3d", "--basetemp=/tmp/fl_b"])
print("inner rc", rc)
P
echo b=$?
grep -h "passed\|failed\|AssertionError: (\|inner rc" gpurun_out/r06/flaky2/b.log | cut -c1-200
