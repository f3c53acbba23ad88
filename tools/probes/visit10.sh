cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v10
timeout 3300 python -m pytest tests -q -m gpu --timeout=900 > gpurun_out/r05_v10/gpu_tests.log 2>&1; echo gputests=$?
grep -E "passed|failed|^FAILED" gpurun_out/r05_v10/gpu_tests.log | tail -15
