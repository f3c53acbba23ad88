cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06/flaky3
timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_coop_cg.py tests/test_gpu_multirank.py -q --timeout=600 -k "not fuzz and not bench and not stuck and not fortran and not refuses and not rejected and not two_gpus and not dist_overhead" > gpurun_out/r06/flaky3/c.log 2>&1; echo c=$?
grep -h "passed\|failed\|AssertionError: (\|FAILED" gpurun_out/r06/flaky3/c.log | cut -c1-200
ls gpurun_out/rank_retries 2>/dev/null | head
