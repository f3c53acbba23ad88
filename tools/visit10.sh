cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
SGM_KEEP_RANK_TRACES=1 timeout 900 python -m pytest tests/test_gpu_multirank.py -q --timeout=600 -k "share_one_gpu and 8-laplace3d" > gpurun_out/r06/keep_traces.log 2>&1; echo rc=$?
grep -h "passed\|failed" gpurun_out/r06/keep_traces.log | tail -2
ls gpurun_out/rank_retries | tail -4
