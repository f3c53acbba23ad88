cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v6
SGM_TRACE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --timeout=600 -k "scattered" -s 2>&1 | grep -v "^$" | tail -30
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q --timeout=900 -k "forms_p_halo or share_one_gpu or scattered" > gpurun_out/r05_v6/t.log 2>&1; echo t=$?; tail -12 gpurun_out/r05_v6/t.log
