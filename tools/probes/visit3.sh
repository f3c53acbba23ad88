cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v3
timeout 1500 python -m pytest tests/test_gpu_reorder.py tests/test_gpu_coop_cg.py -x -q --timeout=900 > gpurun_out/r05_v3/t1.log 2>&1; echo t1=$?; tail -15 gpurun_out/r05_v3/t1.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q --timeout=900 -k "partition or ell_column or ildu or pc" > gpurun_out/r05_v3/t2.log 2>&1; echo t2=$?; tail -15 gpurun_out/r05_v3/t2.log
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_boundary.py -x -q --timeout=900 > gpurun_out/r05_v3/t3.log 2>&1; echo t3=$?; tail -15 gpurun_out/r05_v3/t3.log
