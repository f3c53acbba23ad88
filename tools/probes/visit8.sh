cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v8
timeout 1500 python -m pytest tests/test_gpu_reorder.py tests/test_gpu_parity.py -x -q --timeout=900 -k "reorder or colour or ildu or pc or scattered or partition" > gpurun_out/r05_v8/t.log 2>&1; echo t=$?; tail -8 gpurun_out/r05_v8/t.log
bash tools/probes/ildu_parts_stats.sh > /dev/null 2>&1
grep -E "^\{|all kernels|k_trsv|k_csr_sl" gpurun_out/r05_parts/summary.txt | cut -c1-200
