cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v9
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q --timeout=900 -k "long_row or part_by_part or scattered or general" > gpurun_out/r05_v9/t.log 2>&1; echo t=$?; tail -12 gpurun_out/r05_v9/t.log
timeout 1200 python tools/general_rows.py 2>&1 | grep '^{' | cut -c1-260 > gpurun_out/r05_v9/general_rows.jsonl; cat gpurun_out/r05_v9/general_rows.jsonl
