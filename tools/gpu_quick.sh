cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "column_blocked or full_size_c4" 2>&1 | tail -3
