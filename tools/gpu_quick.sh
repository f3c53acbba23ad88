cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_multirank.py -q -x -m gpu -k "host_staged" 2>&1 | tail -25
