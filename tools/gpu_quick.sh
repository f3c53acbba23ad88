cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "$1" 2>&1 | tail -8
