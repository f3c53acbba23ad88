cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pipelines_at_size" --durations=5 2>&1 | tail -12
