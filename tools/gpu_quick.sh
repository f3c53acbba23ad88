cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -25 gpurun_out/r02/quick.log
