cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "composite or lanczos or binding" > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -25 gpurun_out/r02/quick.log
