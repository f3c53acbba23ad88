# one GPU-box visit: multi-rank tests first, then the full gpu suite, then the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r02/multirank.log 2>&1; echo multirank=$?
tail -15 gpurun_out/r02/multirank.log
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_multirank.py > gpurun_out/r02/gpu_tests.log 2>&1; echo gputests=$?
tail -5 gpurun_out/r02/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; echo bench=$?
tail -3 gpurun_out/r02/bench.err; cat gpurun_out/r02/bench.json
