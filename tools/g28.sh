cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 2700 python -m pytest tests -q -m gpu --timeout=900 --durations=6 > gpurun_out/r04/gpu_tests.log 2>&1; echo gputests=$?
tail -12 gpurun_out/r04/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
