cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())" 2>&1 | tail -1
(time timeout 3000 python -m pytest tests -q -m gpu --durations=15 --timeout=900) > gpurun_out/r04/gpu_tests.log 2>&1; echo gputests=$?
tail -40 gpurun_out/r04/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
