# one GPU-box visit: full gpu suite (incl. multi-rank), then the profile round
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python3 -c "import torch; print(torch.cuda.is_available())" | tail -1
timeout 3000 python -m pytest tests -q -m gpu --durations=10 --timeout=900 > gpurun_out/r06/gpu_tests.log 2>&1; echo gputests=$?
tail -5 gpurun_out/r06/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh
