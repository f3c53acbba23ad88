# the full GPU suite + smoke() in one GPU-box visit (gpurun -- bash tools/gpu_suite.sh); tools/gpu_round.sh adds the profile round
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3300 python -m pytest tests -q -m gpu --durations=15 --timeout=900 > gpurun_out/r06/gpu_tests.log 2>&1; echo gputests=$?
tail -30 gpurun_out/r06/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
