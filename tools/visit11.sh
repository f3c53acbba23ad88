cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1500 python3 tests/fuzz_ranks.py 24 1 700000 > gpurun_out/r06/fuzz_ranks_700000.log 2>&1; echo rc=$?
tail -8 gpurun_out/r06/fuzz_ranks_700000.log | cut -c1-300
