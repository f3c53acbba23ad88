cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh 2>&1 | tail -40
KRYLOV_GRAPH=1 timeout 600 python tools/cg_small.py > gpurun_out/prof_round/cg_small.jsonl 2>&1
SGM_CG_COOP=0 KRYLOV_GRAPH=1 timeout 600 python tools/cg_small.py > gpurun_out/prof_round/cg_small_launch_loop.jsonl 2>&1
tail -4 gpurun_out/prof_round/bench.err
