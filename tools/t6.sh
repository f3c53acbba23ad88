cd $GRAFT_REPO_ROOT
for nw in 2048 256 64 0; do echo "narrow=$nw"; SGM_TRSV_NARROW=$nw python tools/ildu_bench.py 1000 | tail -1 | cut -c1-200; done
