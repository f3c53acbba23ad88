cd $GRAFT_REPO_ROOT
for g in 3162 -160 -200; do timeout 900 python tools/ildu_bench.py $g ildu0 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330; done
