cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python tools/dot_order_region.py > gpurun_out/r06/dot_order_region.md 2> gpurun_out/r06/dot_order_region.err; echo region=$?
head -40 gpurun_out/r06/dot_order_region.md
