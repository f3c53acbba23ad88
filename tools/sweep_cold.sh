cd $GRAFT_REPO_ROOT/tools
for cfg in 256,2,1,0,1,0 256,2,1,0,1,3 256,2,1,0,1,4 512,2,1,0,1,0 512,2,1,0,1,2 256,2,1,1536,1,0 256,2,1,1792,1,0 256,2,1,0,0,0; do
  SGM_SPMV_CFG=$cfg SGM_BENCH_FLUSH=1 ./spmv_bench 3162 3162 40
  SGM_SPMV_CFG=$cfg ./spmv_bench 300 300 15 7
done
