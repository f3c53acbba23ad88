cd $GRAFT_REPO_ROOT/tools
for pf in 0 1; do
  SGM_SPMV_CFG=256,2,1,0,1,0,$pf ./spmv_bench 3162 3162 50
  SGM_SPMV_CFG=256,2,1,0,1,0,$pf SGM_BENCH_FLUSH=1 ./spmv_bench 3162 3162 50
  SGM_SPMV_CFG=256,2,1,0,1,0,$pf ./spmv_bench 300 300 20 7
  SGM_SPMV_CFG=256,2,1,0,1,0,$pf SGM_CSR_DO=0 ./spmv_bench 3162 3162 50
done
for g in 1280 1536 1792; do SGM_SPMV_CFG=256,2,1,$g,1,0,1 ./spmv_bench 3162 3162 50; SGM_SPMV_CFG=256,2,1,$g,1,0,1 ./spmv_bench 300 300 20 7; done
