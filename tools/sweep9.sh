cd $GRAFT_REPO_ROOT/tools
for cfg in 256,2,1,0,1,0 256,2,1,0,2,0; do
  SGM_SPMV_CFG=$cfg ./spmv_bench 3162 3162 50
  SGM_SPMV_CFG=$cfg ./spmv_bench 215 215 30 7
  SGM_SPMV_CFG=$cfg ./spmv_bench 464 464 20 7
done
