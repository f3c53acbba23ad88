cd $GRAFT_REPO_ROOT/tools
SGM_CSR_DO=0 ./spmv_bench 3162 3162 100
for cfg in 256,2,1,2048,1,4 256,2,1,2048,1,3 256,2,1,2048,1,2 512,2,1,2048,1,3 512,2,1,2048,1,2 512,2,1,1024,1,3 256,2,1,2048,0,4 256,2,1,4096,1,4; do
  SGM_SPMV_CFG=$cfg ./spmv_bench 3162 3162 100
done
SGM_CSR_DO=0 ./spmv_bench 215 215 50 7
./spmv_bench 215 215 50 7
cd .. && python -m pytest tests -m gpu -q -x 2>&1 | tail -5
