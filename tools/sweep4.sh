cd $GRAFT_REPO_ROOT/tools
./spmv_bench 3162 3162 100
for cfg in 256,2,1,2048,1,0 256,2,1,2048,1,3 512,2,1,2048,1,4 512,2,1,1024,1,4 512,2,1,1024,1,3 256,2,1,1536,1,4 256,2,1,1792,1,4; do
  SGM_SPMV_CFG=$cfg ./spmv_bench 215 215 50 7
done
cd .. && python -m pytest tests -m gpu -q 2>&1 | tail -5
