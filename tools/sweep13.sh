cd $GRAFT_REPO_ROOT/tools
for vc in 2048,2048 1024,2048 512,2048 1024,1024 1536,2048; do
echo "vec $vc"; SGM_VEC_CFG=$vc SGM_BENCH_CG=100 ./spmv_bench 3162 3162 20 | grep -E "CG rep 2"
SGM_VEC_CFG=$vc SGM_BENCH_CG=100 ./spmv_bench 300 300 20 7 | grep -E "CG rep 2"
done
