cd $GRAFT_REPO_ROOT/tools
for vc in 1024,2048 1024,4096 1024,8192 1024,65536 1024,1024; do
echo "vec $vc"; SGM_VEC_CFG=$vc SGM_BENCH_CG=100 ./spmv_bench 3162 3162 10 | grep -E "CG rep 2"
done
