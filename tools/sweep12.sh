cd $GRAFT_REPO_ROOT/tools
SGM_BENCH_CG=100 ./spmv_bench 3162 3162 20 | grep -E "CG rep 2"
SGM_VEC_CFG=1024,2048 SGM_BENCH_CG=100 ./spmv_bench 3162 3162 20 | grep -E "CG rep 2"
SGM_VEC_CFG=4096,4096 SGM_BENCH_CG=100 ./spmv_bench 3162 3162 20 | grep -E "CG rep 2"
SGM_BENCH_CG=100 ./spmv_bench 300 300 20 7 | grep -E "CG rep 2"
