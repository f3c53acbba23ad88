cd $GRAFT_REPO_ROOT/tools
export SGM_BENCH_CG=100
./spmv_bench 3162 3162 20
SGM_VEC_CFG=2048,2048 ./spmv_bench 3162 3162 20 | grep CG
SGM_VEC_CFG=4096,65536 ./spmv_bench 3162 3162 20 | grep CG
SGM_VEC_CFG=1024,65536 ./spmv_bench 3162 3162 20 | grep CG
SGM_VEC_CFG=2048,8192 ./spmv_bench 3162 3162 20 | grep CG
unset SGM_BENCH_CG
cd .. && python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log
