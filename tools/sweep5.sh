cd $GRAFT_REPO_ROOT/tools
./spmv_bench 3162 3162 100
SGM_SGM_DEBUG=1 ./spmv_bench 1000 1000 20
./spmv_bench 215 215 50 7
SGM_SPMV_CFG=256,2,1,0,1,3 ./spmv_bench 215 215 50 7
SGM_CSR_DO=0 ./spmv_bench 215 215 50 7
cd .. && python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
