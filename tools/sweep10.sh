cd $GRAFT_REPO_ROOT/tools
./spmv_bench 3162 3162 50
SGM_CSR_DO=0 ./spmv_bench 3162 3162 50
SGM_CSR_DO=0 SGM_CSR_RO=0 ./spmv_bench 3162 3162 50
SGM_CSR_DO=0 ./spmv_bench 215 215 30 7
SGM_CSR_DO=0 SGM_CSR_RO=0 ./spmv_bench 215 215 30 7
SGM_CSR_DO=0 SGM_BENCH_FLUSH=1 ./spmv_bench 3162 3162 50
cd .. && python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | head -5
