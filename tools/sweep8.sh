cd $GRAFT_REPO_ROOT
./tools/spmv_bench 3162 3162 50
./tools/spmv_bench 215 215 30 7
for cfg in 8,1,2048 8,0,2048 16,1,2048 4,1,2048 8,1,4096 16,1,4096 16,1,1024; do
  echo "ELL cfg $cfg"; SGM_ELL_CFG=$cfg python tools/bench_configs.py --configs c4 --scale 0.4 | cut -c1-220
done
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed"
