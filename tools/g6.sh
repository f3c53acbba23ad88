cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_coop_cg.py -q -m gpu --timeout=240 2>&1 | tail -15
for r in 0 1 2 4; do
  echo "== SGM_CG_COOP_RMAX=$r"
  SGM_CG_COOP_RMAX=$r KRYLOV_GRAPH=1 timeout 600 python tools/cg_small.py 2>&1 | grep '"cg"' | grep -v '"nx": 32,'
done > gpurun_out/r04/cg_coop_rmax.txt
cat gpurun_out/r04/cg_coop_rmax.txt
