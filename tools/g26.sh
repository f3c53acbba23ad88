cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 1500 python -m pytest tests/test_gpu_coop_cg.py -q -m gpu --timeout=900 -x > gpurun_out/r04/t_bicg.log 2>&1; echo rc=$?
tail -5 gpurun_out/r04/t_bicg.log
export NXS=100,181,256,316,500,700,1000 SOLVERS=bicgstab KRYLOV_GRAPH=1 NO_C1=1
echo "== default"; timeout 300 python tools/cg_small.py 2>&1 | grep '^{'
echo "== stream"; NXS=100,181,256 SGM_CG_COOP_STREAM=1 timeout 300 python tools/cg_small.py 2>&1 | grep '^{'
echo "== rmax1 beyond"; NXS=256,316,500 SGM_CG_COOP_RMAX=1 timeout 300 python tools/cg_small.py 2>&1 | grep '^{'
echo "== rmax2 "; NXS=316,500,700 SGM_CG_COOP_RMAX=2 timeout 300 python tools/cg_small.py 2>&1 | grep '^{'
