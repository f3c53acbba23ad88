cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 900 python -m pytest tests/test_gpu_coop_cg.py -q -m gpu --timeout=600 -x > gpurun_out/r04/t_coop.log 2>&1; echo rc=$?
tail -15 gpurun_out/r04/t_coop.log
export NXS=100,128,150,181,256,300,316,330,362,500,1000 SOLVERS=cg KRYLOV_GRAPH=1 NO_C1=1
echo "== default"; timeout 300 python tools/cg_small.py 2>&1 | tee gpurun_out/r04/coop_xl_default.jsonl
echo "== XCD=0"; SGM_CG_COOP_XCD=0 timeout 300 python tools/cg_small.py 2>&1 | tee gpurun_out/r04/coop_xl_off.jsonl
echo "== STREAM"; NXS=150,256,300 SGM_CG_COOP_STREAM=1 timeout 300 python tools/cg_small.py 2>&1
echo "== RMAX=4"; NXS=150,256,300,316 SGM_CG_COOP_RMAX=4 timeout 300 python tools/cg_small.py 2>&1
echo "== RMAX=3 stream"; NXS=300 SGM_CG_COOP_RMAX=3 SGM_CG_COOP_STREAM=1 timeout 300 python tools/cg_small.py 2>&1
