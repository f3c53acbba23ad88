cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 900 python -m pytest tests/test_gpu_coop_cg.py -q -m gpu --timeout=600 -x > gpurun_out/r04/t_coop.log 2>&1; echo rc=$?
tail -5 gpurun_out/r04/t_coop.log
echo "== coop probe"; NXS=100,300,500,1000 timeout 300 python tools/probes/coop_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/coop_probe_xl.jsonl
export NXS=100,256,300,316,362,500,700,1000 SOLVERS=cg KRYLOV_GRAPH=1 NO_C1=1
echo "== default"; timeout 300 python tools/cg_small.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/coop_xl_default.jsonl
