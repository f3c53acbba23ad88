cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
./tools/probes/wave_sum_probe
echo "== coop probe (one XCD where it fits)"; timeout 300 python tools/probes/coop_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/coop_probe_xl.jsonl
export NXS=100,128,150,181,256,300,316,330,362,500,1000 SOLVERS=cg KRYLOV_GRAPH=1
echo "== default"; timeout 300 python tools/cg_small.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/coop_xl_default.jsonl
timeout 1500 python -m pytest tests/test_gpu_coop_cg.py tests/test_gpu_parity.py -q -m gpu --timeout=600 -x -k "coop or small or golden or dot_order or cg or bicgstab or gmres" > gpurun_out/r04/t_ws.log 2>&1; echo rc=$?
tail -5 gpurun_out/r04/t_ws.log
