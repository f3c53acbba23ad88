cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 1200 python -m pytest tests/test_gpu_reorder.py tests/test_gpu_coop_cg.py -q -m gpu --timeout=900 -x > gpurun_out/r04/t_pcg.log 2>&1; echo rc=$?
tail -25 gpurun_out/r04/t_pcg.log
export NXS=45,64,100 SOLVERS=cg KRYLOV_GRAPH=1
timeout 300 python tools/cg_small.py 2>&1 | grep -v amdgpu.ids
timeout 900 python bench.py > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench2.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'])
r=d['roofline']
for k,v in r.items():
    if isinstance(v,(int,float,str)) or v is None: print(' ',k,v)
print(json.dumps(d.get('pcg_time_to_solution'),indent=1))
PY
tail -5 gpurun_out/r04/bench2.err
