cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r04/t_boundary.log 2>&1; echo boundary=$?
tail -15 gpurun_out/r04/t_boundary.log
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "stuck or typed_plainly or torch_distributed_run or rejected" > gpurun_out/r04/t_stall.log 2>&1; echo stall=$?
tail -25 gpurun_out/r04/t_stall.log
(time timeout 900 python bench.py > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err); echo bench=$?
tail -5 gpurun_out/r04/bench1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench1.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:v for k,v in r.items() if not isinstance(v,(dict,str))})
print(json.dumps(d.get('pcg_time_to_solution'))[:1500])
print(json.dumps(d.get('c4'))[:800])
PY
