cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_boundary.py -q -x --timeout=900 -k "8 or refuses or gpus_2 or fortran or boundary" > gpurun_out/r06/v1_multirank.log 2>&1; echo multirank=$?
tail -15 gpurun_out/r06/v1_multirank.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x --timeout=900 -k "tolerance_is_live or solver_semantics or fortran_host or reference_side" > gpurun_out/r06/v1_tol.log 2>&1; echo tol=$?
tail -15 gpurun_out/r06/v1_tol.log
timeout 900 python bench.py > gpurun_out/r06/v1_bench.json 2> gpurun_out/r06/v1_bench.err; echo bench=$?
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06/v1_bench.json').read().strip().splitlines()[-1])
print(list(d['roofline'].items())[:26])
P
