cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v2
timeout 1200 python -m pytest tests/test_gpu_coop_cg.py tests/test_gpu_multirank.py -x -q --timeout=900 > gpurun_out/r05_v2/t1.log 2>&1; echo t1=$?; tail -5 gpurun_out/r05_v2/t1.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q --timeout=900 -k "partition or rccl or dot_order" > gpurun_out/r05_v2/t2.log 2>&1; echo t2=$?; tail -5 gpurun_out/r05_v2/t2.log
timeout 900 python bench.py --no-c3 --no-c4 --no-pcg --no-cpu --no-variants > gpurun_out/r05_v2/bench.json 2> gpurun_out/r05_v2/bench.err; echo bench=$?
tail -3 gpurun_out/r05_v2/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_v2/bench.json"))
print(json.dumps(d.get("c5_parts_model"), indent=1)[:3000])
print({k: v for k, v in d["roofline"].items() if k.startswith(("c5_", "ceiling", "c2_frac"))})
PY
