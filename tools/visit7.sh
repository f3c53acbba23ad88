cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06/flaky
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x --timeout=500 -k "share_one_gpu and 8-laplace3d" --basetemp=/tmp/flaky$i > gpurun_out/r06/flaky/run$i.log 2>&1; rc=$?
  echo "run $i rc=$rc"
  for r in 0 1 2 3 4 5 6 7; do f=$(ls /tmp/flaky$i/*/rank$r.json 2>/dev/null | head -1); [ -n "$f" ] && python - "$f" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
s=d.get("solves",{})
print("   rank",d["rank"],"ok",d["ok"],{k:(v.get("iterations"),v.get("oracle_iterations")) for k,v in s.items() if isinstance(v,dict) and "oracle_iterations" in v}, (d.get("error") or "")[-200:].replace("\n"," "))
P
  done
done 2>&1 | tee gpurun_out/r06/flaky/summary.txt
