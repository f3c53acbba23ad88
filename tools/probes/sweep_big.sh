#!/bin/bash
# knob sweep of the offset-dict kernel in the streaming (cache-cold) regime
spec=${1:-2d:8000x8000}
for cfg in "256,2,1,0,1,0" "256,2,1,1024,1,0" "256,2,1,1536,1,0" "256,2,1,2048,1,0" "256,2,1,4096,1,0" "256,2,1,16384,1,0" "256,2,1,0,0,0" "256,2,1,0,2,0" "256,2,1,0,1,1" "256,2,1,0,1,3" "256,2,1,0,1,4"; do
  echo -n "CFG $cfg  "
  SGM_SPMV_CFG=$cfg python tools/probes/size_sweep.py $spec 2>&1 | grep spec | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["dict"], d["int32"])'
done
