#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 1500 python3 tests/fuzz_graphs.py ${1:-300} ${2:-700000} > gpurun_out/fuzz/graphs_${2:-700000}.log 2>&1; echo "rc=$?"
tail -1 gpurun_out/fuzz/graphs_${2:-700000}.log | cut -c1-400
grep "MISMATCH\|Traceback\|Error" gpurun_out/fuzz/graphs_${2:-700000}.log | head -20 | cut -c1-300
