#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 1500 python3 tests/fuzz_solvers.py ${1:-480} ${2:-5000} > gpurun_out/fuzz/solvers_${2:-5000}.log 2>&1; echo "rc=$?" >> gpurun_out/fuzz/solvers_${2:-5000}.log
tail -3 gpurun_out/fuzz/solvers_${2:-5000}.log | cut -c1-400
grep -c "^seed" gpurun_out/fuzz/solvers_${2:-5000}.log
grep "MISMATCH\|Error\|error\|Traceback" gpurun_out/fuzz/solvers_${2:-5000}.log | head -30 | cut -c1-400
