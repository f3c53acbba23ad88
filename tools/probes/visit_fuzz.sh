#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 1500 python3 tests/fuzz_formats.py ${1:-600} ${2:-1000} > gpurun_out/fuzz/fuzz_${2:-1000}.log 2>&1; echo "rc=$?" >> gpurun_out/fuzz/fuzz_${2:-1000}.log
tail -4 gpurun_out/fuzz/fuzz_${2:-1000}.log | cut -c1-400
grep -c "^seed" gpurun_out/fuzz/fuzz_${2:-1000}.log
grep "MISMATCH\|Error\|error" gpurun_out/fuzz/fuzz_${2:-1000}.log | head -20 | cut -c1-400
