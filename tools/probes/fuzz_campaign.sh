#!/bin/bash
# one campaign of each fuzzer with the final gates (new seeds)
mkdir -p gpurun_out/fuzz
timeout 1500 python3 tests/fuzz_solvers.py ${1:-420} 100000 > gpurun_out/fuzz/solvers_100000.log 2>&1; echo "solvers rc=$?"
tail -1 gpurun_out/fuzz/solvers_100000.log | cut -c1-300
grep "MISMATCH\|Traceback" gpurun_out/fuzz/solvers_100000.log | head -20 | cut -c1-300
timeout 1500 python3 tests/fuzz_formats.py ${2:-300} 100000 > gpurun_out/fuzz/fuzz_100000.log 2>&1; echo "formats rc=$?"
tail -1 gpurun_out/fuzz/fuzz_100000.log | cut -c1-300
