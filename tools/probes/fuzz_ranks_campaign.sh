#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 2400 python3 tests/fuzz_ranks.py ${1:-60} ${2:-2} 400000 > gpurun_out/fuzz/ranks_400000.log 2>&1; echo "rc=$?"
tail -12 gpurun_out/fuzz/ranks_400000.log | cut -c1-400
