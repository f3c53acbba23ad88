#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -k "rank_fuzz" > gpurun_out/fuzz/rank_fuzz_tests.log 2>&1; echo "tests=$?"
tail -30 gpurun_out/fuzz/rank_fuzz_tests.log | cut -c1-300
