#!/bin/bash
mkdir -p gpurun_out/survey gpurun_out/fuzz
timeout 600 python3 tools/probes/colblock_threshold.py > gpurun_out/survey/colblock_threshold_after.jsonl 2> gpurun_out/survey/cb_err.log
cut -c1-260 gpurun_out/survey/colblock_threshold_after.jsonl | grep -v '"n": 250000\|"n": 500000'
timeout 900 python -m pytest tests -q -m gpu -x -k "scattered or column_blocked or colblock or c4 or ellpack or fuzzer" --timeout=600 2>&1 | tail -3
timeout 600 python3 tests/fuzz_formats.py 150 600000 > gpurun_out/fuzz/fuzz_600000.log 2>&1; echo "fuzz_formats=$?"; tail -1 gpurun_out/fuzz/fuzz_600000.log | cut -c1-200
