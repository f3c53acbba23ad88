#!/bin/bash
mkdir -p gpurun_out/survey
timeout 600 python3 tools/probes/long_row_probe.py > gpurun_out/survey/long_row_after.jsonl 2> gpurun_out/survey/long_row.err
cat gpurun_out/survey/long_row_after.jsonl; tail -2 gpurun_out/survey/long_row.err
timeout 900 python -m pytest tests -q -m gpu -x -k "randomised or ragged or fuzzer or kernel or general or long" --timeout=900 2>&1 | tail -3
timeout 600 python3 tests/fuzz_formats.py 120 810000 > gpurun_out/fuzz/fuzz_810000.log 2>&1; echo fuzz_formats=$?; tail -1 gpurun_out/fuzz/fuzz_810000.log | cut -c1-200
