#!/bin/bash
# GPU suite, two minutes of the format fuzzer, then the performance survey
mkdir -p gpurun_out/suite gpurun_out/fuzz gpurun_out/survey
timeout 3000 python -m pytest tests -q -m gpu -x --timeout=900 > gpurun_out/suite/gpu_tests.log 2>&1; echo gputests=$?
tail -4 gpurun_out/suite/gpu_tests.log
timeout 600 python3 tests/fuzz_formats.py 150 800000 > gpurun_out/fuzz/fuzz_800000.log 2>&1; echo fuzz_formats=$?; tail -1 gpurun_out/fuzz/fuzz_800000.log | cut -c1-200
timeout 900 python3 tools/perf_survey.py 240 900000 > gpurun_out/survey/perf_survey_after.jsonl 2> gpurun_out/survey/err.log; echo "survey rc=$?"
grep "^#" gpurun_out/survey/perf_survey_after.jsonl
