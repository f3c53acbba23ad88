#!/bin/bash
mkdir -p gpurun_out/survey
timeout 1500 python3 tools/perf_survey.py ${1:-300} ${2:-900000} ${3:-} > gpurun_out/survey/perf_survey_${3:-csr}.jsonl 2> gpurun_out/survey/err.log; echo "rc=$?"
grep "^#" gpurun_out/survey/perf_survey_${3:-csr}.jsonl
grep -c "^{" gpurun_out/survey/perf_survey_${3:-csr}.jsonl
tail -3 gpurun_out/survey/err.log
