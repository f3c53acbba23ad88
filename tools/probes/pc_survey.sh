#!/bin/bash
mkdir -p gpurun_out/survey
timeout 1500 python3 tools/pc_survey.py ${1:-240} ${2:-980000} > gpurun_out/survey/pc_survey.jsonl 2> gpurun_out/survey/pc_err.log; echo "rc=$?"
grep "^#" gpurun_out/survey/pc_survey.jsonl
grep -c "^{" gpurun_out/survey/pc_survey.jsonl
tail -3 gpurun_out/survey/pc_err.log
