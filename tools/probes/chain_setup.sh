#!/bin/bash
mkdir -p gpurun_out/fuzz gpurun_out/survey
SGM_PC_TIMING=1 timeout 600 python3 tools/probes/chain_setup.py 2>&1 | grep -E "factorisation|n=|apply" | cut -c1-200
timeout 1200 python -m pytest tests -q -m gpu -x -k "ildu or pc_info or reorder or fuzzer or golden or precond or ldu" --timeout=600 2>&1 | tail -3
timeout 900 python3 tests/fuzz_solvers.py 150 130000 > gpurun_out/fuzz/solvers_130000.log 2>&1; echo "fuzz_solvers=$?"; tail -1 gpurun_out/fuzz/solvers_130000.log | cut -c1-300
timeout 600 python3 tools/pc_survey.py 100 980000 > gpurun_out/survey/pc_survey_after.jsonl 2> gpurun_out/survey/pc_err.log; grep "^#" gpurun_out/survey/pc_survey_after.jsonl
