#!/bin/bash
# the full GPU suite + smoke, then two minutes of each fuzzer (no profile round)
mkdir -p gpurun_out/suite gpurun_out/fuzz
timeout 3000 python -m pytest tests -q -m gpu -x --durations=5 --timeout=900 > gpurun_out/suite/gpu_tests.log 2>&1; echo gputests=$?
tail -9 gpurun_out/suite/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 tests/fuzz_formats.py 120 60000 > gpurun_out/fuzz/fuzz_60000.log 2>&1; echo fuzz_formats=$?; tail -1 gpurun_out/fuzz/fuzz_60000.log | cut -c1-200
timeout 600 python3 tests/fuzz_solvers.py 120 60000 > gpurun_out/fuzz/solvers_60000.log 2>&1; echo fuzz_solvers=$?; tail -1 gpurun_out/fuzz/solvers_60000.log | cut -c1-300
