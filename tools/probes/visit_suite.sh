#!/bin/bash
# the full GPU suite + smoke (no profile round)
mkdir -p gpurun_out/suite
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 --timeout=900 > gpurun_out/suite/gpu_tests.log 2>&1; echo gputests=$?
tail -14 gpurun_out/suite/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
