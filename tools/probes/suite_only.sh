#!/bin/bash
mkdir -p gpurun_out/suite
timeout 3000 python -m pytest tests -q -m gpu -x --timeout=900 > gpurun_out/suite/gpu_tests.log 2>&1; echo gputests=$?
grep -n "passed\|failed" gpurun_out/suite/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
