#!/bin/bash
# GMRES after the small dense steps moved to one wave: the tests that solve with it, then C3's figures
mkdir -p gpurun_out/gmres
timeout 1500 python -m pytest tests -m gpu -x -q -k "gmres or Gmres or solver or dist or multirank" > gpurun_out/gmres/tests.log 2>&1; echo "tests=$?" >> gpurun_out/gmres/tests.log
tail -5 gpurun_out/gmres/tests.log
timeout 600 python3 tools/bench_configs.py --configs c3 > gpurun_out/gmres/c3.jsonl 2> gpurun_out/gmres/c3.err
timeout 600 python3 tools/bench_configs.py --configs c3 >> gpurun_out/gmres/c3.jsonl 2>> gpurun_out/gmres/c3.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gmres/stats -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --configs c3 > $GRAFT_REPO_ROOT/gpurun_out/gmres/prof.log 2>&1
