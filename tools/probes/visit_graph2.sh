#!/bin/bash
mkdir -p gpurun_out/graph
timeout 900 python -m pytest tests -m gpu -x -q -k "replayed_iteration_groups" > gpurun_out/graph/tests2.log 2>&1; echo "tests=$?" >> gpurun_out/graph/tests2.log
tail -30 gpurun_out/graph/tests2.log
