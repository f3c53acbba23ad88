#!/bin/bash
# in-process partitions + two-level ILDU through the replayed groups (hipGraph): parity tests, then the C2-size probe
mkdir -p gpurun_out/graph
timeout 1500 python -m pytest tests -m gpu -x -q -k "partition or parts or block_jacobi or graph or reorder or multirank" > gpurun_out/graph/tests.log 2>&1; echo "tests=$?" >> gpurun_out/graph/tests.log
tail -5 gpurun_out/graph/tests.log
for g in graph nograph; do
  timeout 600 python3 tools/probes/ildu_parts.py 3162 8 640 "" $g >> gpurun_out/graph/parts_$g.jsonl 2>> gpurun_out/graph/err.log
done
cat gpurun_out/graph/parts_*.jsonl | cut -c1-260
