#!/bin/bash
# one campaign of every fuzzer on new seeds (profiles/r05/fuzz_campaigns.txt)
mkdir -p gpurun_out/fuzz
S=${1:-500000}
timeout 900 python3 tests/fuzz_formats.py 170 $S > gpurun_out/fuzz/fuzz_$S.log 2>&1; echo "formats rc=$?"; tail -1 gpurun_out/fuzz/fuzz_$S.log | cut -c1-200
timeout 900 python3 tests/fuzz_solvers.py 200 $S > gpurun_out/fuzz/solvers_$S.log 2>&1; echo "solvers rc=$?"; tail -1 gpurun_out/fuzz/solvers_$S.log | cut -c1-300
timeout 600 python3 tests/fuzz_graphs.py 90 $S > gpurun_out/fuzz/graphs_$S.log 2>&1; echo "graphs rc=$?"; tail -1 gpurun_out/fuzz/graphs_$S.log | cut -c1-200
timeout 900 python3 tests/fuzz_ranks.py 25 1 $S > gpurun_out/fuzz/ranks_$S.log 2>&1; echo "ranks rc=$?"; tail -1 gpurun_out/fuzz/ranks_$S.log | cut -c1-200
grep -h "MISMATCH\|FAILED" gpurun_out/fuzz/*_$S.log | head -10 | cut -c1-300
