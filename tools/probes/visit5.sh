cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v5
timeout 3300 python -m pytest tests -q -m gpu --durations=15 --timeout=900 > gpurun_out/r05_v5/gpu_tests.log 2>&1; echo gputests=$?
tail -25 gpurun_out/r05_v5/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_v5/stats_ildu_parts -- python3 tools/probes/ildu_parts.py 3162 8 100 > gpurun_out/r05_v5/ildu_parts.log 2>&1
grep '^{' gpurun_out/r05_v5/ildu_parts.log
find gpurun_out/r05_v5 -name "*kernel_trace.csv" -delete
ls gpurun_out/r05_v5/stats_ildu_parts/*/ | head
