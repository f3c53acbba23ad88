cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reorder.py -q -x --timeout=900 -k "colour or reorder or partition or fuzzer" > gpurun_out/r06/v5_parity.log 2>&1; echo parity=$?
tail -12 gpurun_out/r06/v5_parity.log
timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -x --timeout=900 -k "share_one_gpu or fuzz or fortran" > gpurun_out/r06/v5_multirank.log 2>&1; echo multirank=$?
tail -8 gpurun_out/r06/v5_multirank.log
for a in "one:none" "one:ildu" "parts:none" "parts:ildu"; do timeout 600 python tools/probes/ildu_parts.py 3162 8 320 $a 2>&1 | grep '^{' | cut -c1-400; done | tee gpurun_out/r06/v5_ildu_parts.jsonl
