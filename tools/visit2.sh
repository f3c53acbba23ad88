cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -x --timeout=900 -k "share_one_gpu and (2 or 3)" > gpurun_out/r06/v2_multirank.log 2>&1; echo multirank=$?
grep -n "Fatal" -A8 gpurun_out/r06/v2_multirank.log | head -60
tail -8 gpurun_out/r06/v2_multirank.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --timeout=900 -k "tolerance_is_live or ildu_on_ellpack or preconditioners_golden or solvers_golden or reference_side or fortran_host or ellpack" > gpurun_out/r06/v2_parity.log 2>&1; echo parity=$?
tail -15 gpurun_out/r06/v2_parity.log
