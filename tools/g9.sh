cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout=150 -k "slice_schedule" > gpurun_out/r04/t_sched.log 2>&1; echo rc=$?
grep -n "Timeout\|File \"/\|line [0-9]* in\|set_option\|matvec\|solve" gpurun_out/r04/t_sched.log | head -40
tail -5 gpurun_out/r04/t_sched.log
