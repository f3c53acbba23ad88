cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export PROBE_CHECK=0
for sc in "0,64" "1,64" "1,32" "1,16" "1,8" "1,4"; do
  SGM_SLICE_SCHED=$sc timeout 300 python tools/probes/sched_probe.py 3d:464 3d:464x464x58 3d:300 2>/dev/null
done > gpurun_out/r04/sched_sweep.jsonl
cat gpurun_out/r04/sched_sweep.jsonl
for sc in "0,64" "1,16" "1,8"; do
  echo "== FETCH_SIZE, SGM_SLICE_SCHED=$sc"
  SGM_SLICE_SCHED=$sc PROBE_REPS=3 KERNEL=k_csr_sl OUT=gpurun_out/pmc_sched bash tools/pmc_fetch.sh python3 tools/probes/sched_probe.py 3d:464
done > gpurun_out/r04/sched_sweep_fetch.txt 2>&1
cat gpurun_out/r04/sched_sweep_fetch.txt
