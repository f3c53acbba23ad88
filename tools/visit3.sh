cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
B="--no-cpu --no-pcg --no-c5-parts --no-dist-overhead --no-variants --no-c4 --no-ceilings --no-c5"
for rep in 1 2; do
for mode in tail notail; do
  if [ $mode = notail ]; then export SGM_NO_TAIL=1; else unset SGM_NO_TAIL; fi
  timeout 600 python bench.py $B > gpurun_out/r06/v3_bench_$mode.json 2> gpurun_out/r06/v3_bench_$mode.err; echo bench_$mode=$?
  python - $mode <<'P'
import json,sys
d=json.loads(open('gpurun_out/r06/v3_bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[1], {k:round(r[k],4) for k in ('frac','c2_cg_iters_per_s','c3_bicgstab_iters_per_s','c3_gmres30_iters_per_s')}, d['cg']['final_res2'], d['c3']['bicgstab']['final_res2'])
P
done
done
unset SGM_NO_TAIL
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --timeout=900 -k "last_workgroup_collapse or rccl_single_rank" > gpurun_out/r06/v3_parity.log 2>&1; echo parity=$?
tail -5 gpurun_out/r06/v3_parity.log
