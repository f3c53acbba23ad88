# PMC passes for the C4 ELLPACK gather kernel (run on the GPU box): HBM-side bytes, L2 hit rate, request counts.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_c4
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  tag=$(echo $C | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- python tools/bench_configs.py --configs c4 ${C4_ARGS:-} > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_c4/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_ell" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        print(k, c, "n=%d mean=%.4g" % (len(v), sum(v) / len(v)))
PY
