# Slice schedule A/B on the GPU box: times per configuration, then FETCH_SIZE of the two ends.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sched; rm -rf $OUT; mkdir -p $OUT
SPECS="${SPECS:-3d:464 3d:300 3d:215}"
run() { echo "## SGM_SLICE_SCHED=$1 SGM_SPMV_CFG=$2"; SGM_SLICE_SCHED=$1 SGM_SPMV_CFG=$2 timeout 600 python tools/probes/sched_probe.py $SPECS 2>&1 | grep -v Warning; }
{
run 0 256,2,1,0,1,0
run 1,64 256,2,1,0,1,0
run 1,64 256,2,1,1792,1,0
run 1,64 256,2,1,2048,1,0
run 1,64 256,2,1,4096,1,0
run 1,32 256,2,1,0,1,0
run 1,32 256,2,1,2048,1,0
run 1,128 256,2,1,2048,1,0
run 0 256,2,1,2048,1,0
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
for S in 0 1,64; do
  tag=$(echo $S | tr ',' '_')
  SGM_SLICE_SCHED=$S PROBE_REPS=3 PROBE_CHECK=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python tools/probes/sched_probe.py 3d:464 > $OUT/fetch_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/sched/fetch_*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_csr_sl" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        print(f.split("/")[2], k, c, "n=%d mean=%.5g (x2 KB = %.4g GB read)" % (len(v), sum(v) / len(v), 2 * 1024 * sum(v) / len(v) / 1e9))
PY
