# FETCH_SIZE of one command (one pass): HBM-side read bytes per launch for kernels matching $KERNEL (gfx950: x 2 x 1024).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/pmc_fetch_one}; rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/g -- "$@" > $OUT/g.log 2>&1; echo rc=$?
python - <<PY
import csv, glob, collections, os
kern = os.environ.get("KERNEL", "k_csr")
for f in sorted(glob.glob("$OUT/g/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, "launches=%d FETCH_SIZE mean=%.5g KB -> %.4g GB read per launch" % (len(v), sum(v) / len(v), 2 * 1024 * sum(v) / len(v) / 1e9))
PY
