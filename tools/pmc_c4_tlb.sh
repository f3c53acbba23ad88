cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for S in "-1,512,0" "0,256,0" "65536,256,0"; do
  OUT=gpurun_out/pmc_tlb_$(echo $S | tr ',' '_'); rm -rf $OUT
  SGM_ELLCB_PIPE=${PIPE:-0} timeout 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum --output-format csv -d $OUT -- python tools/ellcb_band.py $S > $OUT.log 2>&1
  grep '^{' $OUT.log | cut -c1-80
  python - "$OUT" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_ellcb_sum" in r["Kernel_Name"] or "k_ellcb_mul" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void sgm::", "")[:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("  ", k, c, "launches=%d total_per_product=%.5g" % (len(v), sum(v) / 24))
PY
done
