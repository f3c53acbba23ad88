# per-kernel times of the C4 product for several band settings (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_c4_band
rm -rf $OUT; mkdir -p $OUT
for S in ${C4_SETTINGS:--1,512,0 0,256,0 0,512,0 524288,256,0}; do
  tag=$(echo $S | tr ',' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python tools/ellcb_band.py $S > $OUT/$tag.log 2>&1
  echo "== $S rc=$?"; grep '^{' $OUT/$tag.log
  python - "$OUT/$tag" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_ellcb_mul" in r["Name"] or "k_ellcb_sum" in r["Name"]:
            print("   ", r["Name"].split("(")[0][:70], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
done
