# kernel-time sums per CG / colour-ILDU-PCG iteration at C2 size, one part against 8 in-process parts (rocprofv3 --stats of one
# configuration per run: 2 solves of 320 iterations each + setup; beyond 64 iterations the loops replay captured groups)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_parts; rm -rf $OUT; mkdir -p $OUT
for cfg in one:none one:ildu parts:none parts:ildu; do
  tag=$(echo $cfg | tr ':' '_')
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 tools/probes/ildu_parts.py 3162 8 320 $cfg > $OUT/$tag.log 2>&1
  grep '^{' $OUT/$tag.log | cut -c1-220
  python3 - "$OUT/$tag" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("   all kernels: %.1f ms in %d launches; per iteration (640 iterations incl. setup and warm-up): %.1f us, %.1f launches" % (tot / 1e6, calls, tot / 1e3 / 640, calls / 640))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("     %-60s calls %6s avg %8.1f us total %8.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt
