cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ildu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ildu -- python tools/ildu_bench.py 600 > gpurun_out/prof_ildu.log 2>&1
tail -1 gpurun_out/prof_ildu.log | cut -c1-200
python - <<'PY'
import csv,glob
f=max(glob.glob('gpurun_out/prof_ildu/*/*kernel_stats.csv'))
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:70].ljust(72), r['Calls'].rjust(6), "%10.1f us"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
