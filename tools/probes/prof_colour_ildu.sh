# per-kernel times of ILDU(0)-PCG on the colour-ordered 3162^2 grid (row-space level sweeps)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_col
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_col -o col -- python3 tools/ildu_bench.py ${1:-3162} ildu0 colour > gpurun_out/prof_col.log 2>&1
python3 - <<'PY'
import csv,glob
for fn in glob.glob('gpurun_out/prof_col/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(fn)))[:10]:
        print(r['Name'][:100], r['Calls'], r['AverageNs'], r['Percentage'])
PY
grep "^{" gpurun_out/prof_col.log
