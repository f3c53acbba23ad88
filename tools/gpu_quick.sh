cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column_blocked" > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -3 gpurun_out/r02/quick.log
rm -rf gpurun_out/r02/c4stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/c4stats -- python tools/bench_configs.py --configs c4 > gpurun_out/r02/c4stats.log 2>&1
grep '^{' gpurun_out/r02/c4stats.log | cut -c1-400
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02/c4stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r["Name"][:90].ljust(92), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
