# HBM bytes per launch of the row-space level sweeps (k_trsv_rows*) and the other kernels of a PCG iteration on the colour-ordered 3162^2 grid: FETCH_SIZE and
# WRITE_SIZE in separate passes (gfx950: read bytes = 2 * FETCH_SIZE * 1024 for wide streaming reads, WRITE_SIZE in KB exact)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_col_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_col_$c -- python3 tools/ildu_bench.py 3162 ildu0 colour > gpurun_out/pmc_col_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for fn in glob.glob(f"gpurun_out/pmc_col_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c and ("k_trsv_rows" in r["Kernel_Name"] or "k_csr_sl" in r["Kernel_Name"] or "k_elem" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(int")[0].split("(long")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c] = (len(v), sum(v) / len(v))
for k, d in sorted(out.items()):
    f = d.get("FETCH_SIZE", (0, 0.0)); w = d.get("WRITE_SIZE", (0, 0.0))
    print(f"{k[:70]:70s} launches {f[0]:6d}  read {2 * f[1] * 1024 / 1e6:8.1f} MB  written {w[1] * 1024 / 1e6:8.1f} MB per launch")
PY
rm -rf gpurun_out/pmc_col_FETCH_SIZE gpurun_out/pmc_col_WRITE_SIZE      # (the raw per-dispatch tables: tens of MB)
