#!/usr/bin/env python3
"""Condense gpurun_out/prof_round (tools/profile_round.sh) into profiles/<tag>/."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/prof_round", os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
ks = max(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(ks, os.path.join(dst, "kernel_stats.csv"))
cfgs = glob.glob(os.path.join(src, "stats_configs", "*", "*kernel_stats.csv"))
if cfgs:     # tools/bench_configs.py --configs c3,c4,c5 (one process, all three configs)
    shutil.copy(max(cfgs, key=os.path.getmtime), os.path.join(dst, "kernel_stats_configs_c3_c4_c5.csv"))
    if os.path.exists(os.path.join(src, "configs.jsonl")):
        shutil.copy(os.path.join(src, "configs.jsonl"), os.path.join(dst, "configs_c3_c4_c5.jsonl"))
ild = glob.glob(os.path.join(src, "stats_ildu", "*", "*kernel_stats.csv"))
if ild:      # ILDU(0)-PCG on the 1000^2 grid (tools/ildu_bench.py 1000 ildu0)
    shutil.copy(max(ild, key=os.path.getmtime), os.path.join(dst, "kernel_stats_ildu_pcg_1000x1000.csv"))
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 5 --warmup 1 --cg-steps 0 --no-cpu",
       "units": "KB per dispatch as reported; gfx950 correction (MI355X_MICROARCH.md HBM section): FETCH_SIZE counts "
                "1/2 of the bytes of wide coalesced streaming reads -> read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE exact",
       "kernels": {}}
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = max(glob.glob(os.path.join(src, d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c or "sgm::k_csr" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out["kernels"].setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
for k, v in out["kernels"].items():
    for c in list(v):
        vals = v[c]
        v[c] = {"dispatches": len(vals), "mean_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals)}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_read_bytes_corrected"] = 2 * v["FETCH_SIZE"]["mean_KB"] * 1024
        v["hbm_write_bytes"] = v["WRITE_SIZE"]["mean_KB"] * 1024
        v["hbm_traffic_bytes"] = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
main = [k for k in out["kernels"] if "k_csr_sl" in k] or [k for k in out["kernels"] if "k_csr_do" in k]
if main:
    out["hbm_traffic_bytes"] = out["kernels"][main[0]]["hbm_traffic_bytes"]
    out["dominant_kernel"] = main[0]
out["algorithmic_bytes"] = 799707748
json.dump(out, open(os.path.join(dst, "pmc_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
for r in list(csv.DictReader(open(ks)))[:8]:
    print(r["Name"][:80].ljust(82), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
