#!/usr/bin/env python3
"""Condense gpurun_out/prof_round (tools/profile_round.sh) into profiles/<tag>/."""
import csv, glob, hashlib, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_round"), os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))


def newest(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return max(f, key=os.path.getmtime) if f else None


for sub, name in (("stats", "kernel_stats.csv"), ("stats_ildu", "kernel_stats_ildu_pcg_1000x1000.csv"), ("stats_ildu3", "kernel_stats_ildu_100cubed.csv"),
                  ("stats_c3_lowsync", "kernel_stats_c3_gmres_lowsync.csv"), ("stats_c3_mgs", "kernel_stats_c3_gmres_mgs.csv"),
                  ("stats_configs", "kernel_stats_configs_c4_c5.csv"), ("stats_ildu_colour", "kernel_stats_ildu_pcg_colour_3162x3162.csv")):
    f = newest(os.path.join(sub, "*", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, name))
if os.path.exists(os.path.join(src, "time_to_solution.log")):
    shutil.copy(os.path.join(src, "time_to_solution.log"), os.path.join(dst, "time_to_solution_cg_jacobi_ildu.txt"))
for name, to in (("pmc_colour_ildu.txt", "pmc_colour_ildu.txt"), ("cg_small_coop.jsonl", "cg_per_iteration_cooperative.jsonl"),
                 ("cg_small_launch_loop.jsonl", "cg_per_iteration_launch_loop.jsonl"), ("bicgstab_small.jsonl", "bicgstab_per_iteration.jsonl"),
                 ("coop_probe.jsonl", "cg_coop_phase_timers.jsonl"), ("wave_sum_probe.txt", "wave_sum_probe.txt"),
                 ("scattered_csr.json", "scattered_csr.json"), ("ildu_colour_parts_kernel_sums.txt", "ildu_colour_parts_kernel_sums.txt"),
                 ("stream_ceilings.txt", "stream_ceilings.txt"), ("c5_counters.txt", "c5_product_counters.txt"),
                 ("fuzz_campaigns.txt", "fuzz_campaigns.txt"), ("bench_wall.txt", "bench_wall.txt")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, to))
if os.path.exists(os.path.join(src, "configs.jsonl")):
    shutil.copy(os.path.join(src, "configs.jsonl"), os.path.join(dst, "configs_c3_c4_c5.jsonl"))

h = hashlib.sha1()
for f in ("sgm_spmv.hip", "sgm_spmv_select.hpp", "sgm_internal.hpp"):
    h.update(open(os.path.join(root, "sigma_amd", "csrc", f), "rb").read())
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 --spmv-per-step 4 "
                  "--cg-steps 0 --no-cpu --no-c5 --no-variants",
       "units": "KB per dispatch as reported; gfx950 correction (MI355X_MICROARCH.md HBM section): FETCH_SIZE counts "
                "1/2 of the bytes of wide coalesced streaming reads -> read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE exact",
       "csrc_sha1": h.hexdigest(), "kernels": {}}
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = newest(os.path.join(d, "*", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c or "sgm::k_csr" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out["kernels"].setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
# the CG update kernels and the C5 product: their own passes (bench.py with a CG leg and the C5 leg)
for d, c in (("pmc_fetch_cg_c5", "FETCH_SIZE"), ("pmc_write_cg_c5", "WRITE_SIZE")):
    f = newest(os.path.join(d, "*", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if r["Counter_Name"] != c or not ("k_elem<sgm::FCg" in nm or "k_csr_sl<7" in nm):
            continue
        k = nm.split("(")[0].replace("void ", "")
        out["kernels"].setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
for k, v in out["kernels"].items():
    for c in list(v):
        vals = v[c]
        v[c] = {"dispatches": len(vals), "mean_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals)}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_read_bytes_corrected"] = 2 * v["FETCH_SIZE"]["mean_KB"] * 1024
        v["hbm_write_bytes"] = v["WRITE_SIZE"]["mean_KB"] * 1024
        v["hbm_traffic_bytes"] = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
main = [k for k in out["kernels"] if "k_csr_sl<" in k and "hbm_traffic_bytes" in out["kernels"][k]]
bench = json.load(open(os.path.join(dst, "bench.json")))
if main:
    out["hbm_traffic_bytes"] = out["kernels"][main[0]]["hbm_traffic_bytes"]
    out["dominant_kernel"] = main[0]
    moved = bench["roofline"]["moved_bytes_per_launch"]
    out["moved_bytes_by_construction"] = moved
    out["traffic_over_moved"] = out["hbm_traffic_bytes"] / moved
    ks = os.path.join(dst, "kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            if "k_csr_sl<5, false, false, false>" in r["Name"] or "k_csr_sl<5,false,false,false>" in r["Name"].replace(" ", ""):
                t = float(r["AverageNs"]) * 1e-9
                out["rocprof_avg_launch_s"] = t
                out["frac_of_8TBs_on_moved_bytes"] = moved / t / 8e12
                out["frac_of_8TBs_on_pmc_traffic"] = out["hbm_traffic_bytes"] / t / 8e12
                break
out["algorithmic_bytes_reference_layout"] = bench["roofline"]["algorithmic_bytes_per_launch_reference_layout"]
json.dump(out, open(os.path.join(dst, "pmc_hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
ks = os.path.join(dst, "kernel_stats.csv")
if os.path.exists(ks):
    for r in list(csv.DictReader(open(ks)))[:8]:
        print(r["Name"][:80].ljust(82), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
