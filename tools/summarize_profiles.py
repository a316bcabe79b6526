#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs merged back under gpurun_out/prof_<tag>/ into the small summaries committed under
profiles/ (kernel stats, PMC per-launch averages) and update profiles/pmc_traffic.json, which bench.py reads
for `roofline.traffic`.  HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE is NOT doubled
here: the guide's x2 correction applies to wide (16 B/lane) coalesced streaming reads, whereas this kernel's
reads are single-lane 4-byte tape words and one 1-KiB blob prologue — stated as uncalibrated in DESIGN.md."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as g:
    g.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    for r in rows:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        g.write(",".join([name.strip('"'), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]]) + "\n")
pmc = {}
meta = {}
for sub in ("fetch", "write", "sq1", "sq2"):
    fs = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        if "k_rollout" not in r["Kernel_Name"]:
            continue
        pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
with open(os.path.join(dst, f"{tag}_pmc_k_rollout.csv"), "w") as g:
    g.write("# k_rollout<Tile16>, bench.py default workload (WPS_hard_x2, 4096 envs x 150 steps per launch); per-launch means\n")
    g.write("# " + ", ".join(f"{k}={v}" for k, v in meta.items()) + "\n")
    g.write("Counter,MeanPerLaunch,Launches\n")
    for k, v in pmc.items():
        g.write(f"{k},{sum(v) / len(v):.6g},{len(v)}\n")
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    traffic = (sum(pmc["FETCH_SIZE"]) / len(pmc["FETCH_SIZE"]) + sum(pmc["WRITE_SIZE"]) / len(pmc["WRITE_SIZE"])) * 1024
    tp = os.path.join(dst, "pmc_traffic.json")
    d = json.load(open(tp)) if os.path.exists(tp) else {}
    d[f"{bench['config']['workload'].split(':')[0]}:{bench['config']['envs_per_gpu']}"] = traffic
    d["_note"] = "HBM bytes per k_rollout launch = (FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes (tools/collect_profiles.sh)"
    json.dump(d, open(tp, "w"), indent=1)
    print("traffic bytes/launch", traffic)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
print(open(os.path.join(dst, f"{tag}_pmc_k_rollout.csv")).read())
