#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs merged back under gpurun_out/prof_<tag>/<case>/ into the small summaries committed under
profiles/ (kernel stats, PMC per-launch averages, derived per-env-step figures) and update profiles/pmc_traffic.json,
which bench.py reads for `roofline.traffic` (keyed by case:envs, valid only for the kernel sources it was taken on).
HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE is NOT doubled here: the guide's x2 correction
applies to wide (16 B/lane) coalesced streaming reads, whereas most of this kernel's reads are 8-byte rows of the
per-env HBM record and single-lane tape words — stated as uncalibrated in DESIGN.md."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
from muavta_amd.native import source_hash  # (not from bench: importing bench sets process-wide HIP runtime knobs)
for src in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "*"))):
    case = os.path.basename(src)
    ks = sorted(glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)  # (gpurun merges: newest collection first)
    if not ks:
        continue
    rows = list(csv.DictReader(open(ks[0])))
    with open(os.path.join(dst, f"{tag}_{case}_kernel_stats.csv"), "w") as g:
        g.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            g.write(",".join([name.strip('"'), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]]) + "\n")
    pmc, meta = {}, {}
    lanes = None
    for sub in ("fetch", "write", "sq1", "sq2", "sq3", "sq4"):
        fs = sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
        if not fs:
            continue
        own = {}
        for r in csv.DictReader(open(fs[0])):
            if "k_rollout" not in r["Kernel_Name"]:
                continue
            own.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if not meta:  # launch geometry / register figures of a k_rollout row (NOT of whatever kernel the CSV happens to end with)
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}
            if sub == "sq4" and r["Counter_Name"] in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"):
                continue  # (already collected in sq1 / sq2; here only as the same-pass denominators of the lane figure below)
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if sub == "sq4" and "SQ_THREAD_CYCLES_VALU" in own and "SQ_ACTIVE_INST_VALU" in own:
            # rocprofv3's VALUUtilization: thread-cycles / (instruction cycles x 64) — mean share of the 64 lanes a VALU instruction has enabled
            lanes = 64.0 * sum(own["SQ_THREAD_CYCLES_VALU"]) / (sum(own["SQ_ACTIVE_INST_VALU"]) * 64.0)
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(dst, f"{tag}_{case}_bench.json"), "w"), indent=1)
    envs = bench["config"]["envs_per_gpu"]
    steps = envs * 150
    mean = {k: sum(v) / len(v) for k, v in pmc.items()}
    with open(os.path.join(dst, f"{tag}_{case}_pmc_k_rollout.csv"), "w") as g:
        g.write(f"# k_rollout, {case}, {envs} envs x 150 steps per launch; per-launch means; per_env_step = mean / {steps}\n")
        g.write("# " + ", ".join(f"{k}={v}" for k, v in meta.items()) + "\n")
        g.write("Counter,MeanPerLaunch,Launches,PerEnvStep\n")
        for k, v in pmc.items():
            g.write(f"{k},{mean[k]:.6g},{len(v)},{mean[k] / steps:.4g}\n")
        if "SQ_WAVE_CYCLES" in mean:
            wc = mean["SQ_WAVE_CYCLES"]
            g.write("# derived: " + ", ".join(f"{k}/SQ_WAVE_CYCLES={mean[k] / wc:.3f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in mean) + "\n")
            insts = sum(mean.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"))
            g.write(f"# derived: instructions per env-step (VALU+SALU+LDS+SMEM+VMEM+branch) = {insts / steps:.0f}, wave cycles per env-step = {wc / steps:.0f}\n")
        if lanes is not None:
            g.write(f"# derived: mean enabled lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, same pass) = {lanes:.2f} of 64\n")
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        traffic = (mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        tp = os.path.join(dst, "pmc_traffic.json")
        d = json.load(open(tp)) if os.path.exists(tp) else {}
        d = {k: v for k, v in d.items() if isinstance(v, dict) or k == "_note"}
        entry = {"bytes_per_launch": traffic, "source_hash": source_hash(), "profile": f"{tag}_{case}_pmc_k_rollout.csv"}
        if "SQ_INSTS_VALU" in mean and "SQ_WAVE_CYCLES" in mean and "SQ_ACTIVE_INST_VALU" in mean:
            # issue side (bench.py: roofline.issue).  Waves per SIMD: one single-wave workgroup per env over 256 CUs x 4 SIMDs,
            # all resident at once for the BASELINE sizes; the VALU port of a SIMD serves its waves one at a time, so its occupancy
            # over the launch is waves_per_simd x (a wave's VALU-active cycles / its resident cycles), both in quad-cycles
            wps = max(1.0, min(envs / 1024.0, 8.0))
            entry["issue"] = {
                "valu_per_env_step": mean["SQ_INSTS_VALU"] / steps, "salu_per_env_step": mean.get("SQ_INSTS_SALU", 0) / steps,
                "lds_per_env_step": mean.get("SQ_INSTS_LDS", 0) / steps, "branch_per_env_step": mean.get("SQ_INSTS_BRANCH", 0) / steps,
                "wave_quad_cycles_per_env_step": mean["SQ_WAVE_CYCLES"] / steps, "waves_per_simd": wps,
                "valu_port_busy": wps * mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"],
                "wait_any_frac": mean.get("SQ_WAIT_ANY", 0) / mean["SQ_WAVE_CYCLES"],
                "mean_enabled_lanes_per_valu": lanes,
            }
        d[f"{case}:{envs}"] = entry
        d["_note"] = "HBM bytes per k_rollout launch = (FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes (tools/collect_profiles.sh); valid for the kernel sources with this source_hash only"
        json.dump(d, open(tp, "w"), indent=1)
        print(case, "traffic bytes/launch", traffic)
    print(open(os.path.join(dst, f"{tag}_{case}_kernel_stats.csv")).read())
    print(open(os.path.join(dst, f"{tag}_{case}_pmc_k_rollout.csv")).read())
