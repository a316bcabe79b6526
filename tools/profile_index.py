#!/usr/bin/env python3
"""Builder's tool: the per-round index of profiles/ — for each BASELINE tile, the k_rollout durations of the rocprofv3 kernel trace of `bench.py`
(gpurun_out/prof_<tag>/<case>/kt) next to the HIP-event figures the SAME bench run printed (bench_kt.json).  usage: profile_index.py <tag>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
print(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --case <case> --envs <n> --no-cpu-baseline --no-extras --steps 10 --warmup 2` (tools/collect_profiles.sh),")
print("# the default handle: the 10 timed launches are queued back to back and alternate between the handle's two state lanes, so they OVERLAP on the device;")
print("# the first warm-up launches run alone.  `trace` = End - Start of each k_rollout dispatch in the kernel trace; `bench` = what that same run's JSON line says (HIP events).")
print("# PMC passes (<tag>_<case>_pmc_k_rollout.csv, pmc_traffic.json): the same command with --lanes 1 (per-dispatch counters of overlapping launches are not separable).")
for src in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "*"))):
    case = os.path.basename(src)
    f = sorted(glob.glob(os.path.join(src, "kt", "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "k_rollout" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    s = [int(r["Start_Timestamp"]) for r in rows]; e = [int(r["End_Timestamp"]) for r in rows]
    b = json.loads(open(os.path.join(src, "bench_kt.json")).read().strip().splitlines()[-1])
    k = b["steps"]
    print(f"\n{case}: {len(d)} k_rollout launches in the trace; durations ms: {[round(x, 2) for x in d]}")
    print(f"  timed launches (last {k}): trace mean {sum(d[-k:]) / k:.3f} ms | bench roofline.kernel_ms {b['roofline']['kernel_ms']:.3f} ms (HIP events)")
    print(f"  one completes every: trace {(max(e[-k:]) - min(s[-k:])) / 1e6 / k:.3f} ms | bench ms_per_step {b['ms_per_step']:.3f} ms (under the profiler)")
    print(f"  alone on the GPU (warm-up): trace {min(d[:len(d) - k]):.3f} ms | bench roofline.isolated.kernel_ms {b['roofline']['isolated']['kernel_ms']:.3f} ms")
    print(f"  all-launch average of {tag}_{case}_kernel_stats.csv = {sum(d) / len(d):.3f} ms (warm-up launches included)")
    nb = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    print(f"  without the profiler (bench.json, 40 launches): value {nb['value'] / 1e6:.1f} M env-steps/s, ms_per_step {nb['ms_per_step']:.3f}, kernel_ms {nb['roofline']['kernel_ms']:.3f}, isolated {nb['roofline']['isolated']['kernel_ms']:.3f}, "
          f"frac {nb['roofline']['frac']:.3f}, isolated frac {nb['roofline']['isolated']['frac']:.3f}, device_frac {nb['roofline'].get('device_frac', float('nan')):.3f}, "
          f"hbm_measured_frac {nb['roofline'].get('hbm_measured_frac')}, issue_frac {nb['roofline'].get('issue_frac')}, lane_util {nb['roofline'].get('lane_util')}")
