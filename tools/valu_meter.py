#!/usr/bin/env python3
"""VALU-issue meter.  k_rollout at 4 waves per SIMD keeps the SIMD's VALU port ~75-85 % busy (profiles/r02_*_pmc_k_rollout.csv:
4 x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), so the figure to push down is VALU issue cycles per env-step.  This tool measures
it, noise-free, per kernel of a fixed workload:
    drive  : the workload (run under `rocprofv3 --kernel-trace --pmc ...`): 3 fused rollouts with the per-step observation,
             3 without, one episode through the per-step kernels (k_allocate + k_step), 150 x k_observe
    report : per-group means from the counter CSVs of the two passes tools/valu_meter.sh makes
Usage on the GPU box: bash tools/valu_meter.sh <tag> [case] [envs]   (writes gpurun_out/valu/<tag>.txt)"""
import csv, glob, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def drive(case, n):
    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.params import params_for_case
    interval = 12 if "escort" in case else 20
    env = BatchedMultiUAVEnv(params_for_case(case), n, device=0)
    seeds = np.arange(n, dtype=np.uint64)
    for obs in (True, True, True, False, False, False):
        env.rollout(seeds, 150, interval, True, obs)
    env.sync()
    env.reset(seeds)
    for _ in range(150):
        env.allocate(interval, True, fetch=False)
        env.step_staged()
    env.sync()
    for _ in range(150):
        env.refresh_observation()
    env.sync()


def report(tag, case, n):
    groups = {}  # (group, counter) -> [values]
    for sub in ("a", "b", "c", "d"):
        fs = glob.glob(os.path.join(ROOT, "gpurun_out", "valu", f"{tag}_{sub}", "*", "*_counter_collection.csv"))
        if not fs:
            continue
        rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Dispatch_Id"]))
        seen_rollouts = {}
        for r in rows:
            k = r["Kernel_Name"]
            did = r["Dispatch_Id"]
            if "k_rollout" in k:
                order = seen_rollouts.setdefault(did, len(seen_rollouts))
                g = "k_rollout obs on (150 steps)" if order < 3 else "k_rollout obs off (150 steps)"
            elif "k_allocate" in k:
                g = "k_allocate (1 step)"
            elif "k_step" in k:
                g = "k_step incl. obs (1 step)"
            elif "k_observe" in k:
                g = "k_observe (1 step)"
            else:
                continue
            groups.setdefault((g, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    names = sorted({g for g, _ in groups})
    ctrs = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
    if any(c in ("FETCH_SIZE", "WRITE_SIZE") for _, c in groups):  # traffic passes (tools/valu_meter.sh <tag> <case> <n> traffic)
        from muavta_amd.batched import BatchedMultiUAVEnv
        from muavta_amd.params import params_for_case
        p = params_for_case(case)
        A, MT = p.n_agents, p.max_tasks
        obs_bytes = 21 * MT * 4 + A * ((MT + 63) // 64) * 8 + MT + A * 9 * 4 + 5 * 4 + 8 + 1
        print(f"{case}, {n} envs; KB per env-step as the counters report them (x 1024 B); known byte counts: one observation = {obs_bytes} B per env, "
              "the LDS image (read and written once per k_step / k_allocate / k_observe launch, 16 B per lane) = see lds_bytes_per_env")
        for g in names:
            steps = n * (150 if "150" in g else 1)
            f = np.mean(groups[(g, "FETCH_SIZE")]) * 1024 / steps if (g, "FETCH_SIZE") in groups else float("nan")
            w = np.mean(groups[(g, "WRITE_SIZE")]) * 1024 / steps if (g, "WRITE_SIZE") in groups else float("nan")
            print(f"{g:34s} FETCH_SIZE {f:10.1f} B   WRITE_SIZE {w:10.1f} B   per env-step")
        return
    print(f"{case}, {n} envs; per env-step means (SQ_ACTIVE_* and SQ_WAVE_CYCLES count 4-cycle quads)")
    print(f"{'group':34s}" + "".join(f"{c.replace('SQ_', ''):>17s}" for c in ctrs))
    for g in names:
        steps = n * (150 if "150" in g else 1)
        print(f"{g:34s}" + "".join(f"{np.mean(groups[(g, c)]) / steps:17.1f}" if (g, c) in groups else f"{'-':>17s}" for c in ctrs))


if __name__ == "__main__":
    mode = sys.argv[1]
    case = sys.argv[3] if len(sys.argv) > 3 else "WPS_hard_x2"
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    if mode == "drive":
        drive(case, n)
    else:
        report(sys.argv[2], case, n)
