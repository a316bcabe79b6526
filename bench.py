#!/usr/bin/env python3
"""bench.py — env-steps/sec of the fused batched rollout (BASELINE.json metric) on N MI355X.

One bench "step" = one pass of the hot path over one batch: reset(seeds) + 150 x (Local-Hungarian
allocate -> env.step incl. observation write) for `--envs` independent env instances per GPU, in ONE
kernel launch (muavta_rollout).  Workload = BASELINE config 2b: WPS_hard knobs on 16 UAVs
(`WPS_hard_x2`), 4096 envs per GPU, 16x32 tile, seeds = global env index.

    python bench.py [--gpus N --steps K --warmup W]          # N>1: launched by torch.distributed.run

Prints ONE JSON line on rank 0.  `value` is whole-job env-steps/s (all ranks), inputs (seeds) already
resident on the device, timed region bracketed by barrier + torch.cuda.synchronize() on both sides,
max over ranks.  `roofline.achieved` = SURVEY §8(d) algorithmic bytes per env-step x env-steps per
launch / mean kernel duration, the latter measured with HIP events on the library's own stream.
`cpu_baseline` = the CPU oracle (oracle/, a restatement of the reference: kind "port") timed on this
box's host cores over a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HORIZON = 150
# algorithmic bytes per env-step B(A,T,H) = 2*S_state + S_obs + S_act
ALGO_BYTES_PER_ENV_STEP = {"16x32": 39.9e3, "24x48": 65.7e3, "64x128": 255e3}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(case: str, interval: int, seconds: float):
    """Oracle (CPU restatement, test infrastructure) on the host cores: bounded sample, multi-process."""
    import multiprocessing as mp

    cores = max(1, min(os.cpu_count() or 1, 16))
    with mp.get_context("spawn").Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.starmap(_cpu_worker, [(case, interval, seconds, 10_000_000 + 4096 * r) for r in range(cores)])
        wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {
        "value": steps / busy, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{steps // HORIZON} episodes of {case} (150 steps each, Local-Hungarian interval {interval}) "
                  f"split over {cores} processes, {busy:.1f} s of work each ({wall:.1f} s wall incl. spawn)",
    }


def _cpu_worker(case, interval, seconds, seed0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from muavta_amd.params import params_for_case

    e = orc.OracleEnv(params_for_case(case))
    e.rollout(seed0, HORIZON, interval, 1)  # warm
    t0 = time.perf_counter()
    steps, s = 0, seed0 + 1
    while time.perf_counter() - t0 < seconds:
        steps += e.rollout(s, HORIZON, interval, 1)
        s += 1
    return steps, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="env instances per GPU")
    ap.add_argument("--case", default="WPS_hard_x2")
    ap.add_argument("--interval", type=int, default=20)
    ap.add_argument("--no-obs", action="store_true", help="skip the per-step observation write (NOT the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed-base", type=int, default=0, help="first global env index (default 0: seeds = global env index)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run the RCCL path is exercised even at N=1
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))  # RCCL over xGMI

    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.params import params_for_case

    params = params_for_case(args.case)
    env = BatchedMultiUAVEnv(params, args.envs, device=local_rank)
    tile = f"{env.dims.tile_agents}x{env.dims.tile_tasks}"
    from muavta_amd.dist import shard_seeds

    seeds = shard_seeds(rank, args.envs, args.seed_base)  # seed = global env index
    write_obs = not args.no_obs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        env.rollout(seeds, HORIZON, args.interval, True, write_obs)
        env.sync()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        env.rollout(seeds, HORIZON, args.interval, True, write_obs)
        kernel_ms.append(env.last_kernel_ms())  # HIP events on the library's stream; waits for the launch
    barrier()
    elapsed = time.perf_counter() - t0

    # secondary figures: the per-step paths, state blob loaded from / stored to HBM by every launch; untimed
    # w.r.t. the headline.  step_api: one k_allocate + one k_step launch per env step (a caller that looks at the plan);
    # fused_step_api: muavta_rollout(h, NULL, 1, ...) = allocate + step + observe in ONE launch per env step.
    step_api = fused_step_api = None
    if rank == 0 and world == 1:
        env.reset(seeds)
        env.sync()
        t1 = time.perf_counter()
        for _ in range(HORIZON):
            env.allocate(args.interval, True, fetch=False)
            env.step_staged()
        env.sync()
        step_api = args.envs * HORIZON / (time.perf_counter() - t1)
        env.reset(seeds)
        env.sync()
        t1 = time.perf_counter()
        for _ in range(HORIZON):
            env.rollout(None, 1, args.interval, True, write_obs)
        env.sync()
        fused_step_api = args.envs * HORIZON / (time.perf_counter() - t1)
        env.rollout(seeds, HORIZON, args.interval, True, write_obs)  # restore the headline batch's final state
        env.sync()

    # metrics of the last batch: per-rank partials -> the one collective of this path (muavta_amd/dist.py)
    from muavta_amd.dist import reduce_metrics

    m = env.rollout_metrics()
    # An env that needs more than its tile holds (task slots / queue depth) says so instead of returning a wrong episode
    # (about 1 in 10,000 seeds on the 16x32 tile, e.g. global index 9649).  It ran its 150 steps like the
    # others, so the throughput stands; its metrics are left out of the quality summary and it is counted below.
    flagged = env.get("ERROR") != 0
    n_flagged = int(np.count_nonzero(flagged))
    if n_flagged > max(1, args.envs // 1000):
        raise SystemExit(f"{n_flagged} of {args.envs} envs overflowed the tile: pick a larger one (tile_tasks / tile_agents)")
    summary = reduce_metrics(m[~flagged], device="cuda" if dist is not None else None)
    tmax = torch.tensor([elapsed, float(np.mean(kernel_ms)), float(n_flagged)], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax[:2], op=dist.ReduceOp.MAX)
        dist.all_reduce(tmax[2:], op=dist.ReduceOp.SUM)
    elapsed, mean_kernel_ms, n_flagged = float(tmax[0].item()), float(tmax[1].item()), int(tmax[2].item())

    if rank == 0:
        total_envs = args.envs * world
        env_steps = total_envs * HORIZON * args.steps
        value = env_steps / elapsed
        B = ALGO_BYTES_PER_ENV_STEP.get(tile)
        achieved = (args.envs * HORIZON * B) / (mean_kernel_ms * 1e-3) / 1e9 if B else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from the rocprofv3 --pmc passes
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.case}:{args.envs}")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec at N parallel envs, WPS_hard 16x32; 1/2/4/8 GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (reference's generative process, MT19937 seeded by global env index)",
            "config": {"workload": f"{args.case}: reset + 150 fused steps, Local-Hungarian interval {args.interval}, visibility on, "
                                   f"obs write {'on' if write_obs else 'off'}",
                       "envs_per_gpu": args.envs, "total_envs": total_envs, "tile": tile, "n_agents": env.n_agents,
                       "horizon": HORIZON, "parallelism": f"env-sharded x{world}, RCCL all-reduce of the metric vector only"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "k_rollout", "kernel_ms": mean_kernel_ms, "algorithmic_bytes_per_env_step": B},
            "step_api_env_steps_per_s": step_api, "fused_step_api_env_steps_per_s": fused_step_api,
            "quality": {"mean_S_WPS": summary["mean_S_WPS"], "std_S_WPS": summary["std_S_WPS"], "on_time_rate": summary["on_time_rate"],
                        "n_envs": summary["n_envs"], "capacity_flagged_envs": n_flagged},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.case, args.interval, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
