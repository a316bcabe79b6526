#!/usr/bin/env python3
"""bench.py — env-steps/sec of the fused batched rollout (BASELINE.json metric) on N MI355X.

One bench "step" = one pass of the hot path over one batch: reset(seeds) + 150 x (Local-Hungarian
allocate -> env.step incl. observation write) for `--envs` independent env instances per GPU, in ONE
kernel launch (muavta_rollout).  Workload = BASELINE config 2b: WPS_hard knobs on 16 UAVs
(`WPS_hard_x2`), 4096 envs per GPU, the 16-agent tile, seeds = global env index.

    python bench.py [--gpus N --steps K --warmup W]          # N>1: launched by torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
        bench.py --gpus 8 --case WPS_burst64 --envs 1024       # BASELINE config 5 (8192 envs, 64x128 tile)

Prints ONE JSON line on rank 0.  `value` is whole-job env-steps/s (all ranks), inputs (seeds) already
resident on the device, timed region bracketed by barrier + torch.cuda.synchronize() on both sides,
max over ranks.  `roofline.achieved` = SURVEY §8(d) algorithmic bytes per env-step x env-steps per
launch / mean k_rollout duration, the latter measured with HIP events on the library's own stream; the
kernel keeps the env state in LDS, so this is an algorithmic-equivalent rate, NOT measured HBM traffic —
`roofline.traffic` / `roofline.measured_hbm_GBs` are the PMC-measured bytes (see `roofline.note`).
`cpu_baseline` = the CPU oracle (oracle/, a restatement of the reference: kind "port") timed on ALL of this
box's host cores over a bounded sample of the same workload (rank 0, N=1 only).
`other_tiles` (N=1 only) = the same measurement for BASELINE configs 4 and 5 (24x48 escort, 64x128 burst).  The timed loop queues its
launches back to back on ONE handle; since round 5 the handle runs them on two state lanes (muavta_set_lanes: launch i+1 starts in the wave
slots launch i's early finishers free), `--lanes 1` pins it to one lane and `value_one_lane` / `other_tiles.*.one_lane` report that mode.
Further N=1 figures: `launch_gap_ms` (where ms_per_step goes beyond the kernel), `policy_in_loop_env_steps_per_s` + `policy_calls_per_env_step`
(muavta_rl_run_device: the RL trainers' loop, the policy consulted once per replan gate), `facade_steps_per_s` (the drop-in PettingZoo facade).
Progress lines go to stderr.
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
# HIP maps streams onto a small pool of hardware queues (4 by default); kernels of two streams that share one execute in order.  The
# sub-batch figure (fused_step_api, muavta_set_parts) wants the part streams on queues of their own next to torch's and the handle's
# main / seeding streams.  A runtime knob of the HIP runtime, read when it initialises; a value the caller exported wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MUAVTA_EAGER_PART_STREAMS", "8")  # (muavta_create opens the sub-batch streams it will be asked for right behind its main streams: INTEGRATION.md)

HORIZON = 150
# algorithmic bytes per env-step B(A,T,H) = 2*S_state + S_obs + S_act, by agent count of the tile
ALGO_BYTES_PER_ENV_STEP = {16: 39.9e3, 24: 65.7e3, 64: 255e3}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# BASELINE.json configs 4 and 5 (per-GPU size): (case, envs, replan interval)
OTHER_TILES = (("WPS_escort24", 4096, 12), ("WPS_burst64", 1024, 20))


def source_hash() -> str:
    """Hash of the kernel sources: PMC traffic figures under profiles/ are only reused for the build they were taken on."""
    from muavta_amd import native

    return native.source_hash()


def pmc_entry(case: str, envs: int):
    """The committed rocprofv3 --pmc summary of k_rollout for this (case, envs) (tools/collect_profiles.sh ->
    tools/summarize_profiles.py -> profiles/pmc_traffic.json), or None when it was taken on other kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    e = d.get(f"{case}:{envs}")
    if isinstance(e, dict) and e.get("source_hash") == source_hash():
        return e
    return None


def measured_traffic(case: str, envs: int):
    """HBM bytes per k_rollout launch from the PMC passes, or None (see pmc_entry)."""
    e = pmc_entry(case, envs)
    return e.get("bytes_per_launch") if e else None


def host_cpu():
    """(CPU model, cores this process may actually use, logical CPUs visible).  The usable count is the smaller of the
    affinity mask and the cgroup CPU quota: a GPU box leases a share of its host (16 CPUs per GPU on this pool) while
    still showing every logical CPU, and oversubscribing the share only slows the baseline down."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    cores = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    return model, max(1, cores), visible


def oracle_check(case: str, interval: int, seeds, gpu_metrics, n_check: int = 256):
    """The timed GPU batch against the CPU oracle on a subsample (part of the cpu_baseline leg: the only place bench.py may touch
    oracle/): all 30 metrics of `n_check` envs spread over the batch must be bit-equal.  Returns the `quality` additions."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    idx = np.unique(np.linspace(0, len(seeds) - 1, min(n_check, len(seeds))).astype(np.int64))
    want = orc.parallel_metrics(case, [int(seeds[i]) for i in idx], interval)
    got = np.asarray(gpu_metrics)[idx]
    bad = np.nonzero(~np.all(got == want, axis=1))[0]
    if len(bad):
        raise SystemExit(f"bench: GPU metrics differ from the CPU oracle for seeds {[int(seeds[idx[b]]) for b in bad[:8]]} of {case}")
    return {"oracle_checked": True, "oracle_checked_envs": int(len(idx)),
            "oracle_mean_S_WPS_of_checked": float(want[:, 4].mean()), "gpu_mean_S_WPS_of_checked": float(got[:, 4].mean())}


def cpu_baseline(case: str, interval: int, seconds: float):
    """Oracle (CPU restatement, test infrastructure) on every host core this process may use: bounded sample."""
    import multiprocessing as mp

    model, cores, visible = host_cpu()
    if os.environ.get("MUAVTA_CPU_BASELINE_PROCS"):
        cores = int(os.environ["MUAVTA_CPU_BASELINE_PROCS"])
    with mp.get_context("spawn").Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.starmap(_cpu_worker, [(case, interval, seconds, 10_000_000 + 4096 * r) for r in range(cores)])
        wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {
        "value": steps / busy, "unit": "env-steps/s", "cores": cores, "cpu_model": model, "logical_cpus_visible": visible, "kind": "port",
        "sample": f"{steps // HORIZON} episodes of {case} (150 steps each, Local-Hungarian interval {interval}) "
                  f"split over {cores} processes (one per host core this box may use: min of affinity mask and cgroup CPU quota), {busy:.1f} s of work each ({wall:.1f} s wall incl. spawn)",
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


def roofline(case, envs, n_agents_tile, kernel_ms, isolated_ms=None, ms_per_step=None, lanes=1, max_tasks=None, n_agents=None):
    """SURVEY 8(d) roofline block of k_rollout.  `achieved` / `frac` stay exactly as 8(d) defines them (algorithmic bytes per launch / that
    kernel's mean launch duration, against HBM peak) — but the kernel is not HBM-bound, and the labels say so: `bound` = "issue", with
    `issue_frac` (VALU-port occupancy), `lane_util` (enabled lanes per VALU instruction / 64) and the measured HBM traffic beside it, and
    `hbm_floor` = what a state-resident rollout MUST move per env-step (the observation it writes and the actions it reads).  The PMC-derived
    fields are null unless profiles/pmc_traffic.json holds a collection taken on exactly these kernel sources (source_hash)."""
    B = ALGO_BYTES_PER_ENV_STEP.get(n_agents_tile)
    launch_bytes = envs * HORIZON * B if B else None
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9 if B else None
    e = pmc_entry(case, envs)
    traffic = e.get("bytes_per_launch") if e else None
    measured = (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None
    issue = None
    if e and e.get("issue"):
        issue = dict(e["issue"])
        issue["source"] = "profiles/" + str(e.get("profile"))
        issue["source_hash"] = e.get("source_hash")
    out = {
        "bound": "issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
        "kernel": "k_rollout", "kernel_ms": kernel_ms, "algorithmic_bytes_per_env_step": B,
        "issue_frac": issue.get("valu_port_busy") if issue else None,
        "lane_util": (issue.get("mean_enabled_lanes_per_valu") / 64.0) if issue and issue.get("mean_enabled_lanes_per_valu") else None,
        "achieved_is": ("SURVEY 8d as written: algorithmic bytes (2*S_state + S_obs + S_act per env-step) / the kernel's mean launch duration — an algorithmic-equivalent rate, "
                        "not HBM traffic, kept for comparability between rounds.  It is NOT what bounds the kernel (`bound`: instruction issue): the fused kernel keeps the env state "
                        "in LDS for the whole rollout and never moves the 2*S_state per step the formula charges, so values above 1.0 are possible; that the timed work is the whole "
                        "workload is shown by `quality` equalling the CPU oracle (quality.oracle_checked)"),
        "measured_hbm_GBs": measured, "hbm_measured_frac": (measured / HBM_PEAK_GBS) if measured else None,
        "issue": issue,
        "limiter": ("instruction issue, not memory: the env state is LDS-resident, the order-dependent phases of a step execute with one "
                    "lane enabled, and the waves sharing a SIMD keep its VALU port busy most of the launch (`issue_frac` = issue.valu_port_busy = "
                    "waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, `lane_util` = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64, from the matching PMC pass)"),
    }
    if max_tasks and n_agents:
        # what a state-resident rollout has to move per env-step: S_obs + S_act of SURVEY 8(d) with the env's own max_tasks / fleet
        # (the observation it writes: T*84 + T + A*(36 + 4*ceil(T/32)) + 30, and the 4A bytes of actions)
        floor = max_tasks * 84 + max_tasks + n_agents * (36 + 4 * ((max_tasks + 31) // 32)) + 30 + 4 * n_agents
        fl_rate = envs * HORIZON * floor / (kernel_ms * 1e-3) / 1e9
        out["hbm_floor_bytes_per_env_step"] = floor
        out["hbm_floor_frac"] = fl_rate / HBM_PEAK_GBS
        out["hbm_floor_is"] = "S_obs + S_act: the bytes per env-step a rollout that keeps its state on chip still has to write / read; hbm_floor_frac = that rate / HBM peak"
    if isolated_ms:
        iso = launch_bytes / (isolated_ms * 1e-3) / 1e9 if B else None
        out["isolated"] = {"kernel_ms": isolated_ms, "achieved": iso, "frac": (iso / HBM_PEAK_GBS) if iso else None,
                           "is": "one launch alone on the GPU (warm-up launches, HIP events): the figure earlier rounds reported as roofline.frac"}
    if ms_per_step and lanes > 1:
        dev = launch_bytes / (ms_per_step * 1e-3) / 1e9 if B else None
        out["launches_in_flight"] = lanes
        out["device_achieved"] = dev
        out["device_frac"] = (dev / HBM_PEAK_GBS) if dev else None
        out["device_is"] = ("with two state lanes the timed launches overlap: each lasts `kernel_ms` but one completes every ms_per_step; device_achieved = algorithmic bytes per launch / ms_per_step")
    return out


def time_rollouts(env, seeds, interval, write_obs, steps, warmup, barrier):
    """(elapsed s of the timed region, mean k_rollout ms of its launches, mean k_seed ms, mean k_rollout ms of ISOLATED launches).  The warm-up
    launches run one at a time (isolated kernel / seeding times) except the last two, which are queued back to back like the timed ones: the
    handle's second state lane (include/muavta.h: a seeded rollout issued while the previous one still runs goes to the other lane) is
    created there, not inside the timed region."""
    kernel_ms, seed_ms, iso_ms = [], [], []
    n_iso = max(1, warmup - 2)
    for w in range(n_iso):
        env.rollout(seeds, HORIZON, interval, True, write_obs)
        env.sync()
        if w or n_iso == 1:  # (the very first call also allocates the seeding buffers)
            seed_ms.append(env.last_seed_ms())  # k_seed on an idle GPU (in the timed loop it runs next to the previous launch)
            iso_ms.append(env.last_kernel_ms())
    for w in range(warmup - n_iso):
        env.rollout(seeds, HORIZON, interval, True, write_obs)
    env.sync()
    barrier()
    t0 = time.perf_counter()
    # launches are queued back to back (the library seeds launch i+1 on its own stream while launch i runs, and with two state lanes launch
    # i+1 itself starts in the wave slots launch i's early finishers free); the per-launch k_rollout durations come from the handle's ring of
    # 64 HIP event pairs.  Reading them back (an event query per launch, ~0.1 ms each) is bookkeeping, not work: it happens after the closing
    # barrier, and inside the loop only when the ring would wrap.
    pending = 0
    for _ in range(steps):
        env.rollout(seeds, HORIZON, interval, True, write_obs)
        pending += 1
        if pending == 64:
            kernel_ms.extend(env.kernel_ms_history(pending).tolist())
            pending = 0
    env.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    gaps = env.launch_gaps_ms(min(pending, 64)).tolist() if pending >= 2 else []
    if pending:
        kernel_ms.extend(env.kernel_ms_history(pending).tolist())
    if not seed_ms:
        seed_ms.append(env.last_seed_ms())
    time_rollouts.last_gap_ms = float(np.mean(gaps)) if gaps else None   # start(i + 1) - end(i), from the same event rings (negative: launches overlapped on two lanes)
    time_rollouts.isolated_kernel_ms = float(np.mean(iso_ms)) if iso_ms else None
    return elapsed, float(np.mean(kernel_ms)), float(np.mean(seed_ms))


def one_lane_figure(case, n, interval, seeds, write_obs, steps, warmup, barrier, device, want_metrics):
    """The same timed loop on a handle pinned to ONE state lane (muavta_set_lanes(h, 1): launches in order, none overlapping — the only mode
    before round 5); its last batch must be complete (no capacity flags) and bit-equal to `want_metrics` (the default handle's batch)."""
    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.params import params_for_case

    h = BatchedMultiUAVEnv(params_for_case(case), n, device=device)
    try:
        h.set_lanes(1)
        el, kms, _ = time_rollouts(h, seeds, interval, write_obs, steps, warmup, barrier)
        if int(np.count_nonzero(h.get("ERROR"))):
            raise SystemExit(f"bench: capacity-flagged envs in the one-lane batch of {case}")
        if want_metrics is not None and not np.array_equal(h.rollout_metrics(), want_metrics):
            raise SystemExit(f"bench: the one-lane batch of {case} differs from the two-lane handle's batch")
        return {"env_steps_per_s": n * HORIZON * steps / el, "ms_per_step": el / steps * 1e3, "kernel_ms": kms, "launch_gap_ms": getattr(time_rollouts, "last_gap_ms", None),
                "bit_equal_to_default_handle_batch": want_metrics is not None}
    finally:
        h.close()


def facade_figures(device, case="WPS_hard", seeds=range(8), batch_n=64):
    from muavta_amd.env import MultiUAVEnv
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS

    ta, tt, th = TILES[case]
    kw = dict(flags=dict(WPS_ENV_FLAGS), device=device, tile_agents=ta, tile_tasks=tt, tile_threats=th)

    def harness_reads(env):
        live = env.get_live_agents()
        open_tasks = [t for t in env.tasks if t.id != 0 and t.status != 2]
        vis = env.agent_visibility_map()
        return live, open_tasks, vis

    def plan_to_actions(env, aa, ai):
        pairs = [(env.agents_obj[int(a)].name, env.last_tasks_info[int(i)]) for a, i in zip(aa[0], ai[0]) if a >= 0]
        actions = {}
        for name, task in pairs:  # _apply_assign (experiments/wps_eval.py:55-61)
            if env.last_tasks_info and task in env.last_tasks_info:
                actions[name] = env.last_tasks_info.index(task)
        return actions

    env = MultiUAVEnv(CASE_SPECS[case], **kw)
    n_steps, t_total = 0, 0.0
    for k, seed in enumerate([0] + list(seeds)):  # (the first episode warms up)
        t0 = time.perf_counter()
        obs, info = env.reset(seed=seed)
        done, n = False, 0
        while not done:
            harness_reads(env)
            actions = plan_to_actions(env, *env._b.allocate(20, True))
            obs, rew, term, trunc, info = env.step(actions)
            n += 1
            done = all(term.values()) or all(trunc.values())
        if k:
            n_steps += n; t_total += time.perf_counter() - t0
    single = n_steps / t_total
    s_wps = float(info["metrics"]["S_WPS"])
    # the vectorised facade: batch_n env objects over ONE handle, the same per-env Python loop
    batch = MultiUAVEnv.batch(CASE_SPECS[case], batch_n, **kw)
    rate_b = None
    for rep in range(2):
        t0 = time.perf_counter()
        outs = batch.reset(list(range(batch_n)))
        n = 0
        while True:
            acts = []
            for v in batch.envs:
                harness_reads(v)
                acts.append(plan_to_actions(v, *v._b.allocate(20, True)))
            outs = batch.step(acts)
            n += 1
            if all(all(o[3].values()) or all(o[2].values()) for o in outs):
                break
        rate_b = batch_n * n / (time.perf_counter() - t0)
    batch.close()
    return {"facade_steps_per_s": single,
            "facade_is": (f"muavta_amd.env.MultiUAVEnv (HIP backend, 1 env per handle), {case} seeds {list(seeds)[0]}..{list(seeds)[-1]}, the loop of experiments/wps_eval.py:112-133,273 "
                          "(live agents / open tasks / visibility map read every step, plan as [(name, Task)] -> _apply_assign -> env.step(dict) -> observation dicts); the plan itself is the "
                          "device allocator's, read back (the reference's Python HungarianAllocator cannot travel to this box: ~4 % of the reference's step time)"),
            "facade_vs_reference_python": single / 1166.0,
            "facade_reference_python_steps_per_s": 1166.0,
            "facade_reference_is": "BASELINE.md: the reference's own loop, WPS_hard, 1 process, measured in the build container (8-vCPU Xeon 2.1 GHz — NOT this box's CPU; the reference cannot travel)",
            "facade_batch_env_steps_per_s": rate_b, "facade_batch_envs": batch_n,
            "facade_batch_is": f"MultiUAVEnv.batch({batch_n}): the same per-env Python loop over {batch_n} env objects that share ONE handle (one launch + one state mirror + one observation copy per step for all)",
            "facade_last_episode_S_WPS": s_wps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)  # (<= 64: the per-launch durations of the whole timed region fit the handle's event ring, nothing is read back inside it)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--envs", type=int, default=4096, help="env instances per GPU")
    ap.add_argument("--case", default="WPS_hard_x2")
    ap.add_argument("--interval", type=int, default=None, help="replan interval (default 20; 12 for the escort cases)")
    ap.add_argument("--no-obs", action="store_true", help="skip the per-step observation write (NOT the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the step-API / other-tile / IL figures (profiling runs)")
    ap.add_argument("--abi-collective", action="store_true", help="reduce the metrics through muavta_allreduce_metrics (RCCL behind the C ABI) instead of torch.distributed")
    ap.add_argument("--seed-base", type=int, default=0, help="first global env index (default 0: seeds = global env index)")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 1, 2), help="state lanes of the handle (muavta_set_lanes): 0 = library default (a second lane is created when a "
                                                                           "seeded rollout is queued while the previous one still runs: launches overlap), 1 = one lane, launches in order, 2 = always alternate")
    args = ap.parse_args()
    if args.interval is None:
        args.interval = 12 if "escort" in args.case else 20

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
    from muavta_amd.dist import init_abi_comm, reduce_metrics, shard_seeds
    from muavta_amd.params import params_for_case

    env = BatchedMultiUAVEnv(params_for_case(args.case), args.envs, device=local_rank)
    tile = f"{env.dims.tile_agents}x{env.dims.tile_tasks}"
    seeds = shard_seeds(rank, args.envs, args.seed_base)  # seed = global env index
    write_obs = not args.no_obs
    if args.abi_collective:
        init_abi_comm(env, rank, world)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.lanes:
        env.set_lanes(args.lanes)
    elapsed, mean_kernel_ms, mean_seed_ms = time_rollouts(env, seeds, args.interval, write_obs, args.steps, args.warmup, barrier)
    launch_gap_ms = getattr(time_rollouts, "last_gap_ms", None)
    isolated_kernel_ms = getattr(time_rollouts, "isolated_kernel_ms", None)
    lanes_mode, lanes_allocated = env.lanes()

    # metrics of the last batch: per-rank partials -> the one collective of this path (muavta_amd/dist.py).  Every env has
    # to produce a result: a capacity-flagged env (ERROR != 0) fails the run, on every rank alike (the count is reduced
    # before anybody decides).
    m = env.rollout_metrics()
    n_flagged = int(np.count_nonzero(env.get("ERROR")))
    tmax = torch.tensor([elapsed, mean_kernel_ms, mean_seed_ms, float(n_flagged)], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax[:3], op=dist.ReduceOp.MAX)
        dist.all_reduce(tmax[3:], op=dist.ReduceOp.SUM)
    elapsed, mean_kernel_ms, mean_seed_ms, n_flagged = float(tmax[0]), float(tmax[1]), float(tmax[2]), int(tmax[3])
    if n_flagged:
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(f"{n_flagged} env(s) overflowed the {tile} tile (muavta_get ERROR): results would be incomplete")
    summary = reduce_metrics(m, device="cuda" if dist is not None else None, comm=env if args.abi_collective else None)

    # secondary figures (rank 0 at N=1 only, outside the timed region)
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = secondary_figures(env, seeds, args, write_obs, barrier)

    if rank == 0:
        total_envs = args.envs * world
        value = total_envs * HORIZON * args.steps / elapsed
        out = {
            "metric": "env-steps/sec at N parallel envs, WPS_hard 16x32; 1/2/4/8 GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (reference's generative process, MT19937 seeded by global env index)",
            "config": {"workload": f"{args.case}: reset + 150 fused steps, Local-Hungarian interval {args.interval}, visibility on, "
                                   f"obs write {'on' if write_obs else 'off'}",
                       "envs_per_gpu": args.envs, "total_envs": total_envs, "tile": tile, "n_agents": env.n_agents,
                       "lds_bytes_per_env": int(env.dims.lds_bytes),
                       "horizon": HORIZON, "parallelism": f"env-sharded x{world}, RCCL all-reduce of the metric vector only"
                                      + (" (muavta_allreduce_metrics)" if args.abi_collective else " (torch.distributed nccl backend)" if dist is not None else "")},
            "roofline": roofline(args.case, args.envs, env.dims.tile_agents, mean_kernel_ms, isolated_kernel_ms, elapsed / args.steps * 1e3, lanes_allocated, env.max_tasks, env.n_agents),
            "seed_kernel_ms": mean_seed_ms,
            # where ms_per_step goes: the k_rollout launch itself (roofline.kernel_ms), the gap to the next launch on the handle's stream
            # (command processing + waiting for the next batch's seeding, from the same HIP event ring), and what is left: the host's
            # closing synchronisation and barrier, amortised over `steps` launches
            "launch_gap_ms": launch_gap_ms,
            "ms_per_step_unaccounted": (elapsed / args.steps * 1e3 - mean_kernel_ms - launch_gap_ms) if (launch_gap_ms is not None and lanes_allocated == 1) else None,
            # launches are queued back to back, so launch i+1's seeding (upload + k_seed on a second stream) runs under launch i's
            # tail; an isolated batch pays k_seed in front of the rollout kernel:
            "value_unpipelined": total_envs * HORIZON / ((mean_kernel_ms + mean_seed_ms) * 1e-3),
            "value_unpipelined_is": "whole-job env-steps/s of ONE isolated batch: envs x 150 / (k_rollout ms + k_seed ms on an idle GPU)",
            "lanes": {"mode": lanes_mode, "allocated": lanes_allocated,
                      "is": "state lanes of the ONE handle the timed loop drives (include/muavta.h: muavta_set_lanes): with 2 allocated, launch i+1 runs on the other lane and starts in the wave "
                            "slots launch i's early finishers free; roofline.kernel_ms is then the duration of an OVERLAPPED launch, roofline.isolated the launch alone on the GPU"},
            "quality": {"mean_S_WPS": summary["mean_S_WPS"], "std_S_WPS": summary["std_S_WPS"], "on_time_rate": summary["on_time_rate"],
                        "n_envs": summary["n_envs"], "capacity_flagged_envs": 0},
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["quality"].update(oracle_check(args.case, args.interval, seeds, m))
            out["cpu_baseline"] = cpu_baseline(args.case, args.interval, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def secondary_figures(env, seeds, args, write_obs, barrier):
    """Per-step paths of the headline case, BASELINE configs 4 and 5 on their tiles, the IL data loop."""
    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.params import params_for_case

    out = {}

    def mark(what):
        print(f"[bench] {what}", file=sys.stderr, flush=True)

    mark("step_api")
    # step_api: one k_allocate + one k_step launch per env step (a caller that looks at the plan);
    # fused_step_api: muavta_rollout(h, NULL, 1, ...) = allocate + step + observe in ONE launch per env step.
    env.reset(seeds)
    env.sync()
    t1 = time.perf_counter()
    for _ in range(HORIZON):
        env.allocate(args.interval, True, fetch=False)
        env.step_staged()
    env.sync()
    out["step_api_env_steps_per_s"] = args.envs * HORIZON / (time.perf_counter() - t1)
    m_step_api = env.metrics()
    env.reset(seeds)
    env.sync()
    t1 = time.perf_counter()
    for _ in range(HORIZON):
        env.rollout(None, 1, args.interval, True, write_obs)
    env.sync()
    out["fused_step_api_one_stream_env_steps_per_s"] = args.envs * HORIZON / (time.perf_counter() - t1)
    # the same per-step path with the batch split into sub-batches on their own streams (muavta_set_parts): one launch per part and
    # env step, all asynchronous — a part's launch still ends on its slowest env, but the other parts' launches fill the device
    # meanwhile (and a host-side planner would decide for one part while the others are being stepped)
    for parts in (2, 3, 4):
        mark(f"fused step api, {parts} parts")
        env.set_parts(parts)
        rate = 0.0
        for rep in range(2):  # (the first pass also creates the part streams and takes their first launches)
            env.reset(seeds)
            env.sync()
            t1 = time.perf_counter()
            for _ in range(HORIZON):
                for p in range(parts):
                    env.rollout_part(p, 1, args.interval, True, write_obs)
            t_host = time.perf_counter() - t1
            env.sync()
            rate = args.envs * HORIZON / (time.perf_counter() - t1)
            if os.environ.get("MUAVTA_BENCH_DEBUG"):
                print(f"[debug] parts {parts} rep {rep}: host {t_host * 1e3:.2f} ms total {(time.perf_counter() - t1) * 1e3:.2f} ms", file=sys.stderr)
        out[f"fused_step_api_{parts}_parts_env_steps_per_s"] = rate
    env.set_parts(0)
    # (no best-of pick: the figure is the TWO-part one — the smallest split that lets the host decide for one part while the device steps the other)
    out["fused_step_api_env_steps_per_s"], out["fused_step_api_parts"] = out["fused_step_api_2_parts_env_steps_per_s"], 2
    out["fused_step_api_is"] = ("one k_rollout(1 step) launch per sub-batch and env step, 2 sub-batches on their own streams "
                                "(muavta_rollout_part); 150 env steps of the whole batch, host-timed")
    # the same one-launch-per-step path with 8x the envs per launch (BASELINE config 3's 32768 on one GPU): a launch ends on its
    # slowest env (one that replans: ~80 us against a 24 us mean step), so a wider batch amortises that tail
    mark("fused step api, wide batch")
    try:
        n_wide = 8 * args.envs
        ew = BatchedMultiUAVEnv(params_for_case(args.case), n_wide, device=env.device_index)
        sw = np.arange(n_wide, dtype=np.uint64)
        ew.reset(sw)
        ew.rollout(None, 1, args.interval, True, write_obs)
        ew.reset(sw)
        ew.sync()
        t1 = time.perf_counter()
        for _ in range(HORIZON):
            ew.rollout(None, 1, args.interval, True, write_obs)
        ew.sync()
        out["fused_step_api_wide_env_steps_per_s"] = n_wide * HORIZON / (time.perf_counter() - t1)
        out["fused_step_api_wide_envs"] = n_wide
        ew.close()
    except Exception as exc:
        out["fused_step_api_wide_env_steps_per_s"] = None
        out["fused_step_api_wide_error"] = repr(exc)
    # (r5) the per-step path with the run-ahead of a host-side planner: k_allocate stages the plan, muavta_step_run applies it and keeps stepping
    # every env with empty actions until ITS allocator gate fires again (HungarianAllocator.should_replan: interval or any event), its episode
    # ends or 5 steps were taken — the loop of wps_eval.py:248-254,273 where the planner is only consulted at a gate
    mark("step_run")
    try:
        for rep in range(2):
            env.reset(seeds)
            env.sync()
            t1 = time.perf_counter()
            k_launch = 0
            while True:
                env.allocate(args.interval, True, fetch=False)
                _, prk, _ = env.step_run(None, None, gate="allocator", replan_interval=args.interval, max_steps=5, write_obs=write_obs)
                k_launch += 1
                if np.all(prk & 3):
                    break
            dt_sr = time.perf_counter() - t1
        out["step_run_env_steps_per_s"] = args.envs * HORIZON / dt_sr
        out["step_run_launches"] = k_launch
        out["step_run_is"] = ("k_allocate + muavta_step_run(staged plan, allocator gate, at most 5 steps per launch), the park flags read back after every launch: "
                              f"{k_launch} launch pairs for {HORIZON} env steps of {args.envs} envs (step_api_env_steps_per_s: one pair per env step)")
        if not np.array_equal(env.metrics(), m_step_api):
            raise RuntimeError("step_run ended in other metrics than the per-step path")
    except Exception as exc:
        out["step_run_env_steps_per_s"] = None
        out["step_run_error"] = repr(exc)
    # obs_ring: the fused rollout in launches of K steps whose per-step observations (+ reward, done) land in slot t of device
    # rings [K][N][...] (muavta_rollout_record) instead of overwriting one buffer — every step's observation stays readable by
    # a consumer on the device, at K steps per launch instead of one
    mark("observation rings")
    try:
        import torch

        shapes = env.obs_ring_shapes(1)
        slot_bytes = sum(int(np.prod(sh)) * np.dtype(dt).itemsize for sh, dt in shapes.values())
        dev = torch.device("cuda", env.device_index)
        for key, k_max in (("obs_ring", HORIZON), ("obs_ring_k10", 10)):
            K = max(1, min(k_max, int(32e9 // slot_bytes)))
            while HORIZON % K:
                K -= 1
            rings = {k: torch.empty(sh, dtype=getattr(torch, np.dtype(dt).name), device=dev) for k, (sh, dt) in env.obs_ring_shapes(K).items()}
            env.rollout_record(seeds, K, args.interval, True, obs_rings=rings)  # warm-up
            env.sync()
            t1 = time.perf_counter()
            for c in range(HORIZON // K):
                env.rollout_record(seeds if c == 0 else None, K, args.interval, True, obs_rings=rings)
            env.sync()
            out[key + "_env_steps_per_s"] = args.envs * HORIZON / (time.perf_counter() - t1)
            out[key + "_is"] = (f"muavta_rollout_record, {HORIZON // K} launch(es) of {K} steps per episode batch, every step's observation dict + reward "
                                f"+ done kept in device rings [{K}][{args.envs}][...] ({slot_bytes * K / 1e9:.2f} GB)")
            del rings
        torch.cuda.empty_cache()
    except Exception as exc:
        out["obs_ring_env_steps_per_s"] = None
        out["obs_ring_error"] = repr(exc)
    env.rollout(seeds, HORIZON, args.interval, True, write_obs)  # restore the headline batch's final state
    env.sync()
    # the trainers' data loop: samples = (env, step) pairs with token tensors + expert labels + step reward
    mark("il rings / il stream")
    try:
        from muavta_amd.il import il_record, il_stream

        n_il = min(args.envs, 1024)
        e3 = BatchedMultiUAVEnv(params_for_case(args.case), n_il, device=env.device_index)
        s3 = np.arange(n_il, dtype=np.uint64)
        rings = il_record(e3, s3, HORIZON, args.interval)  # warm-up (allocates the rings)
        t1 = time.perf_counter()
        for _ in range(3):
            il_record(e3, s3, HORIZON, args.interval, rings=rings)
        out["il_samples_per_s"] = 3 * n_il * HORIZON / (time.perf_counter() - t1)
        out["il_samples_per_s_is"] = (f"muavta_rollout_record: {n_il} envs x {HORIZON} steps per launch, pair tokens 32x16 + expert mask + S_WPS "
                                      "series written to device rings (no host hop)")
        t1 = time.perf_counter()
        for _t, _b in il_stream(e3, s3, 30, args.interval, with_reward=True):
            pass
        out["il_stream_per_step_api_samples_per_s"] = n_il * 30 / (time.perf_counter() - t1)
        del rings
        e3.close()
    except Exception as exc:  # torch without CUDA tensors etc.: the figure is optional
        out["il_samples_per_s"] = None
        out["il_error"] = repr(exc)
    # the RL trainers' loop with the policy IN the loop (RL half; experiments/train_pair_cost.py:132-156): per env step
    # one muavta_rl_step_device launch = Hungarian with the caller's edge scores under the trainer's gate -> env.step -> S_WPS before /
    # after -> next tokens.  The "policy" here is a fixed seeded score tensor on the GPU, so the figure times the env side of the loop.
    mark("policy in the loop")
    try:
        import torch

        dev = torch.device("cuda", env.device_index)
        e4 = BatchedMultiUAVEnv(params_for_case(args.case), args.envs, device=env.device_index)
        gen = torch.Generator(device=dev); gen.manual_seed(1234)
        scores = ((torch.rand((args.envs, 16, 32), generator=gen, device=dev) * 2 - 1) * 0.35).contiguous()
        tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}
        bufs = [{k: torch.empty(sh, dtype=tdt[dt], device=dev) for k, (sh, dt) in e4.token_shapes("pair", 32, 16).items()} for _ in range(2)]
        sel = torch.empty((args.envs, 16, 32), dtype=torch.float32, device=dev)
        rep = torch.empty((args.envs,), dtype=torch.int32, device=dev)
        sw = torch.empty((2, args.envs), dtype=torch.float64, device=dev)
        dn = torch.empty((args.envs,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for rep_i in range(2):  # (first pass: warm-up)
            e4.reset(seeds)
            e4.tokens("pair", 32, 16, out=bufs[0])
            e4.sync()
            t1 = time.perf_counter()
            for t in range(HORIZON):
                e4.rl_step("pair", 32, 16, edge_scores=scores, gate="trainer", replan_interval=args.interval, selected=sel, replanned=rep,
                           next_tok=bufs[(t + 1) & 1], s_wps=sw, done=dn)
            e4.sync()
            dt_rl = time.perf_counter() - t1
        out["policy_in_loop_per_step_env_steps_per_s"] = args.envs * HORIZON / dt_rl
        m_per_step = e4.metrics()
        # the same loop with the batch in two sub-batches on their own streams (MuavtaRlStep.part): what a trainer that runs its network on
        # part A's tokens while the device steps part B gets from the env side
        e4.set_parts(2)
        for rep_i in range(2):
            e4.reset(seeds)
            e4.tokens("pair", 32, 16, out=bufs[0])
            e4.sync()
            t1 = time.perf_counter()
            for t in range(HORIZON):
                for p_ in range(2):
                    e4.rl_step("pair", 32, 16, edge_scores=scores, gate="trainer", replan_interval=args.interval, selected=sel, replanned=rep,
                               next_tok=bufs[(t + 1) & 1], s_wps=sw, done=dn, part=p_)
            e4.sync()
            dt_rl2 = time.perf_counter() - t1
        e4.set_parts(0)
        out["policy_in_loop_per_step_2_parts_env_steps_per_s"] = args.envs * HORIZON / dt_rl2
        # (r5) run to the next gate: the reference consults the policy only when _should_replan fires (train_pair_cost.py:139-145) and steps with
        # empty actions otherwise — muavta_rl_run_device does that per env inside one launch (first step = the planned one, then quiet steps up
        # to the env's own next gate), so the policy is called once per GATE.  max_steps bounds the quiet stretch per launch (an env cut short
        # continues at the next launch without a plan): 0 = to the gate (fewest policy calls), 5 = the best env-side rate measured.
        nxt = {k: torch.empty(sh, dtype=tdt[dt], device=dev) for k, (sh, dt) in e4.token_shapes("pair", 32, 16).items()}
        nst = torch.empty((args.envs,), dtype=torch.int32, device=dev)
        prk = torch.empty((args.envs,), dtype=torch.uint8, device=dev)

        def run_ahead(max_steps, check_every):
            res = None
            for rep_i in range(2):
                e4.reset(seeds)
                e4.tokens("pair", 32, 16, out=bufs[0])
                e4.sync()
                t1 = time.perf_counter()
                k = 0
                while True:
                    e4.rl_run("pair", 32, 16, edge_scores=scores, gate="trainer", replan_interval=args.interval, selected=sel, replanned=rep, next_tok=nxt, s_wps=sw,
                              done=dn, park_tok=bufs[(k + 1) & 1], n_stepped=nst, park=prk, max_steps=max_steps)
                    k += 1
                    if k % check_every == 0:
                        e4.sync()
                        if bool(((prk & 3) != 0).all()):
                            break
                    if k > 4 * HORIZON:
                        raise RuntimeError("run-ahead did not finish")
                res = (args.envs * HORIZON / (time.perf_counter() - t1), k)
            if not np.array_equal(e4.metrics(), m_per_step):
                raise SystemExit("bench: the run-ahead policy loop ended in other metrics than the per-step loop")
            return res

        r5, k5 = run_ahead(5, 4)
        r0, k0 = run_ahead(0, 1)
        out["policy_in_loop_env_steps_per_s"] = r5
        out["policy_calls_per_env_step"] = k5 / HORIZON
        out["policy_in_loop_to_the_gate"] = {"env_steps_per_s": r0, "policy_calls_per_env_step": k0 / HORIZON, "launches": k0,
                                             "is": "max_steps 0: every env runs to its own next gate in each launch; the host checks the park flags after every launch (what il.rl_run_stream does)"}
        out["policy_in_loop_is"] = (f"muavta_rl_run_device, {k5} launches per episode batch of {args.envs} envs (at most 5 env steps per env and launch, done-check every 4 launches): per launch "
                                    "the envs parked at a gate plan (Hungarian - caller's f32 edge scores [N,16,32], trainer gate) -> step -> S_WPS before/after -> next pair tokens, "
                                    "then every env steps quietly towards its own next gate and writes the tokens it stops on; the score tensor is fixed and device-resident (times the env "
                                    "side); final metrics bit-equal to the per-step loop's (policy_in_loop_per_step_env_steps_per_s: muavta_rl_step_device, one launch per env step)")
        # the same two loops with a NETWORK in them (il.rl_stream / il.rl_run_stream: host syncs + a small torch MLP edge scorer on the token tensors,
        # hidden 32 — the shape of PairCostHybrid's MLP scorer, random weights): what the policy-call count buys end to end
        try:
            from muavta_amd.il import rl_run_stream, rl_stream

            g2 = torch.Generator(device=dev); g2.manual_seed(7)
            Wa = torch.randn((12, 32), generator=g2, device=dev) * 0.3
            Wt = torch.randn((13, 32), generator=g2, device=dev) * 0.3
            Wo = torch.randn((32,), generator=g2, device=dev) * 0.3

            def mlp_policy(tok):
                h = torch.relu((tok["agent_feats"] @ Wa).unsqueeze(2) + (tok["task_feats"] @ Wt).unsqueeze(1))
                return (torch.tanh(h @ Wo) * 0.35).contiguous()

            net = {}
            for name, gen_ in (("per_step", lambda: rl_stream(e4, seeds, mlp_policy, n_steps=HORIZON, interval=args.interval)),
                               ("run_ahead", lambda: rl_run_stream(e4, seeds, mlp_policy, interval=args.interval))):
                for rep_i in range(2):
                    t1 = time.perf_counter()
                    calls = sum(1 for _ in gen_())
                    dt_n = time.perf_counter() - t1
                net[name] = {"env_steps_per_s": args.envs * HORIZON / dt_n, "policy_calls": calls}
            out["policy_in_loop_with_network"] = net
        except Exception as exc:
            out["policy_in_loop_with_network"] = {"error": repr(exc)}
        # the same gate loop (max_steps 5) over FOUR times the envs on one handle: a launch of 4096 one-wave workgroups fits the chip at once, so every
        # launch ends on its slowest env with the early finishers' slots idle; with more workgroups than slots the early finishers are back-filled.  Extra
        # figure only (the line's policy_in_loop_env_steps_per_s stays at --envs); its first --envs rows must equal the per-step loop's metrics.
        try:
            wide_n = 4 * args.envs
            e5 = BatchedMultiUAVEnv(params_for_case(args.case), wide_n, device=env.device_index)
            seeds_w = np.concatenate([seeds, np.arange(int(seeds[-1]) + 1, int(seeds[-1]) + 1 + wide_n - args.envs, dtype=np.uint64)]).astype(np.uint64)
            sc_w = scores.repeat(4, 1, 1).contiguous()
            shp = e5.token_shapes("pair", 32, 16)
            tok_w = [{k: torch.empty(sh, dtype=tdt[dt], device=dev) for k, (sh, dt) in shp.items()} for _ in range(3)]
            sel_w = torch.empty((wide_n, 16, 32), dtype=torch.float32, device=dev)
            rep_w = torch.empty((wide_n,), dtype=torch.int32, device=dev)
            sw_w = torch.empty((2, wide_n), dtype=torch.float64, device=dev)
            dn_w = torch.empty((wide_n,), dtype=torch.uint8, device=dev)
            nst_w = torch.empty((wide_n,), dtype=torch.int32, device=dev)
            prk_w = torch.empty((wide_n,), dtype=torch.uint8, device=dev)
            for rep_i in range(2):
                e5.reset(seeds_w)
                e5.tokens("pair", 32, 16, out=tok_w[0])
                e5.sync()
                t1 = time.perf_counter()
                k = 0
                while True:
                    e5.rl_run("pair", 32, 16, edge_scores=sc_w, gate="trainer", replan_interval=args.interval, selected=sel_w, replanned=rep_w, next_tok=tok_w[2],
                              s_wps=sw_w, done=dn_w, park_tok=tok_w[(k + 1) & 1], n_stepped=nst_w, park=prk_w, max_steps=5)
                    k += 1
                    if k % 4 == 0:
                        e5.sync()
                        if bool(((prk_w & 3) != 0).all()):
                            break
                    if k > 4 * HORIZON:
                        raise RuntimeError("run-ahead did not finish")
                dt_w = time.perf_counter() - t1
            same = bool(np.array_equal(e5.metrics()[:args.envs], m_per_step))
            out["policy_in_loop_wide"] = {"env_steps_per_s": wide_n * HORIZON / dt_w if same else None, "envs": wide_n, "policy_calls_per_env_step": k / HORIZON,
                                          "first_rows_equal_per_step_loop": same,
                                          "is": "policy_in_loop_env_steps_per_s's loop over 4x the envs on one handle (more workgroups than the chip holds at once: envs that park early are back-filled)"}
            e5.close()
            del tok_w, sel_w, sc_w
        except Exception as exc:
            out["policy_in_loop_wide"] = {"error": repr(exc)}
        n_flag = int(np.count_nonzero(e4.get("ERROR")))
        out["policy_in_loop_capacity_flagged_envs"] = n_flag
        out["policy_in_loop_mean_S_WPS"] = float(e4.metrics()[:, 4].mean()) if not n_flag else None
        e4.close()
        del bufs, sel, scores
    except Exception as exc:
        out["policy_in_loop_env_steps_per_s"] = None
        out["policy_in_loop_error"] = repr(exc)
    # the headline loop again on a handle pinned to ONE state lane (launches in order, nothing overlapping: the only mode before round 5)
    mark("one lane")
    try:
        env.rollout(seeds, HORIZON, args.interval, True, write_obs)
        env.sync()
        tf = one_lane_figure(args.case, args.envs, args.interval, seeds, write_obs, max(args.steps, 8), 4, barrier, env.device_index, env.rollout_metrics())
        out["value_one_lane"] = tf["env_steps_per_s"]
        out["value_one_lane_is"] = (f"the headline loop on a handle with muavta_set_lanes(h, 1): {tf['ms_per_step']:.3f} ms per launch of {args.envs} envs x {HORIZON} steps, k_rollout "
                                    f"{tf['kernel_ms']:.3f} ms, gap to the next launch {tf['launch_gap_ms']}; its batch is bit-equal to the default handle's")
    except Exception as exc:
        out["value_one_lane"] = None
        out["value_one_lane_error"] = repr(exc)
    # the drop-in PettingZoo facade (muavta_amd.env.MultiUAVEnv over the HIP backend) in the loop shape of experiments/wps_eval.py:112-133,273:
    # per step the harness reads the live agents, the open tasks (env.tasks filtered by status) and the visibility map, gets its plan as
    # [(agent name, Task)] — here the device allocator's plan read back, in place of the reference's Python HungarianAllocator, which cannot travel
    # to this box — maps it through env.last_tasks_info like _apply_assign, and calls env.step(actions dict) -> observation dicts.
    mark("facade")
    try:
        out.update(facade_figures(env.device_index))
    except Exception as exc:
        out["facade_steps_per_s"] = None
        out["facade_error"] = repr(exc)
    if args.case == "WPS_hard_x2":
        tiles = {}
        for case, n, interval in OTHER_TILES:
            mark(f"other tile {case}")
            e2 = BatchedMultiUAVEnv(params_for_case(case), n, device=env.device_index)
            s2 = np.arange(n, dtype=np.uint64)
            el, kms, sms = time_rollouts(e2, s2, interval, True, 8, 4, barrier)
            flagged = int(np.count_nonzero(e2.get("ERROR")))
            tiles[case] = {"env_steps_per_s": n * HORIZON * 8 / el, "envs": n, "tile": f"{e2.dims.tile_agents}x{e2.dims.tile_tasks}",
                           "interval": interval, "ms_per_step": el / 8 * 1e3, "lds_bytes_per_env": int(e2.dims.lds_bytes),
                           "capacity_flagged_envs": flagged, "seed_kernel_ms": sms, "lanes_allocated": e2.lanes()[1],
                           "roofline": roofline(case, n, e2.dims.tile_agents, kms, getattr(time_rollouts, "isolated_kernel_ms", None), el / 8 * 1e3, e2.lanes()[1], e2.max_tasks, e2.n_agents)}
            try:
                tiles[case]["one_lane"] = one_lane_figure(case, n, interval, s2, True, 8, 4, barrier, env.device_index, e2.rollout_metrics())
            except Exception as exc:
                tiles[case]["one_lane"] = {"env_steps_per_s": None, "error": repr(exc)}
            e2.close()
        out["other_tiles"] = tiles
    return out


if __name__ == "__main__":
    main()
