"""N>1 path on CPU: two gloo ranks shard the env indices, each simulates its shard (oracle stand-in for the
GPU, test-only), and the one collective of the path — the metric reduction — gives the whole-job totals."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

import orc
from muavta_amd.dist import partial_sums, reduce_metrics, shard_seeds
from muavta_amd.params import params_for_case

CASE, ENVS_PER_RANK, WORLD = "WPS_hard_x2", 6, 2


def _simulate(seeds):
    o = orc.OracleEnv(params_for_case(CASE))
    rows = []
    for s in seeds:
        o.rollout(int(s), 150, 20, 1)
        rows.append(o.metrics())
    return np.stack(rows)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _simulate(shard_seeds(rank, ENVS_PER_RANK))
    out = reduce_metrics(m)
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_rank_sharding_and_metric_reduction():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert results[0] == results[1]  # every rank holds the same whole-job summary
    # single-process truth over the union of the shards
    assert list(shard_seeds(1, ENVS_PER_RANK)) == list(range(ENVS_PER_RANK, 2 * ENVS_PER_RANK))
    whole = _simulate(np.arange(WORLD * ENVS_PER_RANK))
    f0, c0 = partial_sums(whole[:ENVS_PER_RANK])
    f1, c1 = partial_sums(whole[ENVS_PER_RANK:])
    r = results[0]
    assert r["n_envs"] == WORLD * ENVS_PER_RANK
    assert r["n_on_time"] == int(c0[0] + c1[0]) and r["n_missed_windows"] == int(c0[1] + c1[1])
    assert r["sum_S_WPS"] == float(f0[0] + f1[0])  # rank-ordered float reduction: bit-stable
    assert abs(r["mean_S_WPS"] - whole[:, 4].mean()) < 1e-9


SHARDS4 = [5, 2, 7, 3]  # unequal shard sizes: rank r owns a contiguous block of SHARDS4[r] global env indices


def _worker4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo = sum(SHARDS4[:rank])
    m = _simulate(np.arange(lo, lo + SHARDS4[rank]))
    out = reduce_metrics(m)
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_four_ranks_with_unequal_shards():
    """reduce_metrics over four gloo ranks whose shards differ in size: every rank ends with the same summary, the env count is
    the sum of the shard sizes, integer totals are exact and the float sums are the rank-ordered sums of the partials."""
    world = len(SHARDS4)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(results[r] == results[0] for r in range(1, world))
    whole = _simulate(np.arange(sum(SHARDS4)))
    parts = [partial_sums(whole[sum(SHARDS4[:r]):sum(SHARDS4[:r + 1])]) for r in range(world)]
    f = parts[0][0].copy()
    for pf, _ in parts[1:]:
        f = f + pf                                   # rank order, as torch.stack(parts).sum(dim=0) adds them
    c = sum(pc for _, pc in parts)
    r = results[0]
    assert r["n_envs"] == sum(SHARDS4) == int(c[-1])
    assert r["n_on_time"] == int(c[0]) and r["n_missed_windows"] == int(c[1]) and r["Kills"] == int(whole[:, 7].sum())
    assert abs(r["sum_S_WPS"] - float(f[0])) <= 1e-9 * max(1.0, abs(float(f[0])))
    assert abs(r["mean_S_WPS"] - whole[:, 4].mean()) < 1e-9


def _run_ranks(world, backend, jobs, timeout=600):
    import dist_workers
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=dist_workers.fuzz_rank, args=(r, world, port, q, backend, jobs)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return results


def check_rank_results(results, world, jobs, truth_backend="oracle"):
    """every rank holds the same summary of every job, and it is the rank-ordered reduction of the single-process truth"""
    import dist_workers
    for r in range(1, world):  # (the third field is the rank's OWN capacity-flagged global indices: those differ by construction)
        assert [x[:2] for x in results[r]] == [x[:2] for x in results[0]], f"rank {r} ended with another summary than rank 0"
    for (k, cfg, tile, shards, interval, steps), (k2, got, flagged) in zip(jobs, results[0]):
        assert k == k2
        whole, _ = dist_workers.shard_metrics(truth_backend, cfg, tile, 0, sum(shards), interval, steps)
        flagged_all = sorted(i for r in range(world) for i in results[r][[j[0] for j in jobs].index(k)][2])
        parts, lo = [], 0
        for n in shards:
            keep = [i for i in range(lo, lo + n) if i not in flagged_all]
            parts.append(partial_sums(whole[keep].reshape(-1, 30)))
            lo += n
        f = parts[0][0].copy()
        for pf, _ in parts[1:]:
            f = f + pf                      # rank order, as reduce_metrics adds the gathered partials
        c = sum(pc for _, pc in parts)
        assert got["n_envs"] == int(c[-1]) == sum(shards) - len(flagged_all), f"k={k}: env count"
        assert got["n_on_time"] == int(c[0]) and got["n_missed_windows"] == int(c[1]) and got["Kills"] == int(c[6]), f"k={k}: integer totals"
        assert got["sum_S_WPS"] == float(f[0]) and got["sum_total_distance"] == float(f[2]), f"k={k}: rank-ordered float sums are bit-stable"
        if got["n_envs"]:
            assert got["mean_S_WPS"] == float(f[0] / got["n_envs"])


def fuzz_jobs(first_k, n_jobs, world, rng, max_agents=16):
    from fuzz_reference import wide_config
    jobs, k = [], first_k
    while len(jobs) < n_jobs:
        w = wide_config(k)
        cfg = w["cfg"]
        na, nh = sum(cfg["agents"].values()), sum(n for _, n in cfg["threats_list"])
        tile = (16, 40, 16) if (na <= 16 and nh <= 16) else (24, 48, 24) if (na <= 24 and nh <= 24) else (64, 128, 48)
        if na <= max_agents:
            shards = [int(x) for x in rng.integers(0 if world > 2 else 1, 6, world)]
            if sum(shards) == 0:
                shards[0] = 1
            jobs.append((k, cfg, tile, shards, w["interval"], min(int(cfg["max_time_steps"]), 150)))
        k += 1
    return jobs


def test_random_configurations_unequal_and_empty_shards_2_to_4_ranks():
    """The N > 1 leg of the wide fuzz (container: oracle stand-in for the device): 2, 3 and 4 gloo ranks, shard sizes drawn per configuration
    (unequal, some EMPTY — a rank that owns no env of a batch still joins the collective), random wide_config draws with their own replan
    interval and horizon, through dist.reduce_metrics; every rank ends with the same summary = the rank-ordered reduction of the truth."""
    rng = np.random.default_rng(77)
    for world, first in ((2, 50000), (3, 50100), (4, 50200)):
        jobs = fuzz_jobs(first, 5, world, rng)
        check_rank_results(_run_ranks(world, "oracle", jobs), world, jobs)


def test_comm_abi_argument_paths_without_a_device():
    """muavta_comm_* / muavta_allreduce_metrics reject bad arguments before they touch a device or RCCL (the same checks the N > 1 path
    meets on a GPU box): NULL handle, NULL uid, counts beyond 64, missing buffers."""
    import ctypes as C
    from muavta_amd import native
    L = native.lib()
    assert L.muavta_comm_uid(None) == -1
    assert L.muavta_comm_init(None, 0, 2, None) == -1
    assert L.muavta_allreduce_metrics(None, None, 0, None, 0, None, None) == -1
    assert L.muavta_comm_destroy(None) == -1
    assert L.muavta_set_lanes(None, 2) == -1 and L.muavta_rollout_metrics_back(None, 1, None) == -1
