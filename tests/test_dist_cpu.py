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
