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
