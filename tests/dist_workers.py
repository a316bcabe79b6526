"""Worker processes of the N > 1 tests (spawned: importable top-level functions).  Rank r owns a contiguous block of global env indices
(seed = global index); `backend` "oracle" = the CPU stand-in of the container tests, "hip" = the product on the one GPU of the box (every rank
its own handle on device 0 — the collective runs over gloo either way: RCCL wants one GPU per rank)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def shard_metrics(backend, cfg, tile, lo, n, interval, steps):
    from fuzz_device_params import params_of_wide
    p = params_of_wide(cfg, tile)
    seeds = np.arange(lo, lo + n, dtype=np.uint64)
    if n == 0:
        return np.zeros((0, 30)), np.zeros(0, dtype=np.int32)
    if backend == "hip":
        from muavta_amd.batched import BatchedMultiUAVEnv
        env = BatchedMultiUAVEnv(p, n, device=0)
        env.rollout(seeds, steps, interval, True, True)
        m, err = env.rollout_metrics(), env.get("ERROR")
        env.close()
        return m, err
    import orc
    o = orc.OracleEnv(p)
    rows = []
    for s in seeds:
        o.rollout(int(s), steps, interval, 1)
        rows.append(o.metrics())
    return np.stack(rows), np.zeros(n, dtype=np.int32)


def fuzz_rank(rank, world, port, q, backend, jobs):
    """jobs: [(k, cfg, tile, shard sizes, interval, steps)] — every rank walks the same list and reduces its shard of each"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from muavta_amd.dist import reduce_metrics
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for k, cfg, tile, shards, interval, steps in jobs:
        lo = sum(shards[:rank])
        m, err = shard_metrics(backend, cfg, tile, lo, shards[rank], interval, steps)
        m = m[err == 0] if len(err) else m   # (a capacity-flagged env carries no result: dropped on both sides of the comparison)
        out.append((k, reduce_metrics(m), [int(i) + lo for i in np.nonzero(err)[0]]))
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()
