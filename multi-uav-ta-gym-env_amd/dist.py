"""Multi-GPU sharding of the batched env: env instances are independent, so rank r simply
owns global env indices [r*n, (r+1)*n) — seed == global index — and the ONLY collective is one
all-reduce of a small per-rank metric vector at the end of a batch (RCCL over xGMI with backend
"nccl"; "gloo" in the CPU tests).  Integer counters travel as int64 so the totals are exact; float
sums are reduced in rank order via all_gather so that the result does not depend on the ring order."""
from __future__ import annotations

from typing import Dict

import numpy as np

from .params import METRIC_KEYS

SUM_KEYS = ["S_WPS", "S_ESC", "total_distance", "F_Reward"]
COUNT_KEYS = ["n_on_time", "n_missed_windows", "n_windowed_tasks", "n_task_switches", "n_arrivals", "Losses", "Kills",
              "threats_intercepted", "recon_losses", "protected_rec_completed"]


def shard_seeds(rank: int, envs_per_rank: int, base: int = 0) -> np.ndarray:
    """Seeds of the env instances rank `rank` owns: the global env index (+ base)."""
    return np.arange(base + rank * envs_per_rank, base + (rank + 1) * envs_per_rank, dtype=np.uint64)


def partial_sums(metrics: np.ndarray):
    """Per-rank partials of an [n, 30] metrics array: (float sums incl. sum of squares of S_WPS, int counters)."""
    K = {k: i for i, k in enumerate(METRIC_KEYS)}
    f = [float(metrics[:, K[k]].sum()) for k in SUM_KEYS] + [float((metrics[:, K["S_WPS"]] ** 2).sum())]
    c = [int(metrics[:, K[k]].sum()) for k in COUNT_KEYS] + [int(metrics.shape[0])]
    return np.array(f, dtype=np.float64), np.array(c, dtype=np.int64)


def init_abi_comm(env, rank: int, world: int):
    """Join the RCCL communicator of the C ABI (muavta_comm_init): rank 0 creates the unique id, torch.distributed (already
    initialised by the launcher) only carries its 128 bytes to the other ranks."""
    import torch.distributed as dist

    box = [env.comm_uid() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    env.comm_init(rank, world, box[0])


def reduce_metrics(metrics: np.ndarray, device=None, comm=None) -> Dict[str, float]:
    """All ranks call this with their shard's metrics; every rank gets the whole-job summary.  `comm`: a
    BatchedMultiUAVEnv whose muavta_comm_init has run -> the reduction goes through muavta_allreduce_metrics (RCCL behind
    the C ABI) instead of torch.distributed."""
    f, c = partial_sums(metrics)
    if comm is not None:
        f, c = comm.allreduce_metrics(f, c)
        return _summary(f, c)
    import torch
    import torch.distributed as dist

    tf = torch.from_numpy(f)
    tc = torch.from_numpy(c)
    if device is not None:
        tf, tc = tf.to(device), tc.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.empty_like(tf) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, tf)                 # float partials: fixed (rank) summation order
        tf = torch.stack(parts).sum(dim=0)
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)  # exact integer totals
    return _summary(tf.cpu().numpy(), tc.cpu().numpy())


def _summary(f: np.ndarray, c: np.ndarray) -> Dict[str, float]:
    n = int(c[-1])
    out = {f"sum_{k}": float(v) for k, v in zip(SUM_KEYS, f)}
    out.update({k: int(v) for k, v in zip(COUNT_KEYS, c)})
    out["n_envs"] = n
    out["mean_S_WPS"] = float(f[0] / n)
    out["std_S_WPS"] = float(np.sqrt(max(f[-1] / n - (f[0] / n) ** 2, 0.0)))
    out["on_time_rate"] = float(c[0] / max(c[0] + c[1], 1))
    return out
