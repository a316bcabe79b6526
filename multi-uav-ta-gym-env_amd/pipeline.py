"""Several batches in flight on one GPU (round 4).

A fused rollout launch lasts as long as its slowest env: on the 24- and 64-agent tiles a quarter of the wave slots idle while the
heaviest envs finish (`profiles/r03_end_times.txt`).  Handles are independent (own stream, own env records, own seeding slots), so
launches that alternate between two handles overlap on the device — batch i+1's workgroups start in the slots batch i's early
finishers free: +17 % on config 4, +37 % on config 5, +3 % on config 2 (`profiles/r04_inflight_probe.txt`).  This class is that
pattern behind the batched API: `submit()` queues a whole batch on the next handle, `results()` yields the finished batches in order.

The reference has no counterpart (it runs one Python env at a time, experiments/wps_eval.py:76-291); results are the handles' own,
bit-identical to a single handle's (each batch is one `muavta_rollout`).
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, Iterator, Optional, Tuple

import numpy as np

from .batched import BatchedMultiUAVEnv, MuavtaError


class InFlightRollouts:
    def __init__(self, config, n_envs: int, handles: int = 2, device: int = 0, allocator: str = "hungarian", **kw):
        if handles < 1:
            raise ValueError("handles >= 1")
        self.envs = [BatchedMultiUAVEnv(config, n_envs, device=device, **kw) for _ in range(handles)]
        for e in self.envs:
            e.set_allocator(allocator)
        self.n_envs = n_envs
        self._next = 0
        self._pending: deque = deque()  # (handle index, tag)

    def close(self):
        for e in self.envs:
            e.close()
        self.envs = []

    def submit(self, seeds, n_steps: int = 150, replan_interval: int = 20, use_visibility: bool = True, write_obs: bool = True, tag=None) -> None:
        """Queue reset(seeds) + n_steps fused steps for one batch on the next handle (asynchronous).  A handle whose previous batch
        has not been collected yet is collected first: at most `handles` batches are in flight."""
        k = self._next
        if any(h == k for h, _ in self._pending):
            raise MuavtaError("InFlightRollouts.submit: every handle holds an uncollected batch — take results() first")
        self.envs[k].rollout(seeds, n_steps, replan_interval, use_visibility, write_obs)
        self._pending.append((k, tag))
        self._next = (k + 1) % len(self.envs)

    def results(self, all_pending: bool = False) -> Iterator[Tuple[object, np.ndarray]]:
        """(tag, metrics [n_envs, 30]) of the oldest queued batch — or of every queued batch, oldest first.  Blocks for each.
        An env that overflowed its tile fails the batch loudly, as `metrics()` does."""
        while self._pending:
            k, tag = self._pending.popleft()
            e = self.envs[k]
            m = e.rollout_metrics()  # (synchronises this handle's stream)
            if np.count_nonzero(e.get("ERROR")):
                raise MuavtaError("an env of the batch overflowed its tile (muavta_get ERROR): use BatchedMultiUAVEnv.rollout(escalate=True) for such workloads")
            yield tag, m
            if not all_pending:
                return

    def run(self, seed_batches: Iterable, n_steps: int = 150, replan_interval: int = 20, use_visibility: bool = True, write_obs: bool = True):
        """Metrics of every seed batch, in order, with up to `handles` batches in flight."""
        out = []
        for i, seeds in enumerate(seed_batches):
            if len(self._pending) == len(self.envs):
                out.extend(m for _, m in self.results())
            self.submit(seeds, n_steps, replan_interval, use_visibility, write_obs, tag=i)
        out.extend(m for _, m in self.results(all_pending=True))
        return out
