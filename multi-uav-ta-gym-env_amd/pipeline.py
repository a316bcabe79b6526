"""Several episode batches in flight on one GPU.

A fused rollout launch lasts as long as its slowest env: on the 24- and 64-agent tiles a quarter of the wave slots idle while the
heaviest envs finish (`profiles/r03_end_times.txt`).  Since round 5 that overlap lives INSIDE the C ABI: a handle owns two state lanes
(include/muavta.h: `muavta_set_lanes`), a seeded rollout queued while the previous one still runs goes to the other lane, and
`muavta_rollout_metrics_back(h, 1, ...)` reaches the batch before the latest one — +3 % / +17 % / +37 % on BASELINE configs 2 / 4 / 5
(`profiles/r04_inflight_probe.txt`, round 4: two handles from Python).  This class is the submit / collect pattern over ONE handle
(`handles` > 1 adds more handles, each with its two lanes: only useful to hide a slow host loop).

The reference has no counterpart (it runs one Python env at a time, experiments/wps_eval.py:76-291, 528-546: episodes x seeds);
results are bit-identical to one `muavta_rollout` per batch.
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, Iterator, Tuple

import numpy as np

from .batched import BatchedMultiUAVEnv, MuavtaError


class InFlightRollouts:
    LANES = 2  # batches a handle keeps in flight

    def __init__(self, config, n_envs: int, handles: int = 1, device: int = 0, allocator: str = "hungarian", **kw):
        if handles < 1:
            raise ValueError("handles >= 1")
        self.envs = [BatchedMultiUAVEnv(config, n_envs, device=device, **kw) for _ in range(handles)]
        for e in self.envs:
            e.set_allocator(allocator)
            e.set_lanes(2)
        self.n_envs = n_envs
        self._next = 0
        self._pending: deque = deque()  # (handle index, tag), oldest first
        self.depth = handles * self.LANES

    def close(self):
        for e in self.envs:
            e.close()
        self.envs = []

    def submit(self, seeds, n_steps: int = 150, replan_interval: int = 20, use_visibility: bool = True, write_obs: bool = True, tag=None) -> None:
        """Queue reset(seeds) + n_steps fused steps for one batch (asynchronous): on the next handle's other lane.  At most `depth`
        (= handles x 2 lanes) batches may be uncollected: one more raises MuavtaError — take `results()` first."""
        if len(self._pending) >= self.depth:
            raise MuavtaError("InFlightRollouts.submit: every lane holds an uncollected batch — take results() first")
        k = self._next
        self.envs[k].rollout(seeds, n_steps, replan_interval, use_visibility, write_obs)
        self._pending.append((k, tag))
        self._next = (k + 1) % len(self.envs)

    def results(self, all_pending: bool = False) -> Iterator[Tuple[object, np.ndarray]]:
        """(tag, metrics [n_envs, 30]) of the oldest queued batch — or of every queued batch, oldest first.  Blocks for each.
        An env that overflowed its tile fails the batch loudly, as `metrics()` does; the batch stays queued in that case."""
        while self._pending:
            k, tag = self._pending[0]
            e = self.envs[k]
            back = sum(1 for h, _ in self._pending if h == k) - 1  # batches this handle launched after the one asked for: 0 or 1
            m = e.rollout_metrics(back=back)  # (synchronises that lane's stream)
            if np.count_nonzero(e.error_flags(back=back)):
                raise MuavtaError("an env of the batch overflowed its tile (muavta_get ERROR): use BatchedMultiUAVEnv.rollout(escalate=True) for such workloads")
            self._pending.popleft()
            yield tag, m
            if not all_pending:
                return

    def run(self, seed_batches: Iterable, n_steps: int = 150, replan_interval: int = 20, use_visibility: bool = True, write_obs: bool = True):
        """Metrics of every seed batch, in order, with up to `depth` batches in flight."""
        out = []
        for i, seeds in enumerate(seed_batches):
            if len(self._pending) == self.depth:
                out.extend(m for _, m in self.results())
            self.submit(seeds, n_steps, replan_interval, use_visibility, write_obs, tag=i)
        out.extend(m for _, m in self.results(all_pending=True))
        return out
