"""Batched imitation-/reinforcement-learning data loop over the device env.

The reference trains its pair-cost hybrids one Python env at a time (experiments/train_pair_cost.py:96-156):
at every replan step the expert (Global-Hungarian, `force=True`) plans, `build_pair_tokens` featurises the
state, `_expert_mask` turns the expert's pairs into the label matrix, and the episode follows the expert; the RL
phase rewards each step with `(S_WPS_now - S_WPS_prev) / 20`.  Here the same loop runs for thousands of envs per
launch: expert = `muavta_allocate` in MUAVTA_ALLOC_HUNGARIAN_GATED mode without the visibility mask, tokens +
labels = one `k_tokens` launch (optionally straight into torch tensors), reward from `muavta_metrics`.
The learner itself (the torch nets of TaskAllocation/Hybrid) is the caller's; this module is its data path.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import numpy as np

from .batched import BatchedMultiUAVEnv

S_WPS_COL = 4  # METRIC_KEYS.index("S_WPS")


def il_stream(env: BatchedMultiUAVEnv, seeds, n_steps: int = 150, interval: int = 20, kind: str = "pair",
              max_tasks: int = 32, max_agents: int = 16, out: Optional[dict] = None,
              with_reward: bool = False) -> Iterator[Tuple[int, Dict[str, np.ndarray]]]:
    """run_il_episode (train_pair_cost.py:96-129) for every env of the batch at once.

    Yields `(t, batch)` before each env step: `batch` holds the token tensors of `env.tokens(kind, ...)` plus
    `expert_mask` [N, max_agents, max_tasks] (= `_expert_mask(tok, expert)`) and `replanned` [N] (1 where that env's
    trainer gate fired at step t: only those rows are training samples).  The episode then follows the expert.
    With `with_reward`, `batch["step_reward"]` of the PREVIOUS step (`(S_WPS_now - S_WPS_prev) / 20`, :146-148) is added.
    """
    env.set_allocator("hungarian_gated")
    env.reset(np.asarray(seeds, dtype=np.uint64))
    prev = env.metrics()[:, S_WPS_COL] if with_reward else None
    for t in range(n_steps):
        env.allocate(interval, False, fetch=False)          # Global-Hungarian expert, staged on the device
        batch = env.tokens(kind, max_tasks, max_agents, out=out)
        if with_reward:
            now = env.metrics()[:, S_WPS_COL]
            batch = dict(batch)
            batch["step_reward"] = (now - prev) / 20.0
            prev = now
        yield t, batch
        env.step_staged()                                   # the rollout follows the expert (:126-127)


def step_rewards(s_wps_prev: np.ndarray, s_wps_now: np.ndarray) -> np.ndarray:
    """RL step reward of run_rl_episode (train_pair_cost.py:146-148)."""
    return (np.asarray(s_wps_now) - np.asarray(s_wps_prev)) / 20.0
