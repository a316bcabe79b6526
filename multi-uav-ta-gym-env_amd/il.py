"""Batched imitation-/reinforcement-learning data loop over the device env.

The reference trains its pair-cost hybrids one Python env at a time (experiments/train_pair_cost.py:96-156):
at every replan step the expert (Global-Hungarian, `force=True`) plans, `build_pair_tokens` featurises the
state, `_expert_mask` turns the expert's pairs into the label matrix, and the episode follows the expert; the RL
phase rewards each step with `(S_WPS_now - S_WPS_prev) / 20`.  Here the same loop runs for thousands of envs per
launch: expert = `muavta_allocate` in MUAVTA_ALLOC_HUNGARIAN_GATED mode without the visibility mask, tokens +
labels = one `k_tokens` launch (optionally straight into torch tensors), reward from `muavta_metrics`.
`rl_stream` closes the loop for the RL phase (run_rl_episode, :132-156): the caller's network turns the token tensors into edge
scores on the same GPU and `muavta_rl_step_device` plans with them (Hungarian + scores), steps and emits the next tokens.
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
        if out is not None:
            env.sync()  # k_tokens ran on the handle's own (non-blocking) stream: the caller's stream must not race it
        yield t, batch
        env.step_staged()                                   # the rollout follows the expert (:126-127)


def il_record(env: BatchedMultiUAVEnv, seeds, n_steps: int = 150, interval: int = 20, kind: str = "pair", max_tasks: int = 32,
              max_agents: int = 16, rings: Optional[dict] = None, device=None) -> dict:
    """The same data as `il_stream(..., with_reward=True)` for the WHOLE episode batch in ONE launch (muavta_rollout_record):
    returns CUDA torch tensors `[n_steps, N, ...]` (token tensors, `expert_mask`, `replanned`), `s_wps [n_steps + 1, N]` and
    `step_reward [n_steps, N]` = `(s_wps[t+1] - s_wps[t]) / 20`, the reward of the step taken after sample t
    (train_pair_cost.py:146-148).  Nothing crosses PCIe except the 8-byte seeds; a learner slices the rings on the device.
    `rings`: reuse tensors of a previous call."""
    import torch

    dev = torch.device("cuda", env.device_index if device is None else device)
    shapes = env.record_shapes(kind, n_steps, max_tasks, max_agents)
    if rings is None:
        tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32, np.float64: torch.float64}
        rings = {name: torch.empty(shape, dtype=tdt[dtype], device=dev) for name, (shape, dtype) in shapes.items()}
    env.set_allocator("hungarian_gated")
    # the rings live in torch's caching allocator and are written by the handle's own (non-blocking) stream: whatever torch's
    # current stream still has queued on these blocks — a previous training step reading the old batch, or the kernels that
    # used a block the allocator has just recycled — must be done before the rollout overwrites them
    env.wait_stream(torch.cuda.current_stream(dev).cuda_stream)
    env.rollout_record(np.asarray(seeds, dtype=np.uint64), n_steps, interval, False, rings, kind, max_tasks, max_agents)
    env.sync()  # ... and the read-after-write direction: the rings are complete before torch's stream touches them
    out = dict(rings)
    diff = rings["s_wps"][1:] - rings["s_wps"][:-1]
    out["step_reward"] = torch.div(diff, torch.full_like(diff, 20.0))  # tensor / tensor: IEEE division (a scalar divisor becomes a multiplication by its reciprocal)
    return out


def rl_stream(env: BatchedMultiUAVEnv, seeds, policy, n_steps: int = 150, interval: int = 20, kind: str = "pair", max_tasks: int = 32,
              max_agents: int = 16, gate: str = "trainer", device=None, fused: bool = True, run_ahead: bool = False, max_steps: int = 0, **plan_kw):
    """run_rl_episode (experiments/train_pair_cost.py:132-156) for every env of the batch at once, the policy in the loop:

        for t, tr in rl_stream(env, seeds, policy): buffer.push(tr)          # policy.push(tok, scores, ..., step_r, next_tok, done)

    `policy(tok) -> edge_scores`: a callable on the SAME GPU that maps the token tensors (dict of CUDA torch tensors: task_feats,
    task_mask, agent_feats, agent_mask, edge_valid, ...; the layout of `env.tokens(kind, max_tasks, max_agents)`) to a contiguous
    float32 CUDA tensor [N, max_agents, max_tasks] — what `PairCostHybrid.act` returns (`tanh(logits) * score_clamp`), batched.
    Each step: scores -> `muavta_rl_step_device` (Hungarian with the scores under the trainer's gate, env.step, S_WPS before /
    after, next tokens) — one launch, nothing crosses PCIe.  Yields `(t, transition)` with CUDA tensors: `tok`, `scores`,
    `selected` (_selected_mask), `replanned` (rows that are RL samples: the reference pushes only when it planned), `step_reward`
    ((S_WPS_now - S_WPS_prev) / 20), `next_tok`, `done` (u8: bit 0 terminated, bit 1 truncated).  The token dicts alternate
    between two buffer sets: `tok` of step t is `next_tok` of step t - 1 and is overwritten at step t + 1 — copy what must live
    longer.  `fused=False` runs the same step as four separate launches (tokens / allocate_scored / step / metrics): the
    cross-check of the fused kernel WHILE EVERY ENV IS LIVE — the separate launches keep planning and stepping an env whose episode has
    ended (as four separate calls on the reference would), the fused kernel leaves it alone; the yielded rows of ended envs are masked the
    same way in both modes (replanned 0, selected 0, reward 0), their final `env.metrics()` are not comparable between the modes.  Envs whose episode has ended idle (replanned 0, reward 0); `selected`, `replanned`, `done` and the
    reward tensors are the SAME tensors at every yield, overwritten by the next step — clone what must live longer (like the token dicts).
    `run_ahead=True`: the policy is consulted once per GATE instead of once per step (`rl_run_stream` below; `n_steps` then bounds the
    number of launches)."""
    import torch

    if run_ahead:
        yield from rl_run_stream(env, seeds, policy, interval=interval, kind=kind, max_tasks=max_tasks, max_agents=max_agents, gate=gate, device=device,
                                 max_steps=max_steps, max_launches=n_steps, **plan_kw)
        return
    dev = torch.device("cuda", env.device_index if device is None else device)
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32, np.float64: torch.float64}
    shapes = env.token_shapes(kind, max_tasks, max_agents)
    N = env.n_envs
    bufs = [{name: torch.empty(shape, dtype=tdt[dtype], device=dev) for name, (shape, dtype) in shapes.items()} for _ in range(2)]
    selected = torch.empty((N, max_agents, max_tasks), dtype=torch.float32, device=dev)
    replanned = torch.empty((N,), dtype=torch.int32, device=dev)
    s_wps = torch.empty((2, N), dtype=torch.float64, device=dev)
    done = torch.empty((N,), dtype=torch.uint8, device=dev)
    twenty = torch.full((N,), 20.0, dtype=torch.float64, device=dev)
    env.reset(np.asarray(seeds, dtype=np.uint64))
    env.tokens(kind, max_tasks, max_agents, out=bufs[0])   # tok of step 0
    ended = None
    for t in range(n_steps):
        tok, nxt = bufs[t & 1], bufs[(t + 1) & 1]
        env.sync()                                       # the handle's stream wrote `tok`; the policy runs on torch's stream
        scores = policy(tok)
        if scores.dtype != torch.float32 or not scores.is_contiguous():
            scores = scores.to(torch.float32).contiguous()
        env.wait_stream(torch.cuda.current_stream(dev).cuda_stream)   # ... and the scores must be complete before the plan reads them
        if fused:
            env.rl_step(kind, max_tasks, max_agents, edge_scores=scores, gate=gate, replan_interval=interval, selected=selected,
                        replanned=replanned, next_tok=nxt, s_wps=s_wps, done=done, **plan_kw)
            env.sync()
        else:
            env.allocate_scored(kind, max_tasks, max_agents, edge_scores=scores, gate=gate, replan_interval=interval,
                                out={"selected": selected, "replanned": replanned}, **plan_kw)
            before = env.metrics()[:, S_WPS_COL]
            env.step_staged()
            after = env.metrics()[:, S_WPS_COL]
            _, term, trunc = env.step_result()
            env.tokens(kind, max_tasks, max_agents, out=nxt)
            env.sync()
            if ended is not None and ended.any():  # rows of envs whose episode had ended before this step: as the fused kernel reports them
                after = np.where(ended, before, after)
                keep = torch.from_numpy(~ended).to(dev)
                selected.mul_(keep.to(selected.dtype).view(-1, 1, 1)); replanned.mul_(keep.to(replanned.dtype))
            s_wps.copy_(torch.from_numpy(np.stack([before, after])))
            done.copy_(torch.from_numpy((term.astype(np.uint8) | (trunc.astype(np.uint8) << 1))))
            ended = term | trunc
        yield t, {"tok": tok, "scores": scores, "selected": selected, "replanned": replanned,
                  "step_reward": torch.div(s_wps[1] - s_wps[0], twenty), "next_tok": nxt, "done": done}


def rl_run_stream(env: BatchedMultiUAVEnv, seeds, policy, interval: int = 20, kind: str = "pair", max_tasks: int = 32, max_agents: int = 16,
                  gate: str = "trainer", device=None, max_steps: int = 0, max_launches: Optional[int] = None, **plan_kw):
    """run_rl_episode with the policy consulted only where the reference consults it — when `_should_replan` fires
    (experiments/train_pair_cost.py:139-145) — for every env of the batch at once (muavta_rl_run_device):

        for k, tr in rl_run_stream(env, seeds, policy): buffer.push(rows of tr where tr["replanned"] == 1)

    Every launch plans for the envs that are parked at a gate (scores = `policy(tok)` on the tokens of their parked state), takes that
    step, records the transition run_rl_episode pushes (`tok`, `scores`, `selected`, `step_reward`, `next_tok`, `done`: rows with
    `replanned` == 1; `next_tok` rows of the others are stale) and then steps each env with empty actions up to ITS next gate (at most
    `max_steps` steps per launch when > 0: an env cut short is not at a gate, `replanned` 0 at the next launch, and simply continues).
    Envs advance by different numbers of steps per launch (`n_stepped`); the stream ends when every episode has ended.  Also yielded:
    `park` (u8: bit 0 terminated, bit 1 truncated, bit 2 at a gate) and `reward_sum` (env rewards over the launch's steps).  All tensors
    are reused between yields (two alternating token buffer sets, as in `rl_stream`)."""
    import torch

    dev = torch.device("cuda", env.device_index if device is None else device)
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32, np.float64: torch.float64}
    shapes = env.token_shapes(kind, max_tasks, max_agents)
    N = env.n_envs
    mk = lambda: {name: torch.empty(shape, dtype=tdt[dtype], device=dev) for name, (shape, dtype) in shapes.items()}  # noqa: E731
    bufs, nxt = [mk(), mk()], mk()
    selected = torch.empty((N, max_agents, max_tasks), dtype=torch.float32, device=dev)
    replanned = torch.empty((N,), dtype=torch.int32, device=dev)
    s_wps = torch.empty((2, N), dtype=torch.float64, device=dev)
    done = torch.empty((N,), dtype=torch.uint8, device=dev)
    n_stepped = torch.empty((N,), dtype=torch.int32, device=dev)
    park = torch.empty((N,), dtype=torch.uint8, device=dev)
    reward_sum = torch.empty((N,), dtype=torch.float64, device=dev)
    twenty = torch.full((N,), 20.0, dtype=torch.float64, device=dev)
    env.reset(np.asarray(seeds, dtype=np.uint64))
    env.tokens(kind, max_tasks, max_agents, out=bufs[0])   # tok of the first gate (t = 0)
    k = 0
    while max_launches is None or k < max_launches:
        tok, prk = bufs[k & 1], bufs[(k + 1) & 1]
        env.sync()
        scores = policy(tok)
        if scores.dtype != torch.float32 or not scores.is_contiguous():
            scores = scores.to(torch.float32).contiguous()
        env.wait_stream(torch.cuda.current_stream(dev).cuda_stream)
        env.rl_run(kind, max_tasks, max_agents, edge_scores=scores, gate=gate, replan_interval=interval, selected=selected, replanned=replanned,
                   next_tok=nxt, s_wps=s_wps, done=done, park_tok=prk, n_stepped=n_stepped, park=park, reward_sum=reward_sum, max_steps=max_steps, **plan_kw)
        env.sync()
        yield k, {"tok": tok, "scores": scores, "selected": selected, "replanned": replanned, "step_reward": torch.div(s_wps[1] - s_wps[0], twenty),
                  "next_tok": nxt, "done": done, "n_stepped": n_stepped, "park": park, "reward_sum": reward_sum, "park_tok": prk}
        k += 1
        if bool(((park & 3) != 0).all()):
            return


def step_rewards(s_wps_prev: np.ndarray, s_wps_now: np.ndarray) -> np.ndarray:
    """RL step reward of run_rl_episode (train_pair_cost.py:146-148)."""
    return (np.asarray(s_wps_now) - np.asarray(s_wps_prev)) / 20.0
