"""``BatchedMultiUAVEnv`` — N independent mUAV_TA environments on one MI355X.

Thin numpy view over the C ABI (include/muavta.h): every method maps to one entry point, which maps
to the reference surface cited there (MultiUAVEnv.reset / step / observe / calculate_metrics and
HungarianAllocator.allocate_tasks).  No simulation logic lives in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import native
from .native import MuavtaError
from .params import METRIC_KEYS, N_METRICS, N_SCALARS, MuavtaDims, MuavtaParams, params_from_config

# MuavtaField enum (include/muavta.h)
F = {name: i for i, name in enumerate([
    "AGENT_POS", "AGENT_STATE", "AGENT_HEAD", "AGENT_QUEUE", "AGENT_NFT", "AGENT_NFP", "AGENT_CAPS", "AGENT_ATTACK_CAP",
    "AGENT_TYPE", "AGENT_NAME_IDX", "AGENT_DIST", "AGENT_MISC", "TASK_ID", "TASK_STATUS", "TASK_POS", "TASK_CUR",
    "TASK_ALLOC", "TASK_ORG_DONE", "TASK_META", "TASK_TIMES", "KNOWN", "THREAT_POS", "THREAT_META", "SCALARS",
    "OPEN_IDS", "EVENTS", "EVENT_LIST", "STAGED_ACTIONS", "ERROR", "RELEASE_LOG", "KNOWN_COUNT", "ESCORTS"])}


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _fp(a: np.ndarray):
    """pointer of an array the CALLER keeps alive in a local for the duration of the (synchronous) library call — a third of data_as's
    cost, which holds a reference for the pointer's lifetime; the per-step calls of the facade (get / observe / step / allocate) use it"""
    return C.c_void_p(a.ctypes.data)


class BatchedMultiUAVEnv:
    def __init__(self, config, n_envs: int, device: int = 0, flags=None, **tiles):
        self.params = config if isinstance(config, MuavtaParams) else params_from_config(config, flags, **tiles)
        self.L = native.lib()
        self.h = C.c_void_p()
        rc = self.L.muavta_create(C.byref(self.params), int(n_envs), int(device), C.byref(self.h))
        if rc != 0:
            msg = self.L.muavta_last_error(None).decode()
            self.h = C.c_void_p()
            raise MuavtaError(f"muavta_create failed ({rc}): {msg}")
        d = MuavtaDims()
        self._ck(self.L.muavta_dims(self.h, C.byref(d)))
        self.dims = d
        self.device_index = int(device)
        self.n_envs, self.n_agents = d.n_envs, d.n_agents
        self.T, self.H, self.Q, self.E, self.A_tile = d.tile_tasks, d.n_threats, d.queue_cap, d.event_cap, d.tile_agents
        self.max_tasks = d.max_tasks
        self.possible_agents = self.params.possible_agents
        self._alloc_mode = 0
        self.escalated = {}     # rollout(escalate=True): env index -> (handle of the larger tile, row)
        self._esc_rows = None

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc: int):
        if rc != 0:
            raise MuavtaError(f"muavta error {rc}: {self.L.muavta_last_error(self.h).decode()}")

    def close(self):
        for h in getattr(self, "_esc_handles", {}).values():
            h.close()
        self._esc_handles = {}
        if getattr(self, "h", None) and self.h.value:
            self.L.muavta_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ env surface
    def reset(self, seeds: Sequence[int]):
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        if s.shape != (self.n_envs,):
            raise ValueError(f"seeds must have shape ({self.n_envs},)")
        self._ck(self.L.muavta_reset(self.h, _vp(s)))

    def step(self, act_agent: np.ndarray, act_index: np.ndarray):
        """act_* : int32 [n_envs, cap]; act_agent -1 terminates an env's list.  cap is the tile's action_cap, or any
        length for list-valued actions (`muavta_step_lists`: applied in order inside the one step)."""
        aa = np.ascontiguousarray(act_agent, dtype=np.int32)
        ai = np.ascontiguousarray(act_index, dtype=np.int32)
        if aa.ndim != 2 or aa.shape != ai.shape or aa.shape[0] != self.n_envs or aa.shape[1] < 1:
            raise ValueError(f"actions must have shape ({self.n_envs}, cap), cap >= 1")
        if aa.shape[1] == self.A_tile:
            self._ck(self.L.muavta_step(self.h, _fp(aa), _fp(ai)))
        else:
            self._ck(self.L.muavta_step_lists(self.h, _vp(aa), _vp(ai), aa.shape[1]))

    def pack_actions(self, per_env):
        """[(agent_id, open_index), ...] per env -> the two padded arrays `step` takes (as wide as the longest list needs)."""
        cap = max([self.A_tile] + [len(acts) for acts in per_env])
        aa = np.full((self.n_envs, cap), -1, dtype=np.int32)
        ai = np.zeros((self.n_envs, cap), dtype=np.int32)
        for n, acts in enumerate(per_env):
            for k, (a, i) in enumerate(acts):
                aa[n, k], ai[n, k] = a, i
        return aa, ai

    def allocate(self, replan_interval: int = 20, use_visibility: bool = True, fetch: bool = True):
        if not fetch:
            self._ck(self.L.muavta_allocate(self.h, int(replan_interval), int(use_visibility), None, None))
            return None
        aa = np.empty((self.n_envs, self.A_tile), dtype=np.int32)
        ai = np.empty((self.n_envs, self.A_tile), dtype=np.int32)
        self._ck(self.L.muavta_allocate(self.h, int(replan_interval), int(use_visibility), _fp(aa), _fp(ai)))
        return aa, ai

    def set_allocator(self, name: str = "hungarian"):
        """'hungarian' (Local-/Coalition-Hungarian), 'urgency_pair' (UrgencyPair.plan under the WPS harness gate) or
        'urgency_coalition' (UrgencyCoalition.plan under the escort harness gate, with commit locks) or 'hungarian_gated'
        (the trainers' expert: allocate_tasks(force=True) under _should_replan(env, events, interval), train_pair_cost.py:33-43)."""
        self._alloc_mode = {"hungarian": 0, "urgency_pair": 1, "urgency_coalition": 2, "hungarian_gated": 3}[name]
        self._ck(self.L.muavta_set_allocator(self.h, self._alloc_mode))

    GATES = {"force": 0, "trainer": 1, "escort": 2, "allocator": 3}
    SC_EDGE_VALID_ONLY, SC_FULL_TASK_LIST, SC_COMMIT = 1, 2, 4

    def allocate_scored(self, kind: str = "pair", max_tasks: Optional[int] = None, max_agents: int = 16, *, edge_scores=None, task_pri=None,
                        reserved=None, gate: str = "trainer", replan_interval: int = 20, use_visibility: bool = True,
                        edge_valid_only: Optional[bool] = None, full_task_list: bool = False, commit: bool = False,
                        want_selected: bool = True, fetch: bool = True, out=None):
        """`HungarianAllocator.allocate_tasks(..., task_priorities=, reserved_agent_names=, edge_scores=)` with caller-computed
        inputs in the token layout of `tokens(kind, max_tasks, max_agents)` — what PairCostHybrid.plan / AttentionRAH.plan /
        AttentionEscort._plan_from_scores do with their network's output (include/muavta.h: muavta_allocate_scored).
        numpy inputs go through the host entry point; CUDA torch tensors (all of them, plus `out` = {'selected': f32 [N, MA, MT],
        'replanned': i32 [N]} tensors to fill) through the device one, on the handle's stream, without a copy or a sync.
        Returns {'selected', 'replanned'[, 'act_agent', 'act_index']}; the plan is staged for `step_staged()`."""
        k = self.TOKEN_KINDS[kind][0]
        mt = int(max_tasks if max_tasks is not None else (48 if kind == "escort" else 32))
        ma, N = int(max_agents), self.n_envs
        if edge_valid_only is None:
            edge_valid_only = kind != "escort"
        flags = (self.SC_EDGE_VALID_ONLY if edge_valid_only else 0) | (self.SC_FULL_TASK_LIST if full_task_list else 0) | (self.SC_COMMIT if commit else 0)
        spec = native.MuavtaScored(k, mt, ma, self.GATES[gate], flags, int(replan_interval), int(bool(use_visibility)), 0)
        shapes = {"edge_scores": ((N, ma, mt), 4), "task_pri": ((N, mt), 8), "reserved": ((N,), 8), "selected": ((N, ma, mt), 4), "replanned": ((N,), 4)}
        ins = {"edge_scores": edge_scores, "task_pri": task_pri, "reserved": reserved}
        on_device = any(v is not None and hasattr(v, "data_ptr") for v in ins.values()) or out is not None
        if on_device:
            tens = dict(ins)
            tens.update(out or {})
            for name, t in tens.items():
                if t is None:
                    continue
                shape, size = shapes[name]
                if tuple(t.shape) != shape or not t.is_cuda or not t.is_contiguous() or t.element_size() != size or t.device.index != self.device_index:
                    raise ValueError(f"allocate_scored: {name} must be a contiguous tensor of shape {shape} with {size}-byte elements on cuda:{self.device_index}")
                setattr(spec, name, t.data_ptr())
            self._ck(self.L.muavta_allocate_scored_device(self.h, C.byref(spec)))
            return out
        arrs = {}
        for name, dt in (("edge_scores", np.float32), ("task_pri", np.float64), ("reserved", np.uint64)):
            if ins[name] is not None:
                a = np.ascontiguousarray(ins[name], dtype=dt)
                if a.shape != shapes[name][0]:
                    raise ValueError(f"allocate_scored: {name} must have shape {shapes[name][0]}")
                arrs[name] = a
                setattr(spec, name, a.ctypes.data)
        res = {"replanned": np.empty(N, dtype=np.int32)}
        spec.replanned = res["replanned"].ctypes.data
        if want_selected:
            res["selected"] = np.empty((N, ma, mt), dtype=np.float32)
            spec.selected = res["selected"].ctypes.data
        aa = ai = None
        if fetch:
            aa = np.empty((N, self.A_tile), dtype=np.int32)
            ai = np.empty((N, self.A_tile), dtype=np.int32)
            res["act_agent"], res["act_index"] = aa, ai
        self._ck(self.L.muavta_allocate_scored(self.h, C.byref(spec), _vp(aa), _vp(ai)))
        return res

    def token_shapes(self, kind: str = "pair", max_tasks: Optional[int] = None, max_agents: int = 16):
        """name -> (shape, numpy dtype) of the token tensors `tokens(out=...)` / `rl_step(next_tok=...)` fill."""
        _, dt, da = self.TOKEN_KINDS[kind]
        mt = int(max_tasks if max_tasks is not None else (48 if kind == "escort" else 32))
        N, ma = self.n_envs, int(max_agents)
        return {"task_feats": ((N, mt, dt), np.float32), "task_mask": ((N, mt), np.uint8), "task_ids": ((N, mt), np.int32),
                "agent_feats": ((N, ma, da), np.float32), "agent_mask": ((N, ma), np.uint8), "agent_ids": ((N, ma), np.int32),
                "edge_valid": ((N, ma, mt), np.float32), "n_urgent": ((N,), np.int32)}

    def rl_step(self, kind: str = "pair", max_tasks: Optional[int] = None, max_agents: int = 16, *, edge_scores=None, task_pri=None, reserved=None,
                gate: str = "trainer", replan_interval: int = 20, use_visibility: bool = True, edge_valid_only: Optional[bool] = None,
                full_task_list: bool = False, commit: bool = False, selected=None, replanned=None, next_tok: Optional[dict] = None,
                s_wps=None, done=None, write_obs: bool = False, part: Optional[int] = None):
        """muavta_rl_step_device: one iteration of run_rl_episode's loop body (experiments/train_pair_cost.py:139-153) for every
        env in ONE launch — plan with the caller's scores -> step -> S_WPS before / after -> next tokens.  Every tensor is a
        contiguous CUDA torch tensor on the env's GPU: inputs as in `allocate_scored`; outputs `selected` f32 [N, MA, MT],
        `replanned` i32 [N], `next_tok` (dict with the shapes of `token_shapes`), `s_wps` f64 [2, N], `done` u8 [N] (any may be
        None).  Asynchronous on the handle's stream — or, with `part=p` after `set_parts(k)`, on sub-batch p's stream for its rows only
        (`wait_part(p)` before reading them): the network can work on one part's tokens while the device steps the other."""
        k = self.TOKEN_KINDS[kind][0]
        mt = int(max_tasks if max_tasks is not None else (48 if kind == "escort" else 32))
        ma, N = int(max_agents), self.n_envs
        if edge_valid_only is None:
            edge_valid_only = kind != "escort"
        flags = (self.SC_EDGE_VALID_ONLY if edge_valid_only else 0) | (self.SC_FULL_TASK_LIST if full_task_list else 0) | (self.SC_COMMIT if commit else 0)
        rs = native.MuavtaRlStep()
        rs.plan = native.MuavtaScored(k, mt, ma, self.GATES[gate], flags, int(replan_interval), int(bool(use_visibility)), 0)
        rs.write_obs = int(bool(write_obs))
        rs.part = 0 if part is None else int(part) + 1   # a sub-batch of set_parts() on its own stream; tensors stay whole-batch

        def put(obj, name, t, shape, size):
            if t is None:
                return
            if tuple(t.shape) != shape or not t.is_cuda or not t.is_contiguous() or t.element_size() != size or t.device.index != self.device_index:
                raise ValueError(f"rl_step: {name} must be a contiguous tensor of shape {shape} with {size}-byte elements on cuda:{self.device_index}")
            setattr(obj, name, t.data_ptr())

        put(rs.plan, "edge_scores", edge_scores, (N, ma, mt), 4); put(rs.plan, "task_pri", task_pri, (N, mt), 8); put(rs.plan, "reserved", reserved, (N,), 8)
        put(rs.plan, "selected", selected, (N, ma, mt), 4); put(rs.plan, "replanned", replanned, (N,), 4)
        put(rs, "s_wps", s_wps, (2, N), 8); put(rs, "done", done, (N,), 1)
        if next_tok is not None:
            for name, (shape, dtype) in self.token_shapes(kind, mt, ma).items():
                put(rs, name, next_tok[name], shape, np.dtype(dtype).itemsize)
        self._ck(self.L.muavta_rl_step_device(self.h, C.byref(rs)))

    def rl_run(self, kind: str = "pair", max_tasks: Optional[int] = None, max_agents: int = 16, *, edge_scores=None, task_pri=None, reserved=None,
               gate: str = "trainer", replan_interval: int = 20, use_visibility: bool = True, edge_valid_only: Optional[bool] = None,
               full_task_list: bool = False, commit: bool = False, selected=None, replanned=None, next_tok: Optional[dict] = None,
               s_wps=None, done=None, park_tok: Optional[dict] = None, n_stepped=None, park=None, reward_sum=None, max_steps: int = 0,
               write_obs: bool = False, part: Optional[int] = None):
        """muavta_rl_run_device: `rl_step` for the first step, then every env keeps stepping with empty actions until ITS gate fires
        again, its episode ends or `max_steps` steps were taken (0: no bound) — the reference's loops consult the policy only at a gate
        (experiments/train_pair_cost.py:139-145).  Extra outputs (CUDA tensors, any may be None): `park_tok` (dict like `next_tok`: the tokens
        of the state each env stopped in — what the policy sees next), `n_stepped` i32 [N], `park` u8 [N] (bit 0 terminated, bit 1
        truncated, bit 2 stopped at a gate), `reward_sum` f64 [N].  `next_tok` rows are written for envs that planned (replanned 1) only."""
        k = self.TOKEN_KINDS[kind][0]
        mt = int(max_tasks if max_tasks is not None else (48 if kind == "escort" else 32))
        ma, N = int(max_agents), self.n_envs
        if edge_valid_only is None:
            edge_valid_only = kind != "escort"
        flags = (self.SC_EDGE_VALID_ONLY if edge_valid_only else 0) | (self.SC_FULL_TASK_LIST if full_task_list else 0) | (self.SC_COMMIT if commit else 0)
        rr = native.MuavtaRlRun()
        rs = rr.first
        rs.plan = native.MuavtaScored(k, mt, ma, self.GATES[gate], flags, int(replan_interval), int(bool(use_visibility)), 0)
        rs.write_obs = int(bool(write_obs))
        rs.part = 0 if part is None else int(part) + 1
        rr.max_steps = int(max_steps)

        def put(obj, name, t, shape, size):
            if t is None:
                return
            if tuple(t.shape) != shape or not t.is_cuda or not t.is_contiguous() or t.element_size() != size or t.device.index != self.device_index:
                raise ValueError(f"rl_run: {name} must be a contiguous tensor of shape {shape} with {size}-byte elements on cuda:{self.device_index}")
            setattr(obj, name, t.data_ptr())

        put(rs.plan, "edge_scores", edge_scores, (N, ma, mt), 4); put(rs.plan, "task_pri", task_pri, (N, mt), 8); put(rs.plan, "reserved", reserved, (N,), 8)
        put(rs.plan, "selected", selected, (N, ma, mt), 4); put(rs.plan, "replanned", replanned, (N,), 4)
        put(rs, "s_wps", s_wps, (2, N), 8); put(rs, "done", done, (N,), 1)
        put(rr, "n_stepped", n_stepped, (N,), 4); put(rr, "park", park, (N,), 1); put(rr, "reward_sum", reward_sum, (N,), 8)
        for prefix, obj, toks in (("", rs, next_tok), ("park_", rr, park_tok)):
            if toks is not None:
                for name, (shape, dtype) in self.token_shapes(kind, mt, ma).items():
                    put(obj, prefix + name, toks[name], shape, np.dtype(dtype).itemsize)
        self._ck(self.L.muavta_rl_run_device(self.h, C.byref(rr)))

    def step_run(self, act_agent: Optional[np.ndarray] = None, act_index: Optional[np.ndarray] = None, gate: str = "trainer", replan_interval: int = 20,
                 max_steps: int = 0, write_obs: bool = True, fetch: bool = True):
        """muavta_step_run: `step(act_agent, act_index)` (both None: the plan `allocate(fetch=False)` staged) followed by empty-action
        steps until each env's gate fires, its episode ends or `max_steps` steps were taken — the run-ahead for a host-side planner
        (experiments/wps_eval.py:248-254,273).  Returns (n_stepped i32 [N], park u8 [N], reward_sum f64 [N]) unless fetch=False."""
        aa = ai = None
        if act_agent is not None:
            aa = np.ascontiguousarray(act_agent, dtype=np.int32)
            ai = np.ascontiguousarray(act_index, dtype=np.int32)
            if aa.shape != (self.n_envs, self.A_tile) or ai.shape != aa.shape:
                raise ValueError(f"actions must have shape ({self.n_envs}, {self.A_tile})")
        n = np.empty(self.n_envs, dtype=np.int32) if fetch else None
        pk = np.empty(self.n_envs, dtype=np.uint8) if fetch else None
        rs = np.empty(self.n_envs, dtype=np.float64) if fetch else None
        self._ck(self.L.muavta_step_run(self.h, _vp(aa), _vp(ai), self.GATES[gate], int(replan_interval), int(max_steps), int(bool(write_obs)), _vp(n), _vp(pk), _vp(rs)))
        return (n, pk, rs) if fetch else None

    TOKEN_KINDS = {"pair": (0, 13, 12), "pair_raw": (1, 9, 11), "escort": (2, 22, 16)}

    def tokens(self, kind: str = "pair", max_tasks: Optional[int] = None, max_agents: int = 16, out=None):
        """Batched token builders of the hybrids, from the device state: `build_pair_tokens` ('pair', 'pair_raw') /
        `build_escort_tokens` ('escort') of the reference for every env at once (PairCostHybrid.py:31-65,
        AttentionRAH.py:50-173, AttentionEscort.py:76-243).  Returns numpy arrays in the reference's layout, or — with
        `out` = dict of CUDA torch tensors of the right shapes/dtypes (float32 / uint8 / int32) — fills those in place
        on the handle's stream without a host copy.  Also returned: `expert_mask` (= `_expert_mask(tok, pairs)` of the plan
        the last `allocate` staged, experiments/train_pair_cost.py:54-71) and `replanned` (that allocate planned now)."""
        k, dt, da = self.TOKEN_KINDS[kind]
        mt = int(max_tasks if max_tasks is not None else (48 if kind == "escort" else 32))
        ma = int(max_agents)
        N = self.n_envs
        shapes = {"task_feats": ((N, mt, dt), np.float32), "task_mask": ((N, mt), np.uint8), "task_ids": ((N, mt), np.int32),
                  "agent_feats": ((N, ma, da), np.float32), "agent_mask": ((N, ma), np.uint8), "agent_ids": ((N, ma), np.int32),
                  "edge_valid": ((N, ma, mt), np.float32), "n_urgent": ((N,), np.int32),
                  "expert_mask": ((N, ma, mt), np.float32), "replanned": ((N,), np.int32)}
        if out is not None:
            ptrs = []
            for name, (shape, dtype) in shapes.items():
                t = out.get(name) if name in ("n_urgent", "expert_mask", "replanned") else out[name]  # (the three optional outputs)
                if t is None:
                    ptrs.append(None)
                    continue
                if tuple(t.shape) != shape or not t.is_cuda or not t.is_contiguous() or t.element_size() != np.dtype(dtype).itemsize:
                    raise ValueError(f"tokens(out=...): {name} must be a contiguous CUDA tensor of shape {shape}, {np.dtype(dtype).name}")
                ptrs.append(C.c_void_p(t.data_ptr()))
            self._ck(self.L.muavta_tokens_device(self.h, k, mt, ma, *ptrs))
            return out
        arrs = {name: np.empty(shape, dtype=dtype) for name, (shape, dtype) in shapes.items()}
        self._ck(self.L.muavta_tokens(self.h, k, mt, ma, *[_vp(a) for a in arrs.values()]))
        return arrs

    def context(self, kind: str = "pair", max_tasks: int = 32, out=None):
        """The ContextPair hybrids' context vector for every env (`build_context_summary`, ContextPairHybrid.py:33-78): f32 [N, 8]
        ('pair') or [N, 1] ('pair_raw').  `out`: a contiguous CUDA torch tensor to fill on the handle's stream (no sync)."""
        k = self.TOKEN_KINDS[kind][0]
        if k == 2:
            raise ValueError("the context vector is defined for the pair tokens ('pair', 'pair_raw')")
        shape = (self.n_envs, 1 if k == 1 else 8)
        if out is not None:
            if tuple(out.shape) != shape or not out.is_cuda or not out.is_contiguous() or out.element_size() != 4 or out.device.index != self.device_index:
                raise ValueError(f"context(out=...): a contiguous float32 tensor of shape {shape} on cuda:{self.device_index}")
            self._ck(self.L.muavta_context_device(self.h, k, int(max_tasks), C.c_void_p(out.data_ptr())))
            return out
        a = np.empty(shape, dtype=np.float32)
        self._ck(self.L.muavta_context(self.h, k, int(max_tasks), _vp(a)))
        return a

    OPS = {"uav_allocate": 0, "create_escort": 1, "sync_escorts": 2, "retire_escort": 3, "escort_fighters_near": 4,
           "action_valid": 5, "set_queue": 6}

    def call(self, op: str, iargs=(), darg: float = -1.0, env_index: int = 0) -> np.ndarray:
        """The reference's out-of-step mutators on one env (include/muavta.h: muavta_call): returns out[72]."""
        ia = np.zeros(8, dtype=np.int32)
        ia[: len(iargs)] = np.asarray(list(iargs), dtype=np.int32)
        out = np.zeros(72, dtype=np.int32)
        self._ck(self.L.muavta_call(self.h, int(env_index), self.OPS[op], _vp(ia), float(darg), _vp(out)))
        return out

    # ------------------------------------------------------------------ multi-GPU metric reduction over RCCL (C ABI)
    @staticmethod
    def comm_uid() -> bytes:
        """128-byte RCCL unique id (rank 0 creates it and ships it to the other ranks)."""
        buf = (C.c_uint8 * 128)()
        L = native.lib()
        rc = L.muavta_comm_uid(buf)
        if rc != 0:
            raise MuavtaError(f"muavta_comm_uid failed ({rc}): {L.muavta_last_error(None).decode()}")
        return bytes(buf)

    def comm_init(self, rank: int, n_ranks: int, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._ck(self.L.muavta_comm_init(self.h, int(rank), int(n_ranks), buf))

    def allreduce_metrics(self, f_partials: np.ndarray, counters: np.ndarray):
        f = np.ascontiguousarray(f_partials, dtype=np.float64)
        c = np.ascontiguousarray(counters, dtype=np.int64)
        fo, co = np.empty_like(f), np.empty_like(c)
        self._ck(self.L.muavta_allreduce_metrics(self.h, _vp(f), f.size, _vp(c), c.size, _vp(fo), _vp(co)))
        return fo, co

    def comm_destroy(self):
        self._ck(self.L.muavta_comm_destroy(self.h))

    def set_release_log(self, enable: bool = True):
        """Per-step log of released task slots (`get("RELEASE_LOG")`): id + knower mask; used by the facade."""
        self._ck(self.L.muavta_set_release_log(self.h, int(bool(enable))))

    def step_staged(self):
        self._ck(self.L.muavta_step_staged(self.h))

    def rollout(self, seeds: Optional[Sequence[int]], n_steps: int = 150, replan_interval: int = 20,
                use_visibility: bool = True, write_obs: bool = True, escalate: bool = False):
        """muavta_rollout.  `escalate=True` (needs seeds): an env that overflowed its tile (`ERROR != 0` — the reference's task list,
        queues and event lists are unbounded Python lists, DroneEnv.py:325-328,1894-1895; the device's tiles are not) is re-run
        from its seed in a handle of the next larger tile (16x40 -> 24x48 -> 64x128, same configuration) on the same GPU, and
        `rollout_metrics()` returns its row from there; `escalated` maps env index -> (handle, row in that handle) for callers
        that want the final state of such an env (`handle.get(...)`: the task-slot axis has the larger tile's width).  The
        capacity flag of the small tile stays readable through `get("ERROR")`.  Blocks until the batch is complete."""
        s = None
        if seeds is not None:
            s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
            if s.shape != (self.n_envs,):
                raise ValueError(f"seeds must have shape ({self.n_envs},)")
        self.escalated = {}
        self._esc_rows = None
        self._ck(self.L.muavta_rollout(self.h, _vp(s), int(n_steps), int(replan_interval), int(use_visibility), int(write_obs)))
        if escalate:
            if s is None:
                raise ValueError("rollout(escalate=True) re-runs flagged envs from their seeds: pass seeds")
            self._escalate(s, int(n_steps), int(replan_interval), bool(use_visibility), bool(write_obs))

    TILE_LADDER = ((16, 40, 16), (24, 48, 24), (64, 128, 48))

    def _escalate(self, seeds, n_steps, interval, use_vis, write_obs):
        idx = np.nonzero(self.get("ERROR"))[0]
        if not len(idx):
            return
        rows = {}
        rung = next((i for i, (a, t, h) in enumerate(self.TILE_LADDER) if self.A_tile <= a and self.T <= t), len(self.TILE_LADDER))
        if not hasattr(self, "_esc_handles"):
            self._esc_handles = {}
        while len(idx):
            rung += 1
            if rung >= len(self.TILE_LADDER):
                raise MuavtaError(f"envs {idx[:8].tolist()} overflow the largest tile (64 agents x 128 task slots): no tile left to escalate to")
            a, t, hh = self.TILE_LADDER[rung]
            h = self._esc_handles.get(rung)
            if h is None or h.n_envs < len(idx):
                if h is not None:
                    h.close()
                p = MuavtaParams.from_buffer_copy(self.params)
                p.tile_agents, p.tile_tasks, p.tile_threats = a, t, max(hh, self.H)
                h = BatchedMultiUAVEnv(p, max(len(idx), 64), device=self.device_index)
                self._esc_handles[rung] = h
            h._ck(h.L.muavta_set_allocator(h.h, self._alloc_mode))
            s2 = np.full(h.n_envs, seeds[idx[0]], dtype=np.uint64)
            s2[:len(idx)] = seeds[idx]
            h.rollout(s2, n_steps, interval, use_vis, write_obs)
            m = h.rollout_metrics()
            bad = h.get("ERROR")[:len(idx)] != 0
            for k, i in enumerate(idx):
                if not bad[k]:
                    rows[int(i)] = m[k].copy()
                    self.escalated[int(i)] = (h, k)
            idx = idx[bad]
        self._esc_rows = rows

    def record_shapes(self, kind: str, n_steps: int, max_tasks: int, max_agents: int):
        """name -> (shape, numpy dtype) of the rings `rollout_record` fills."""
        _, dt, da = self.TOKEN_KINDS[kind]
        K, N, mt, ma = int(n_steps), self.n_envs, int(max_tasks), int(max_agents)
        return {"task_feats": ((K, N, mt, dt), np.float32), "task_mask": ((K, N, mt), np.uint8), "task_ids": ((K, N, mt), np.int32),
                "agent_feats": ((K, N, ma, da), np.float32), "agent_mask": ((K, N, ma), np.uint8), "agent_ids": ((K, N, ma), np.int32),
                "edge_valid": ((K, N, ma, mt), np.float32), "n_urgent": ((K, N), np.int32), "expert_mask": ((K, N, ma, mt), np.float32),
                "replanned": ((K, N), np.int32), "s_wps": ((K + 1, N), np.float64)}

    def obs_ring_shapes(self, n_steps: int):
        """name -> (shape, numpy dtype) of the per-step observation rings `rollout_record(obs_rings=...)` fills
        (muavta_observe's device layouts with a leading [n_steps] axis)."""
        K, N, A, MT = int(n_steps), self.n_envs, self.n_agents, self.max_tasks
        return {"obs_tasks": ((K, N, 21, MT), np.float32), "obs_legal": ((K, N, A, self.dims.legal_words), np.uint64), "obs_pad": ((K, N, MT), np.uint8),
                "obs_agents": ((K, N, A, 9), np.float32), "obs_flags": ((K, N, 5), np.float32), "obs_reward": ((K, N), np.float64),
                "obs_done": ((K, N), np.uint8)}

    OBS_UNWRITTEN = 0x80  # include/muavta.h: MUAVTA_OBS_UNWRITTEN

    def rollout_record(self, seeds: Optional[Sequence[int]], n_steps: int, replan_interval: int, use_visibility: bool, rings: Optional[dict] = None,
                       kind: str = "pair", max_tasks: int = 32, max_agents: int = 16, write_obs: bool = False, obs_rings: Optional[dict] = None):
        """muavta_rollout_record: the fused rollout that also fills `rings` (dict of contiguous CUDA torch tensors with the
        shapes of `record_shapes`) with the per-step training data and/or `obs_rings` (shapes of `obs_ring_shapes`; implies
        write_obs) with every step's observation — no host hop, one launch for the whole episode batch.
        Asynchronous on the handle's stream: call `sync()` before reading the rings from another stream."""
        from .params import MuavtaRecord

        if rings is None and obs_rings is None:
            raise ValueError("rollout_record: pass rings and/or obs_rings")
        rec = MuavtaRecord()
        rec.kind = -1
        want = {}
        if rings is not None:
            k, _, _ = self.TOKEN_KINDS[kind]
            rec.kind, rec.max_tasks, rec.max_agents = k, int(max_tasks), int(max_agents)
            want.update({n: (rings, v) for n, v in self.record_shapes(kind, n_steps, max_tasks, max_agents).items()})
        if obs_rings is not None:
            write_obs = True
            want.update({n: (obs_rings, v) for n, v in self.obs_ring_shapes(n_steps).items()})
        for name, (src, (shape, dtype)) in want.items():
            t = src[name]
            if tuple(t.shape) != shape or not t.is_cuda or not t.is_contiguous() or t.element_size() != np.dtype(dtype).itemsize:
                raise ValueError(f"rollout_record: {name} must be a contiguous CUDA tensor of shape {shape}, {np.dtype(dtype).name}")
            if t.device.index != self.device_index:
                raise ValueError(f"rollout_record: {name} lives on {t.device}, the env batch on cuda:{self.device_index}")
            setattr(rec, name, t.data_ptr())
        s = None
        if seeds is not None:
            s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
            if s.shape != (self.n_envs,):
                raise ValueError(f"seeds must have shape ({self.n_envs},)")
        self._ck(self.L.muavta_rollout_record(self.h, _vp(s), int(n_steps), int(replan_interval), int(use_visibility), int(write_obs), C.byref(rec)))

    # ------------------------------------------------------------------ sub-batches on their own streams
    def set_parts(self, n_parts: int):
        """Split the env range into `n_parts` contiguous parts with a stream each (0 / 1: off): the *_part calls step one part
        asynchronously, so the host can decide for one part while the device steps another (include/muavta.h)."""
        self._ck(self.L.muavta_set_parts(self.h, int(n_parts)))
        self.n_parts = 0 if n_parts <= 1 else int(n_parts)

    def part_range(self, part: int):
        f, c = C.c_int32(), C.c_int32()
        self._ck(self.L.muavta_part_range(self.h, int(part), C.byref(f), C.byref(c)))
        return int(f.value), int(c.value)

    def rollout_part(self, part: int, n_steps: int = 1, replan_interval: int = 20, use_visibility: bool = True, write_obs: bool = True):
        self._ck(self.L.muavta_rollout_part(self.h, int(part), int(n_steps), int(replan_interval), int(use_visibility), int(write_obs)))

    def allocate_part(self, part: int, replan_interval: int = 20, use_visibility: bool = True, fetch: bool = True):
        if not fetch:
            self._ck(self.L.muavta_allocate_part(self.h, int(part), int(replan_interval), int(use_visibility), None, None))
            return None
        _, cnt = self.part_range(part)
        aa = np.empty((cnt, self.A_tile), dtype=np.int32)
        ai = np.empty((cnt, self.A_tile), dtype=np.int32)
        self._ck(self.L.muavta_allocate_part(self.h, int(part), int(replan_interval), int(use_visibility), _vp(aa), _vp(ai)))
        return aa, ai

    def step_part(self, part: int, act_agent: Optional[np.ndarray] = None, act_index: Optional[np.ndarray] = None):
        """act_*: int32 [count of the part, action_cap]; both None: the actions `allocate_part` staged on the device."""
        if act_agent is None:
            self._ck(self.L.muavta_step_part(self.h, int(part), None, None))
            return
        _, cnt = self.part_range(part)
        aa = np.ascontiguousarray(act_agent, dtype=np.int32)
        ai = np.ascontiguousarray(act_index, dtype=np.int32)
        if aa.shape != (cnt, self.A_tile) or ai.shape != (cnt, self.A_tile):
            raise ValueError(f"actions of part {part} must have shape {(cnt, self.A_tile)}")
        self._ck(self.L.muavta_step_part(self.h, int(part), _vp(aa), _vp(ai)))

    def observe_part(self, part: int):
        """The part's rows of `observe()` plus (reward, terminated, truncated); waits for that part's stream only."""
        _, n = self.part_range(part)
        A, MT = self.n_agents, self.max_tasks
        tasks = np.empty((n, 21, MT), dtype=np.float32)
        legal = np.empty((n, A, self.dims.legal_words), dtype=np.uint64)
        pad = np.empty((n, MT), dtype=np.uint8)
        agents = np.empty((n, A, 9), dtype=np.float32)
        flags = np.empty((n, 5), dtype=np.float32)
        r = np.empty(n, dtype=np.float64)
        d = np.empty(n, dtype=np.uint8)
        self._ck(self.L.muavta_observe_part(self.h, int(part), _vp(tasks), _vp(legal), _vp(pad), _vp(agents), _vp(flags), _vp(r), _vp(d)))
        tasks = np.ascontiguousarray(tasks.transpose(0, 2, 1))
        bits = (legal[..., :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)
        legal = bits.reshape(n, A, -1)[:, :, :MT]
        return ({"tasks": tasks, "legal_mask": legal.astype(bool), "mask": pad.astype(bool), "agents": agents, "event_flags": flags},
                r, (d & 1).astype(bool), (d & 2).astype(bool))

    def wait_part(self, part: int = -1):
        self._ck(self.L.muavta_wait_part(self.h, int(part)))

    def sync(self):
        self._ck(self.L.muavta_sync(self.h))

    def wait_stream(self, stream_handle: Optional[int] = None):
        """Order the handle's stream after `stream_handle` (a hipStream_t as an integer, e.g. `torch.cuda.current_stream().cuda_stream`;
        None = torch's current stream on this device): call it before handing the library tensors that stream may still be using."""
        if stream_handle is None:
            import torch

            stream_handle = torch.cuda.current_stream(self.device_index).cuda_stream
        self._ck(self.L.muavta_wait_stream(self.h, C.c_void_p(int(stream_handle))))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._ck(self.L.muavta_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def kernel_ms_history(self, n: int) -> np.ndarray:
        """Durations (ms) of the last n rollout launches, oldest first (n <= 64); waits for the newest."""
        out = np.empty(int(n), dtype=np.float32)
        self._ck(self.L.muavta_kernel_ms_history(self.h, _vp(out), int(n)))
        return out

    def launch_gaps_ms(self, n: int) -> np.ndarray:
        """Idle time (ms) of the handle's stream between the last n rollout launches: n - 1 gaps, oldest first."""
        out = np.empty(int(n) - 1, dtype=np.float32)
        self._ck(self.L.muavta_launch_gaps_ms(self.h, _vp(out), int(n)))
        return out

    def last_seed_ms(self) -> float:
        ms = C.c_float()
        self._ck(self.L.muavta_last_seed_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def observe(self):
        N, A, MT = self.n_envs, self.n_agents, self.max_tasks
        tasks = np.empty((N, 21, MT), dtype=np.float32)  # feature-major on the device
        legal = np.empty((N, A, self.dims.legal_words), dtype=np.uint64)  # bit rows on the device
        pad = np.empty((N, MT), dtype=np.uint8)
        agents = np.empty((N, A, 9), dtype=np.float32)
        flags = np.empty((N, 5), dtype=np.float32)
        self._ck(self.L.muavta_observe(self.h, _fp(tasks), _fp(legal), _fp(pad), _fp(agents), _fp(flags)))
        tasks = np.ascontiguousarray(tasks.transpose(0, 2, 1))    # -> [N, max_tasks, 21] like the reference's rows
        bits = (legal[..., :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)
        legal = bits.reshape(N, A, -1)[:, :, :MT]
        return {"tasks": tasks, "legal_mask": legal.astype(bool), "mask": pad.astype(bool), "agents": agents, "event_flags": flags}

    def step_result(self):
        r = np.empty(self.n_envs, dtype=np.float64)
        d = np.empty(self.n_envs, dtype=np.uint8)
        self._ck(self.L.muavta_step_result(self.h, _fp(r), _fp(d)))
        return r, (d & 1).astype(bool), (d & 2).astype(bool)

    def metrics(self) -> np.ndarray:
        m = np.empty((self.n_envs, N_METRICS), dtype=np.float64)
        self._ck(self.L.muavta_metrics(self.h, _vp(m)))
        return m

    def set_slot_cap(self, cap: int = 0):
        """Test hook (muavta_set_slot_cap): an env may use at most `cap` of its tile's task slots (0: all) — overflow raises the capacity
        flag exactly as a full tile does."""
        self._ck(self.L.muavta_set_slot_cap(self.h, int(cap)))
        self.slot_cap = int(cap)

    def set_lanes(self, lanes: int = 0):
        """State lanes of the handle (include/muavta.h): 0 = a second lane is created when a seeded rollout is issued while the previous one
        still runs (default), 1 = one lane only, 2 = create it now and always alternate.  With two lanes `rollout(seeds[i + 1])` can be
        queued before batch i is read: `rollout_metrics(back=1)` / `error_flags(back=1)` reach the batch before the latest one."""
        self._ck(self.L.muavta_set_lanes(self.h, int(lanes)))

    def lanes(self):
        """(mode, lanes allocated)"""
        m, a = C.c_int32(), C.c_int32()
        self._ck(self.L.muavta_lanes(self.h, C.byref(m), C.byref(a)))
        return int(m.value), int(a.value)

    def error_flags(self, back: int = 0) -> np.ndarray:
        """`get("ERROR")` of the latest seeded batch (back=0) or of the one before it on the other lane (back=1)."""
        out = np.empty(self.n_envs, dtype=np.int32)
        self._ck(self.L.muavta_error_flags_back(self.h, int(back), _vp(out)))
        return out

    def rollout_metrics(self, back: int = 0) -> np.ndarray:
        m = np.empty((self.n_envs, N_METRICS), dtype=np.float64)
        if back:
            self._ck(self.L.muavta_rollout_metrics_back(self.h, int(back), _vp(m)))
            return m
        self._ck(self.L.muavta_rollout_metrics(self.h, _vp(m)))
        if self._esc_rows:  # rollout(escalate=True): rows of the envs that were re-run on a larger tile
            for i, row in self._esc_rows.items():
                m[i] = row
        return m

    def metrics_dicts(self):
        return [dict(zip(METRIC_KEYS, row)) for row in self.metrics()]

    def refresh_observation(self):
        self._ck(self.L.muavta_refresh_observation(self.h))

    # ------------------------------------------------------------------ state access
    def _shape(self, name):
        shapes = self.__dict__.get("_shapes")
        if shapes is not None:
            return shapes[name]
        N, A, T, H, Q, E = self.n_envs, self.n_agents, self.T, self.H, self.Q, self.E
        f64, i32, u32 = np.float64, np.int32, np.uint32
        self._shapes = {
            "AGENT_POS": ((N, A, 2), f64), "AGENT_STATE": ((N, A), i32), "AGENT_HEAD": ((N, A), i32),
            "AGENT_QUEUE": ((N, A, Q), i32), "AGENT_NFT": ((N, A), f64), "AGENT_NFP": ((N, A, 2), f64),
            "AGENT_CAPS": ((N, A, 6), f64), "AGENT_ATTACK_CAP": ((N, A), i32), "AGENT_TYPE": ((N, A), i32),
            "AGENT_NAME_IDX": ((N, A), i32), "AGENT_DIST": ((N, A), f64), "AGENT_MISC": ((N, A, 6), i32),
            "TASK_ID": ((N, T), i32), "TASK_STATUS": ((N, T), i32), "TASK_POS": ((N, T, 2), f64),
            "TASK_CUR": ((N, T, 6), f64), "TASK_ALLOC": ((N, T, 6), f64), "TASK_ORG_DONE": ((N, T, 2), f64),
            "TASK_META": ((N, T, 8), i32), "TASK_TIMES": ((N, T, 2), f64), "KNOWN": ((N, A, self.dims.known_words), u32),
            "THREAT_POS": ((N, H, 2), f64), "THREAT_META": ((N, H, 8), i32), "SCALARS": ((N, N_SCALARS), f64),
            "OPEN_IDS": ((N, T), i32), "EVENTS": ((N, E, 2), i32), "EVENT_LIST": ((N, E, 2), i32),
            "STAGED_ACTIONS": ((N, self.A_tile, 3), i32), "ERROR": ((N,), i32),
            "RELEASE_LOG": ((N, 1 + 29 * T), f64), "KNOWN_COUNT": ((N, A), i32), "ESCORTS": ((N, self.A_tile, 2), i32),
        }
        return self._shapes[name]

    def get(self, name: str) -> np.ndarray:
        shape, dt = self._shape(name)
        out = np.empty(shape, dtype=dt)
        self._ck(self.L.muavta_get(self.h, F[name], _fp(out), out.nbytes))
        return out

    def set(self, name: str, value: np.ndarray):
        shape, dt = self._shape(name)
        v = np.ascontiguousarray(value, dtype=dt)
        if v.shape != shape:
            raise ValueError(f"{name} must have shape {shape}")
        self._ck(self.L.muavta_set(self.h, F[name], _vp(v), v.nbytes))

    def get_state(self) -> np.ndarray:
        buf = np.empty(self.n_envs * self.dims.state_bytes, dtype=np.uint8)
        self._ck(self.L.muavta_get_state(self.h, _vp(buf), buf.nbytes))
        return buf

    def set_state(self, buf: np.ndarray):
        b = np.ascontiguousarray(buf, dtype=np.uint8)
        self._ck(self.L.muavta_set_state(self.h, _vp(b), b.nbytes))

    def get_rng(self) -> np.ndarray:
        buf = np.empty((self.n_envs, 4, 2, 624), dtype=np.uint32)
        self._ck(self.L.muavta_get_rng(self.h, _vp(buf), buf.nbytes))
        return buf

    def set_rng(self, buf: np.ndarray):
        b = np.ascontiguousarray(buf, dtype=np.uint32)
        self._ck(self.L.muavta_set_rng(self.h, _vp(b), b.nbytes))

    def device_ptrs(self):
        ptrs = [C.c_void_p() for _ in range(6)]
        self._ck(self.L.muavta_device_ptrs(self.h, *[C.byref(p) for p in ptrs]))
        keys = ["state", "obs_tasks", "obs_legal", "obs_agents", "metrics", "stream"]
        return {k: p.value for k, p in zip(keys, ptrs)}


def lsap(cost: np.ndarray, device: int = 0, impl: str = "auto"):
    """Batched scipy-compatible linear_sum_assignment on the GPU: cost [n, nr, nc] (or [nr, nc]).
    impl: 'auto' | 'lds' (64 x 128 solver) | 'registers' (the allocator path's solver, up to 32 x 64)."""
    c = np.ascontiguousarray(cost, dtype=np.float64)
    single = c.ndim == 2
    if single:
        c = c[None]
    n, nr, nc = c.shape
    m = min(nr, nc)
    row = np.empty((n, m), dtype=np.int64)
    col = np.empty((n, m), dtype=np.int64)
    L = native.lib()
    rc = L.muavta_lsap_impl(int(device), _vp(c), n, nr, nc, _vp(row), _vp(col), {"auto": 0, "lds": 1, "registers": 2}[impl])
    if rc != 0:
        raise MuavtaError(f"muavta_lsap failed ({rc}): {L.muavta_last_error(None).decode()}")
    return (row[0], col[0]) if single else (row, col)


def avoid_obstacles(agent_pos, obstacles, movement, device: int = 0):
    """core_sim.SimCore.avoid_obstacles for n (position, movement) pairs (core_sim/src/sim_core.rs:25-59)."""
    p = np.ascontiguousarray(np.atleast_2d(agent_pos), dtype=np.float64)
    m = np.ascontiguousarray(np.atleast_2d(movement), dtype=np.float64)
    o = np.ascontiguousarray(np.asarray(obstacles, dtype=np.float64).reshape(-1, 3))
    out = np.empty_like(p)
    L = native.lib()
    rc = L.muavta_avoid_obstacles(int(device), _vp(p), _vp(m), p.shape[0], _vp(o) if len(o) else None, len(o), _vp(out))
    if rc != 0:
        raise MuavtaError(f"muavta_avoid_obstacles failed ({rc}): {L.muavta_last_error(None).decode()}")
    return out


def domain_log(x, device: int = 0):
    """The obstacle path's natural logarithm as the device computes it (diagnostic, `muavta_domain_log`)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.empty_like(x)
    L = native.lib()
    rc = L.muavta_domain_log(int(device), _vp(x), x.shape[0], _vp(out))
    if rc != 0:
        raise MuavtaError(f"muavta_domain_log failed ({rc}): {L.muavta_last_error(None).decode()}")
    return out


def domain_atan2(y, x, device: int = 0):
    """The obstacle rule's two-argument arctangent as the device computes it (diagnostic, `muavta_domain_atan2`)."""
    y = np.ascontiguousarray(y, dtype=np.float64).ravel()
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    if y.shape != x.shape:
        raise ValueError("domain_atan2: y and x differ in length")
    out = np.empty_like(x)
    L = native.lib()
    rc = L.muavta_domain_atan2(int(device), _vp(y), _vp(x), x.shape[0], _vp(out))
    if rc != 0:
        raise MuavtaError(f"muavta_domain_atan2 failed ({rc}): {L.muavta_last_error(None).decode()}")
    return out


def domain_math(x, y, device: int = 0):
    """The kernels' range-restricted sqrt / division on the device (diagnostic, `muavta_domain_math`):
    returns (sqrt(x), x / y, -x / y) as the device computes them."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    y = np.ascontiguousarray(y, dtype=np.float64).ravel()
    if x.shape != y.shape:
        raise ValueError("x and y must have the same length")
    outs = [np.empty_like(x) for _ in range(3)]
    L = native.lib()
    rc = L.muavta_domain_math(int(device), _vp(x), _vp(y), x.shape[0], *[_vp(o) for o in outs])
    if rc != 0:
        raise MuavtaError(f"muavta_domain_math failed ({rc}): {L.muavta_last_error(None).decode()}")
    return tuple(outs)
