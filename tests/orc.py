"""ctypes wrapper around oracle/liboracle.so — test infrastructure (the checker, never the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("MUAVTA_ORACLE_SO") or os.path.join(ORACLE_DIR, "liboracle.so")  # (override: the sanitizer build, `make -C oracle asan`)
        src = os.path.join(ORACLE_DIR, "muavta_oracle.cpp")
        if "MUAVTA_ORACLE_SO" not in os.environ and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(so)
        L.orc_create.restype = C.c_void_p
        L.orc_rng_new.restype = C.c_void_p
        L.orc_rng_random.restype = C.c_double
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_randint.restype = C.c_int64
        L.orc_rng_randbelow.restype = C.c_uint64
        L.orc_norm2.restype = C.c_double
        L.orc_np_sum.restype = C.c_double
        L.orc_get_lsap.restype = C.c_int64
        L.orc_run_quiet.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class OracleEnv:
    QCAP = 16

    def __init__(self, params):
        self.L = lib()
        self.params = params
        self.h = C.c_void_p(self.L.orc_create(C.byref(params)))
        self.A = params.n_agents

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def reset(self, seed):
        self.L.orc_reset(self.h, C.c_uint64(seed))

    def dims(self):
        d = np.zeros(16, dtype=np.int32)
        self.L.orc_dims(self.h, _p(d))
        keys = ["n_agents", "n_task_ids", "n_threats", "max_tasks", "n_open", "n_events", "n_actions", "n_lsap",
                "n_replans", "time_steps", "n_pending", "n_reached", "pending_reset", "terminated", "truncated", "max_queue"]
        return dict(zip(keys, d.tolist()))

    def allocate(self, interval, use_vis=1):
        aa = np.full(self.A + 1, -1, dtype=np.int32)
        ai = np.zeros(self.A + 1, dtype=np.int32)
        n = self.L.orc_allocate(self.h, int(interval), int(use_vis), _p(aa), _p(ai), self.A)
        return aa[:n].copy(), ai[:n].copy()

    def allocate_mode(self, interval, use_vis, mode):
        aa = np.full(self.A + 1, -1, dtype=np.int32)
        ai = np.zeros(self.A + 1, dtype=np.int32)
        n = self.L.orc_allocate_mode(self.h, int(interval), int(use_vis), int(mode), _p(aa), _p(ai), self.A)
        return aa[:n].copy(), ai[:n].copy()

    def allocate_scored(self, interval, use_vis, gate, kind, max_tasks, max_agents, flags=0, scores=None, pri=None, reserved=0):
        """HungarianAllocator.allocate_tasks with caller-supplied edge scores / priorities / reserved agents in token layout.
        Returns (act_agent, act_index, selected [max_agents, max_tasks])."""
        aa = np.full(self.A + 1, -1, dtype=np.int32)
        ai = np.zeros(self.A + 1, dtype=np.int32)
        sel = np.zeros((max_agents, max_tasks), np.float32)
        sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float32)
        pr = None if pri is None else np.ascontiguousarray(pri, dtype=np.float64)
        assert sc is None or sc.shape == (max_agents, max_tasks)
        assert pr is None or pr.shape == (max_tasks,)
        n = self.L.orc_allocate_scored(self.h, int(interval), int(use_vis), int(gate), int(kind), int(max_tasks), int(max_agents), int(flags),
                                       None if sc is None else _p(sc), None if pr is None else _p(pr), C.c_uint64(int(reserved)),
                                       _p(aa), _p(ai), self.A, _p(sel))
        return aa[:n].copy(), ai[:n].copy(), sel

    def gate(self, gate, interval):
        """The callers' replan gate (MUAVTA_GATE_*) on the current state."""
        return bool(self.L.orc_gate(self.h, int(gate), int(interval)))

    def run_quiet(self, gate, interval, max_steps=0, already=0, reward0=0.0):
        """env.step({}) until the gate fires / the episode ends / max_steps (counting `already`): (steps taken, at_gate, reward0 + the
        rewards of these steps, added in step order)."""
        ag = C.c_int(0)
        rs = C.c_double(float(reward0))
        n = self.L.orc_run_quiet(self.h, int(gate), int(interval), int(max_steps), int(already), C.byref(ag), C.byref(rs))
        return int(n), bool(ag.value), float(rs.value)

    def rl_run(self, interval, use_vis, gate, kind, max_tasks, max_agents, flags=0, scores=None, pri=None, reserved=0, max_steps=0):
        """One launch of muavta_rl_run_device for this env, composed from the oracle's own pieces: the first step of run_rl_episode's loop body
        (plan with the caller's scores under the gate -> step -> S_WPS before / after -> next tokens, experiments/train_pair_cost.py:139-152)
        and then the quiet stretch up to the next gate.  Returns a dict; an env whose episode had ended is left alone."""
        d = self.dims()
        out = {"n_stepped": 0, "replanned": False, "selected": np.zeros((max_agents, max_tasks), np.float32), "reward_sum": 0.0}
        if d["terminated"] or d["truncated"]:
            s = self.metrics()[4]
            out.update(s_before=s, s_after=s, done=int(d["terminated"]) | (int(d["truncated"]) << 1), at_gate=False)
        else:
            t0 = d["time_steps"]
            out["s_before"] = self.metrics()[4]
            oa, oi, sel = self.allocate_scored(interval, use_vis, gate, kind, max_tasks, max_agents, flags, scores=scores, pri=pri, reserved=reserved)
            out["replanned"] = self.scalars_last_plan() == t0
            out["selected"] = sel
            self.step(oa, oi)
            out["reward_sum"] = float(self.scalars()[1])
            out["s_after"] = self.metrics()[4]
            d = self.dims()
            out["done"] = int(d["terminated"]) | (int(d["truncated"]) << 1)
            if out["replanned"]:
                out["next_tok"] = self.tokens(kind, max_tasks, max_agents)
            n, ag, rs = self.run_quiet(gate, interval, max_steps, 1, out["reward_sum"])  # (same order of additions as the device: first step, then the quiet ones)
            out["n_stepped"] = 1 + n
            out["at_gate"] = ag
            out["reward_sum"] = rs
        d = self.dims()
        out["park"] = int(d["terminated"]) | (int(d["truncated"]) << 1) | (4 if out["at_gate"] else 0)
        out["park_tok"] = self.tokens(kind, max_tasks, max_agents)
        return out

    def rollout_mode(self, seed, n_steps, interval, use_vis, mode):
        return self.L.orc_rollout_mode(self.h, C.c_uint64(seed), int(n_steps), int(interval), int(use_vis), int(mode))

    def step(self, act_agent, act_index):
        aa = np.ascontiguousarray(act_agent, dtype=np.int32)
        ai = np.ascontiguousarray(act_index, dtype=np.int32)
        return self.L.orc_step(self.h, len(aa), _p(aa), _p(ai))

    def rollout(self, seed, n_steps, interval, use_vis=1, do_reset=1):
        return self.L.orc_rollout(self.h, C.c_uint64(seed), int(do_reset), int(n_steps), int(interval), int(use_vis))

    def metrics(self):
        m = np.zeros(30)
        self.L.orc_metrics(self.h, _p(m))
        return m

    def agents(self):
        rows = np.zeros((self.A, 16)); caps = np.zeros((self.A, 6)); q = np.zeros((self.A, self.QCAP), dtype=np.int32)
        self.L.orc_get_agents(self.h, _p(rows), _p(caps), _p(q), self.QCAP)
        return rows, caps, q

    def tokens(self, kind, max_tasks, max_agents):
        """kind 0 pair tokens, 1 raw pair tokens, 2 escort tokens -> dict of arrays (reference layout)."""
        dt, da = {0: (13, 12), 1: (9, 11), 2: (22, 16)}[kind]
        out = {"task_feats": np.zeros((max_tasks, dt), np.float32), "task_mask": np.ones(max_tasks, np.uint8),
               "task_ids": np.full(max_tasks, -1, np.int32), "agent_feats": np.zeros((max_agents, da), np.float32),
               "agent_mask": np.ones(max_agents, np.uint8), "agent_ids": np.full(max_agents, -1, np.int32),
               "edge_valid": np.zeros((max_agents, max_tasks), np.float32)}
        nu = np.zeros(1, np.int32)
        out["expert_mask"] = np.zeros((max_agents, max_tasks), np.float32)
        self.L.orc_tokens_expert(self.h, int(kind), int(max_tasks), int(max_agents), _p(out["task_feats"]), _p(out["task_mask"]),
                                 _p(out["task_ids"]), _p(out["agent_feats"]), _p(out["agent_mask"]), _p(out["agent_ids"]),
                                 _p(out["edge_valid"]), _p(nu), _p(out["expert_mask"]))
        out["n_urgent"] = int(nu[0])
        return out

    def context(self, kind, max_tasks):
        """build_context_summary of the ContextPair hybrids: f32 [8] (kind 0) or [1] (kind 1, raw)."""
        out = np.zeros(8, np.float32)
        n = self.L.orc_context(self.h, int(kind == 1), int(max_tasks), _p(out))
        return out[:n].copy()

    def scalars_last_plan(self):
        self.L.orc_last_plan_step.restype = C.c_longlong
        return int(self.L.orc_last_plan_step(self.h))

    def agent_commit_until(self):
        c = np.zeros(self.A, dtype=np.int32)
        self.L.orc_get_commit(self.h, _p(c))
        return c

    def tasks(self):
        nt = self.dims()["n_task_ids"]
        rows = np.zeros((nt, 14)); reqs = np.zeros((nt, 3, 6))
        self.L.orc_get_tasks(self.h, _p(rows), _p(reqs))
        return rows, reqs

    def task_org(self):
        """orgReqs[typeIdx] by task id."""
        out = np.zeros(self.dims()["n_task_ids"])
        self.L.orc_get_task_org(self.h, _p(out))
        return out

    def known(self):
        nt = self.dims()["n_task_ids"]
        k = np.zeros((self.A, nt), dtype=np.uint8)
        self.L.orc_get_known(self.h, _p(k))
        return k.astype(bool)

    def obstacles(self):
        out = np.zeros((8, 3))
        return out[:self.L.orc_get_obstacles(self.h, _p(out))]

    def threats(self):
        rows = np.zeros((self.dims()["n_threats"], 10))
        self.L.orc_get_threats(self.h, _p(rows))
        return rows

    def scalars(self):
        s = np.zeros(24)
        self.L.orc_get_scalars(self.h, _p(s))
        return s

    def open_ids(self):
        ids = np.zeros(self.dims()["n_open"], dtype=np.int32)
        self.L.orc_get_open(self.h, _p(ids))
        return ids

    def events(self):
        ev = np.zeros((self.dims()["n_events"], 2), dtype=np.int32)
        self.L.orc_get_events(self.h, _p(ev))
        return ev

    def last_actions(self):
        a = np.zeros((self.dims()["n_actions"], 2), dtype=np.int32)
        self.L.orc_get_actions(self.h, _p(a))
        return a

    def lsap_calls(self):
        n = self.dims()["n_lsap"]
        shapes = np.zeros((n, 2), dtype=np.int32)
        total = self.L.orc_get_lsap(self.h, _p(shapes), None, None, None)
        costs = np.zeros(total); nrc = int(np.minimum(shapes[:, 0], shapes[:, 1]).sum()) if n else 0
        rows = np.zeros(nrc, dtype=np.int64); cols = np.zeros(nrc, dtype=np.int64)
        self.L.orc_get_lsap(self.h, _p(shapes), _p(costs), _p(rows), _p(cols))
        return shapes, costs, rows, cols

    def observe(self):
        T = self.dims()["max_tasks"]
        ti = np.zeros((T, 21), dtype=np.float32); legal = np.zeros((self.A, T), dtype=np.uint8)
        pad = np.zeros(T, dtype=np.uint8); ag = np.zeros((self.A, 9), dtype=np.float32); fl = np.zeros(5, dtype=np.float32)
        self.L.orc_observe(self.h, _p(ti), _p(legal), _p(pad), _p(ag), _p(fl))
        return ti, legal.astype(bool), pad.astype(bool), ag, fl


def lsap(cost):
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    m = min(nr, nc)
    row = np.zeros(m, dtype=np.int64); col = np.zeros(m, dtype=np.int64)
    n = lib().orc_lsap(_p(cost), nr, nc, _p(row), _p(col))
    assert n == m
    return row, col


# ---------------------------------------------------------------------------------------------------------------
# Oracle episodes in parallel worker processes (spawned, so they never inherit a GPU context): the checker for the
# full-size GPU tests.  Returns the [len(seeds), 30] metrics array of `rollout_mode(seed, 150, interval, use_vis, mode)`.
def _metrics_worker(args):
    case, seeds, n_steps, interval, use_vis, mode = args
    import numpy as _np
    from muavta_amd.params import params_for_case
    e = OracleEnv(params_for_case(case))
    out = _np.zeros((len(seeds), 30))
    for i, s in enumerate(seeds):
        e.rollout_mode(int(s), n_steps, interval, use_vis, mode)
        out[i] = e.metrics()
    return out


def parallel_metrics(case, seeds, interval, use_vis=1, mode=0, n_steps=150, procs=None):
    import multiprocessing as mp
    lib()  # build the oracle once, before the workers race for it
    seeds = [int(s) for s in seeds]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    procs = max(1, min(procs or 32, cores, (len(seeds) + 63) // 64))
    if procs == 1:
        return _metrics_worker((case, seeds, n_steps, interval, use_vis, mode))
    chunk = (len(seeds) + procs * 4 - 1) // (procs * 4)
    jobs = [(case, seeds[i:i + chunk], n_steps, interval, use_vis, mode) for i in range(0, len(seeds), chunk)]
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.map(_metrics_worker, jobs)
    return np.concatenate(parts, axis=0)
