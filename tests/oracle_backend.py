"""Test-only stand-in for BatchedMultiUAVEnv (n_envs = 1) on top of the CPU oracle, so the pure-Python
facade (muavta_amd.env.MultiUAVEnv) can be exercised without a GPU.  Lives under tests/ on purpose:
the product never imports the oracle.  Slot space here is simply slot == task id."""
import numpy as np

import orc


class OracleBackend:
    def __init__(self, params):
        self.params = params
        self.o = orc.OracleEnv(params)
        self.n_envs, self.n_agents = 1, params.n_agents
        self.A_tile = max(16, params.n_agents)
        self.max_tasks = params.max_tasks
        self.possible_agents = params.possible_agents
        self._last_alloc = None

    def reset(self, seeds):
        self.o.reset(int(seeds[0]))

    def pack_actions(self, per_env):
        cap = max(self.A_tile, len(per_env[0]))  # (as the product: as wide as the longest list needs)
        aa = np.full((1, cap), -1, dtype=np.int32)
        ai = np.zeros((1, cap), dtype=np.int32)
        for k, (a, i) in enumerate(per_env[0]):
            aa[0, k], ai[0, k] = a, i
        return aa, ai

    def step(self, aa, ai):
        n = int((aa[0] >= 0).sum()) if (aa[0] < 0).any() else aa.shape[1]
        k = 0
        while k < aa.shape[1] and aa[0, k] >= 0:
            k += 1
        self.o.step(aa[0, :k], ai[0, :k])

    def set_allocator(self, name="hungarian"):
        self._mode = {"hungarian": 0, "urgency_pair": 1, "urgency_coalition": 2, "hungarian_gated": 3}[name]

    def allocate(self, interval=20, use_visibility=True, fetch=True):
        a, i = self.o.allocate_mode(interval, int(use_visibility), getattr(self, "_mode", 0))
        aa, ai = self.pack_actions([list(zip(a.tolist(), i.tolist()))])
        return aa, ai

    def step_result(self):
        d = self.o.dims()
        return np.array([self.o.scalars()[1]]), np.array([bool(d["terminated"])]), np.array([bool(d["truncated"])])

    def metrics(self):
        return self.o.metrics()[None]

    def observe(self):
        ti, legal, pad, ag, fl = self.o.observe()
        return {"tasks": ti[None], "legal_mask": legal[None], "mask": pad[None], "agents": ag[None], "event_flags": fl[None]}

    def get_state(self):
        raise NotImplementedError

    get_rng = get_state

    OPS = {"uav_allocate": 0, "create_escort": 1, "sync_escorts": 2, "retire_escort": 3, "escort_fighters_near": 4,
           "action_valid": 5, "set_queue": 6}

    def call(self, op, iargs=(), darg=-1.0, env_index=0):
        ia = np.zeros(8, dtype=np.int32)
        ia[: len(iargs)] = np.asarray(list(iargs), dtype=np.int32)
        out = np.zeros(72, dtype=np.int32)
        rc = self.o.L.orc_call(self.o.h, self.OPS[op], orc._p(ia), orc.C.c_double(float(darg)), orc._p(out))
        if rc != 0:
            raise ValueError(f"orc_call({op}) rejected its arguments")
        return out

    def set(self, name, value):
        """The attribute writes the facade's proxies make (slot == task id here)."""
        L, h, v = self.o.L, self.o.h, np.asarray(value)
        if name == "AGENT_POS":
            for a in range(self.n_agents):
                L.orc_set_agent_pos(h, a, orc.C.c_double(float(v[0, a, 0])), orc.C.c_double(float(v[0, a, 1])))
        elif name == "AGENT_STATE":
            for a in range(self.n_agents):
                L.orc_set_agent_state(h, a, int(v[0, a]))
        elif name == "AGENT_MISC":
            for a in range(self.n_agents):
                L.orc_set_agent_commit(h, a, int(v[0, a, 4]))
        elif name == "TASK_POS":
            for t in range(1, v.shape[1]):
                L.orc_set_task_pos(h, t, orc.C.c_double(float(v[0, t, 0])), orc.C.c_double(float(v[0, t, 1])))
        elif name == "TASK_META":
            for t in range(1, v.shape[1]):
                L.orc_set_task_required(h, t, int(v[0, t, 3]))
        else:
            raise KeyError(name)

    def set_release_log(self, enable=True):  # the oracle keeps every task id resident: nothing is ever released
        pass

    def get(self, name):
        if name == "RELEASE_LOG":
            return np.zeros((1, 30), dtype=np.float64)
        if name == "ESCORTS":
            out = np.full((self.A_tile, 2), -1, dtype=np.int32)
            self.o.L.orc_get_escorts(self.o.h, orc._p(out), self.A_tile)
            return out[None]
        rows, caps, q = self.o.agents()
        trow, reqs = self.o.tasks()
        nt = trow.shape[0]
        if name == "TASK_ID":
            ids = np.arange(nt, dtype=np.int32); ids[0] = -1
            return ids[None]
        if name == "TASK_STATUS":
            return trow[:, 0].astype(np.int32)[None]
        if name == "TASK_POS":
            return trow[:, 1:3][None].copy()
        if name == "TASK_CUR":
            return reqs[:, 0][None].copy()
        if name == "TASK_ALLOC":
            return reqs[:, 1][None].copy()
        if name == "TASK_ORG_DONE":
            ty = trow[:, 6].astype(int)
            done = reqs[np.arange(nt), 2, ty]
            return np.stack([self.o.task_org(), done], axis=1)[None]
        if name == "TASK_TIMES":
            return trow[:, 3:5][None].copy()
        if name == "TASK_META":
            m = np.zeros((nt, 8), dtype=np.int32)
            m[:, 0:5] = trow[:, 6:11]; m[:, 5] = trow[:, 5]; m[:, 6] = trow[:, 11]; m[:, 7] = trow[:, 12]
            return m[None]
        if name == "AGENT_POS":
            return rows[:, 0:2][None].copy()
        if name == "AGENT_STATE":
            return rows[:, 2].astype(np.int32)[None]
        if name == "AGENT_HEAD":
            return rows[:, 3].astype(np.int32)[None]
        if name == "AGENT_QUEUE":
            return q[None].copy()
        if name == "AGENT_NFT":
            return rows[:, 5][None].copy()
        if name == "AGENT_NFP":
            return rows[:, 6:8][None].copy()
        if name == "AGENT_CAPS":
            return caps[None].copy()
        if name == "AGENT_ATTACK_CAP":
            return rows[:, 8].astype(np.int32)[None]
        if name == "AGENT_TYPE":
            return rows[:, 12].astype(np.int32)[None]
        if name == "AGENT_NAME_IDX":
            return rows[:, 13].astype(np.int32)[None]
        if name == "AGENT_MISC":
            m = np.zeros((rows.shape[0], 6), dtype=np.int32)
            m[:, 0] = rows[:, 9]; m[:, 1] = rows[:, 14]; m[:, 2] = rows[:, 10]; m[:, 3] = rows[:, 11]; m[:, 4] = self.o.agent_commit_until(); m[:, 5] = rows[:, 4]
            return m[None]
        if name == "KNOWN":
            k = self.o.known()
            words = (nt + 31) // 32
            out = np.zeros((k.shape[0], words), dtype=np.uint32)
            for a in range(k.shape[0]):
                for t in np.nonzero(k[a])[0]:
                    out[a, t >> 5] |= np.uint32(1 << (t & 31))
            return out[None]
        if name == "THREAT_POS":
            return self.o.threats()[:, 1:3][None].copy()
        if name == "THREAT_META":
            th = self.o.threats()
            m = np.stack([th[:, 0], th[:, 3], th[:, 4], th[:, 5], th[:, 6], th[:, 7], th[:, 8], th[:, 9]], axis=1).astype(np.int32)
            return m[None]
        if name == "SCALARS":
            d = self.o.dims()
            s = np.concatenate([self.o.scalars(), [d["pending_reset"], d["n_reached"], d["n_pending"], d["n_task_ids"] - 1]])
            return s[None]
        if name == "OPEN_IDS":
            oi = self.o.open_ids()
            out = np.full(max(nt, 1), -1, dtype=np.int32); out[: len(oi)] = oi
            return out[None]
        if name == "EVENTS":
            ev = self.o.events()
            out = np.full((max(len(ev), 1) + 1, 2), -1, dtype=np.int32); out[: len(ev)] = ev
            return out[None]
        raise KeyError(name)



class OracleBatchBackend:
    """n OracleBackends behind the surface of an n-env BatchedMultiUAVEnv — the test stand-in for the VECTORISED facade (MultiUAVEnv.batch):
    every array stacked on axis 0; the task- / event-indexed axes, whose length is per env here (slot == task id), padded to the widest env
    (pad rows of TASK_ID / OPEN_IDS / EVENTS are -1, i.e. free slots)."""
    PAD_MINUS_ONE = ("TASK_ID", "OPEN_IDS", "EVENTS")

    def __init__(self, params, n_envs):
        self.backs = [OracleBackend(params) for _ in range(int(n_envs))]
        b = self.backs[0]
        self.params, self.n_envs, self.n_agents, self.A_tile = params, int(n_envs), b.n_agents, b.A_tile
        self.max_tasks, self.possible_agents = b.max_tasks, b.possible_agents
        self.T, self.dims, self.device_index = None, None, 0
        self.launches = {"reset": 0, "step": 0, "allocate": 0, "observe": 0, "get": 0}  # whole-batch calls: the batch pays ONE per step

    def reset(self, seeds):
        self.launches["reset"] += 1
        for b, s in zip(self.backs, np.asarray(seeds).tolist()):
            b.reset([s])

    def pack_actions(self, per_env):
        cap = max([self.A_tile] + [len(acts) for acts in per_env])
        aa = np.full((self.n_envs, cap), -1, dtype=np.int32)
        ai = np.zeros((self.n_envs, cap), dtype=np.int32)
        for n, acts in enumerate(per_env):
            for k, (a, i) in enumerate(acts):
                aa[n, k], ai[n, k] = a, i
        return aa, ai

    def step(self, aa, ai):
        self.launches["step"] += 1
        for i, b in enumerate(self.backs):
            b.step(aa[i:i + 1], ai[i:i + 1])

    def set_allocator(self, name="hungarian"):
        for b in self.backs:
            b.set_allocator(name)

    @staticmethod
    def _stack(rows, fill=0):
        """[1, ...] arrays whose trailing axes may differ in length -> one [n, ...] array, the short ones padded with `fill`"""
        shape = tuple(max(r.shape[d] for r in rows) for d in range(1, rows[0].ndim))
        out = np.full((len(rows),) + shape, fill, dtype=rows[0].dtype)
        for i, r in enumerate(rows):
            out[(i,) + tuple(slice(0, k) for k in r.shape[1:])] = r[0]
        return out

    def allocate(self, interval=20, use_visibility=True, fetch=True):
        self.launches["allocate"] += 1
        plans = [b.allocate(interval, use_visibility) for b in self.backs]
        return self._stack([p[0] for p in plans], -1), self._stack([p[1] for p in plans], 0)

    def step_result(self):
        r = [b.step_result() for b in self.backs]
        return tuple(np.concatenate([x[k] for x in r]) for k in range(3))

    def metrics(self):
        return np.concatenate([b.metrics() for b in self.backs])

    def observe(self):
        self.launches["observe"] += 1
        o = [b.observe() for b in self.backs]
        return {k: np.concatenate([x[k] for x in o]) for k in o[0]}

    def call(self, op, iargs=(), darg=-1.0, env_index=0):
        return self.backs[env_index].call(op, iargs, darg)

    def get(self, name):
        self.launches["get"] += 1
        return self._stack([b.get(name) for b in self.backs], -1 if name in self.PAD_MINUS_ONE else 0)

    def set(self, name, value):
        v = np.asarray(value)
        for i, b in enumerate(self.backs):
            own = b.get(name)  # (this env's own extent: the rows beyond it are padding)
            b.set(name, v[(slice(i, i + 1),) + tuple(slice(0, k) for k in own.shape[1:])])

    def set_release_log(self, enable=True):
        pass

    def get_state(self):
        raise NotImplementedError

    get_rng = get_state

    def close(self):
        pass
