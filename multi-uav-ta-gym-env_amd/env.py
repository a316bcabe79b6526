"""``MultiUAVEnv`` — the reference's single-environment PettingZoo ``ParallelEnv`` surface
(mUAV_TA/DroneEnv.py:70-323, 522-762, 774-1206, 1209-1214) on top of the batched MI355X backend.

It is a *view*: every number comes from the device state through ``BatchedMultiUAVEnv.get`` /
``observe`` (n_envs = 1); no simulation logic runs in Python.  Allocators written against the
reference (``TaskAllocation/*``: they read ``env.agents_obj`` / ``env.tasks`` /
``env.agent_visibility_map()`` and return ``[(agent_name, Task)]``) and the harness glue
(``_open_tasks`` / ``_apply_assign``, experiments/paper_eval.py:85-101, wps_eval.py:55-61) work on the
object views below: UAV / Task / Threat proxies are cached per id, so identity tests such as
``task in env.last_tasks_info`` keep working.

``agent_visibility_map()`` is exact for every task that is still open; ids of tasks that were revealed only
after they had retired (the reference keeps adding those to its sets) may be absent — no caller looks them up.

The reference's out-of-step mutators that tests and allocator scaffolding call on the objects directly —
``UAV.allocate``, ``UAV.tasks = [...]``, ``_create_escort_for``, ``_sync_escorts``, ``_retire_escort``,
``_escort_fighters_near``, ``_is_task_action_valid``, ``_escort_by_recon`` (experiments/test_escort.py:61-75,95-96,236) —
go through ``muavta_call`` (include/muavta.h) and run the same device routines ``step`` uses.
"""
from __future__ import annotations

import random
import sys
from typing import Any, Dict, List, Optional

import numpy as np

from .params import EVENT_TAGS, METRIC_KEYS, INT_METRICS, REWARD_KEYS, TASK_TYPES, UAV_TYPES, _Cfg, params_from_config

MAX_INT = sys.maxsize
_TASK_DURATION = {"Hold": 1, "Rec": 10, "Att": 5, "Def": 5, "Int": 0, "Det": 1}   # MultiDroneEnvData.py:72-85
_MAX_SPEED = {"F1": 20.0, "F2": 15.0, "R1": 5.0, "R2": 8.0, "E1": 5.0, "T1": 14.0, "T2": 12.0}
_FAIL_MULTIPLIER = {"F1": 1.5, "F2": 0.8, "R1": 1.2, "R2": 0.8, "E1": 1.5, "T1": 1.8, "T2": 1.0}                  # MultiDroneEnvData.py:60-66
_ENGAGE = {"F1": 40.0, "F2": 30.0, "R1": 0.0, "R2": 0.0, "E1": 0.0, "T1": 35.0, "T2": 25.0}


class _Snapshot:
    """Lazily fetched device fields of env 0, dropped after every reset / step."""

    def __init__(self, backend):
        self.b = backend
        self.cache: Dict[str, np.ndarray] = {}

    def __getitem__(self, name: str) -> np.ndarray:
        if name not in self.cache:
            self.cache[name] = self.b.get(name)[0]
        return self.cache[name]

    @property
    def slot_map(self) -> Dict[int, int]:
        """Task.id -> device slot of the tasks resident now (one pass per snapshot: every TaskView field read looks its slot up here)"""
        m = self.cache.get("_slot_map")
        if m is None:
            m = self.cache["_slot_map"] = {t: s for s, t in enumerate(self["TASK_ID"].tolist()) if t >= 0}
        return m

    @property
    def status_map(self) -> Dict[int, int]:
        """Task.id -> status of the resident tasks (an id that is not resident is a retired task: status 2)"""
        m = self.cache.get("_status_map")
        if m is None:
            m = self.cache["_status_map"] = {t: s for t, s in zip(self["TASK_ID"].tolist(), self["TASK_STATUS"].tolist()) if t >= 0}
        return m

    def clear(self):
        self.cache.clear()


class TaskView:
    """Task (mUAV_TA/DroneEnvComponents.py:223-263).  ``hard_deadline`` is an attribute only on windowed
    tasks, exactly like the reference (callers use ``getattr(task, 'hard_deadline', None)``)."""

    info = None              # DroneEnvComponents.py:228 (no task on this path is created with one)

    def __init__(self, env, task_id: int):
        self._env, self.id = env, int(task_id)
        self._last: Dict[str, Any] = {}

    @property
    def max_time_steps(self) -> int:
        return self._env.max_time_steps   # :246

    def _slot(self) -> int:
        return self._env._snap.slot_map.get(self.id, -1)

    def _field(self, key: str, name: str, pick):
        s = self._slot()
        if s >= 0:
            self._last[key] = pick(self._env._snap[name][s])
        return self._last.get(key)

    @property
    def position(self):
        if self.id == 0:
            return np.array([0, 0])
        return self._field("position", "TASK_POS", lambda r: np.array(r, dtype=np.float64))

    @position.setter
    def position(self, value):
        s = self._slot()
        if s < 0:
            raise ValueError("task is retired and no longer resident on the device")
        arr = self._env._b.get("TASK_POS")
        arr[0, s] = np.asarray(value, dtype=np.float64)
        self._env._b.set("TASK_POS", arr)
        self._env._snap.clear()

    @property
    def status(self) -> int:
        if self.id == 0:
            return 0
        return self._env._snap.status_map.get(self.id, 2)  # freed slots are retired tasks

    def _meta(self, col: int, default=0):
        v = self._field(f"meta{col}", "TASK_META", lambda r: int(r[col]))
        return default if v is None else v

    @property
    def typeIdx(self) -> int:
        return 0 if self.id == 0 else self._meta(0)

    @property
    def type(self) -> str:
        return TASK_TYPES[self.typeIdx]

    @property
    def task_duration(self) -> int:
        return _TASK_DURATION[self.type]

    @property
    def currentReqs(self):
        if self.id == 0:
            return np.zeros(6)
        return self._field("cur", "TASK_CUR", lambda r: np.array(r))

    @property
    def allocatedReqs(self):
        if self.id == 0:
            return np.zeros(6)
        return self._field("alloc", "TASK_ALLOC", lambda r: np.array(r))

    @property
    def orgReqs(self):
        out = np.zeros(6)
        if self.id != 0:
            v = self._field("org", "TASK_ORG_DONE", lambda r: float(r[0]))
            out[self.typeIdx] = 0.0 if v is None else v
        return out

    @property
    def allocationDetails(self):
        """len()-able / `in`-able stand-in: {agent_id: (None, None)} for the agents queuing this task, padded with
        placeholder keys up to the device's len(allocationDetails) (entries the reference leaves behind for agents
        that no longer queue the task, DroneEnvComponents.py:98-127); the last view is kept once the slot is recycled."""
        env = self._env
        if self.id == 0:
            return {}
        if self._slot() >= 0:
            q = env._snap["AGENT_QUEUE"]
            d = {int(a): (None, None) for a in range(q.shape[0]) if self.id in q[a]}
            for k in range(self._meta(5) - len(d)):
                d[("stale", k)] = (None, None)
            self._last["details"] = d
        return self._last.get("details", {})

    @property
    def initTime(self):
        v = self._field("init", "TASK_TIMES", lambda r: float(r[0]))
        return -1 if v is None else v

    @property
    def doneTime(self):
        v = self._field("done", "TASK_TIMES", lambda r: float(r[1]))
        return -1 if v is None else v

    @property
    def created_at(self) -> int:
        return self._meta(2)

    @property
    def required_agents(self) -> int:
        return self._meta(3)

    @required_agents.setter
    def required_agents(self, value):
        s = self._slot()
        if s < 0:
            raise ValueError("task is retired and no longer resident on the device")
        arr = self._env._b.get("TASK_META")
        arr[0, s, 3] = int(value)
        self._env._b.set("TASK_META", arr)
        self._env._snap.clear()

    @property
    def kind(self):
        return "Escort" if self._meta(4) else None

    @property
    def protected_agent(self):
        a = self._meta(6, -1)
        return None if a < 0 else self._env.agents_obj[a]

    @property
    def eligible_agent_types(self):
        m = self._meta(7, -1)
        return None if m < 0 else {UAV_TYPES[t] for t in range(7) if (m >> t) & 1}

    @property
    def final_quality(self):
        return -1

    def __getattr__(self, name):
        if name == "hard_deadline":  # only present on windowed tasks
            d = self.__dict__.get("_last", {}).get("meta1")
            if self._slot() >= 0:
                d = int(self._env._snap["TASK_META"][self._slot()][1])
                self._last["meta1"] = d
            if d is None or d < 0:
                raise AttributeError(name)
            return d
        raise AttributeError(name)

    def __repr__(self):
        return f"<Task {self.id} {self.type} status={self.status}>"


class TaskSnapshot:
    """What `copy.deepcopy(task)` gives in the reference (DroneEnv.py:768): the public fields of a Task frozen at the moment of the copy
    (`protected_agent` by name: the copy must not follow the live agent)."""

    FIELDS = ("id", "info", "type", "typeIdx", "status", "max_time_steps", "task_duration", "initTime", "doneTime", "created_at", "final_quality",
              "kind", "required_agents")

    def __init__(self, task: "TaskView"):
        for f in self.FIELDS:
            setattr(self, f, getattr(task, f))
        self.position = np.array(task.position, dtype=np.float64)
        self.orgReqs, self.currentReqs, self.allocatedReqs = (np.array(x, dtype=np.float64) for x in (task.orgReqs, task.currentReqs, task.allocatedReqs))
        self.allocationDetails = dict(task.allocationDetails)
        e = task.eligible_agent_types
        self.eligible_agent_types = None if e is None else set(e)
        p = task.protected_agent
        self.protected_agent = None if p is None else p.name
        d = getattr(task, "hard_deadline", None)
        if d is not None:
            self.hard_deadline = d

    def __repr__(self):
        return f"<Task copy {self.id} {self.type} status={self.status}>"


class UAVView:
    """UAV (mUAV_TA/DroneEnvComponents.py:7-52)."""

    altitude = 1000          # DroneEnvComponents.py:9,16
    task_finished = -1       # :21 (never written again)
    has_capability = True    # :52 (never written again; swarm_gap.py:104 reads it)

    def __init__(self, env, agent_id: int):
        self._env, self.id = env, int(agent_id)

    @property
    def env(self):
        return self._env     # :11

    @property
    def fail_multiplier(self) -> float:
        return _FAIL_MULTIPLIER[self.type]   # :39, MultiDroneEnvData.py:60-66

    @property
    def name(self) -> str:
        n = self.__dict__.get("_name")  # (fixed by reset's shuffle for the whole episode: the view is rebuilt at every reset)
        if n is None:
            n = self._name = self._env.possible_agents[int(self._env._snap["AGENT_NAME_IDX"][self.id])]
        return n

    @property
    def typeIdx(self) -> int:
        return int(self._env._snap["AGENT_TYPE"][self.id])

    @property
    def type(self) -> str:
        return UAV_TYPES[self.typeIdx]

    @property
    def position(self):
        return np.array(self._env._snap["AGENT_POS"][self.id])

    @position.setter
    def position(self, value):
        arr = self._env._b.get("AGENT_POS")
        arr[0, self.id] = np.asarray(value, dtype=np.float64)
        self._env._b.set("AGENT_POS", arr)
        self._env._snap.clear()

    @property
    def state(self) -> int:
        return int(self._env._snap["AGENT_STATE"][self.id])

    @state.setter
    def state(self, value):
        arr = self._env._b.get("AGENT_STATE")
        arr[0, self.id] = int(value)
        self._env._b.set("AGENT_STATE", arr)
        self._env._snap.clear()

    @property
    def currentCap2Task(self):
        return np.array(self._env._snap["AGENT_CAPS"][self.id])

    @property
    def tasks(self) -> List[TaskView]:
        q = self._env._snap["AGENT_QUEUE"][self.id]
        return [self._env._task(int(t)) for t in q if t >= 0]

    @tasks.setter
    def tasks(self, value):
        """Plain list assignment (e.g. `a.tasks = [env.task_idle]`, experiments/test_escort.py:95): no Task bookkeeping."""
        ids = [int(t.id) for t in value]
        if len(ids) > 6:
            raise ValueError("at most 6 queued tasks can be assigned at once")
        self._env._b.call("set_queue", [self.id, len(ids)] + ids)
        self._env._after_call()

    @property
    def next_free_position(self):
        return np.array(self._env._snap["AGENT_NFP"][self.id])

    @property
    def next_free_time(self) -> float:
        return float(self._env._snap["AGENT_NFT"][self.id])

    @property
    def attackCap(self) -> int:
        return int(self._env._snap["AGENT_ATTACK_CAP"][self.id])

    @property
    def task_start(self) -> int:
        return int(self._env._snap["AGENT_MISC"][self.id][0])

    @property
    def fail_event(self) -> int:
        return int(self._env._snap["AGENT_MISC"][self.id][1])

    @property
    def re_eval(self) -> bool:
        return bool(self._env._snap["AGENT_MISC"][self.id][2])

    @property
    def commit_until(self) -> int:
        return int(self._env._snap["AGENT_MISC"][self.id][4])

    @commit_until.setter
    def commit_until(self, value):  # TaskAllocation/Hybrid/AttentionCommit.py:44 writes this
        arr = self._env._b.get("AGENT_MISC")
        arr[0, self.id, 4] = int(value)
        self._env._b.set("AGENT_MISC", arr)
        self._env._snap.clear()

    @property
    def max_speed(self) -> float:
        return _MAX_SPEED[self.type] / self._env._params.simulation_frame_rate * 0.02

    @property
    def engage_range(self) -> float:
        return _ENGAGE[self.type]

    def allocate(self, task, time_step):
        """UAV.allocate (DroneEnvComponents.py:55-95) on the device state; returns what the reference returns."""
        if task.id == 0:  # the idle task: `self.tasks = [task]`, next_free_* reset, returns False (:85-92)
            if any(t.id == 0 for t in self.tasks):
                return False
            self._env._b.call("set_queue", [self.id, 0, 0, 0, 0, 0, 0, 1])  # iargs[7] = 1: the allocate(idle) resets as well
            self._env._after_call()
            return False
        out = self._env._b.call("uav_allocate", [self.id, int(task.id), int(time_step)])
        self._env._after_call()
        return bool(out[0])

    def __repr__(self):
        return f"<UAV {self.name} id={self.id} state={self.state}>"


class ThreatView:
    def __init__(self, env, threat_id: int):
        self._env, self.id = env, int(threat_id)

    @property
    def position(self):
        return np.array(self._env._snap["THREAT_POS"][self.id])

    def _m(self, col):
        return int(self._env._snap["THREAT_META"][self.id][col])

    @property
    def status(self) -> int:
        return self._m(0)

    @property
    def target_agent(self):
        a = self._m(1)
        return None if a < 0 else self._env.agents_obj[a]

    @property
    def mission_target_agent(self):
        a = self._m(2)
        return None if a < 0 else self._env.agents_obj[a]

    @property
    def attackCap(self) -> int:
        return self._m(3)

    @property
    def relative_task(self):
        return self._env._task(self._m(4))

    @property
    def threat_type(self) -> str:
        return UAV_TYPES[self._m(5)]

    @property
    def threat_group(self) -> int:
        return self._m(6)

    @property
    def intercepting_agent(self):
        a = self._m(7)
        return None if a < 0 else self._env.agents_obj[a]


class MultiUAVEnv:
    metadata = {"render_modes": ["human"], "name": "multi_agent_env_v0"}

    def __init__(self, config=None, backend=None, device: int = 0, flags: Optional[dict] = None, backend_factory=None, **tiles):
        if config is None:
            config = {"agents": {"F1": 0, "F2": 0, "R1": 1, "R2": 1}, "tasks": {"Att": 0, "Rec": 2, "Hold": 0},
                      "threats_list": [("T1", 4), ("T2", 2)]}
        self.config = config
        self._params = params_from_config(config, flags, **tiles)
        # `backend_factory(params) -> backend`: how a handle for this env is made (kept: a write to `multiple_tasks_per_agent` re-creates it, see the
        # property below).  The default is the HIP library; an injected `backend` INSTANCE has no factory, and such an env refuses that write.
        self._backend_factory = backend_factory
        if backend is None:
            if backend_factory is None:
                def backend_factory(params, _device=device):
                    from .batched import BatchedMultiUAVEnv  # HIP library; raises MuavtaError when unavailable

                    return BatchedMultiUAVEnv(params, 1, _device)
                self._backend_factory = backend_factory
            backend = backend_factory(self._params)
        self._seed = None
        self._b = backend
        self._b.set_release_log(True)  # keeps agent_visibility_map() exact for ids whose slot was recycled
        self._snap = _Snapshot(backend)
        p = self._params
        self.area_width, self.area_height, self.max_coord = 1200, 700, 1200       # MultiDroneEnvData.py:8
        self.bases = [np.array([400, 680])]
        self.max_time_steps = p.max_time_steps
        self.n_agents = p.n_agents
        self.max_agents = max(48, self.n_agents + 8)                                 # DroneEnv.py:122
        self.possible_agents = p.possible_agents
        self.agents = list(self.possible_agents)
        self.n_tasks, self.max_tasks = p.n_tasks, p.max_tasks
        self.commit_horizon = p.commit_horizon
        self.reassign_penalty = p.reassign_penalty
        self.sense_radius, self.threat_delay = p.sense_radius, p.threat_delay
        self.escort_enabled = bool(p.escort_enabled)
        # the configuration as the reference env echoes it on itself (DroneEnv.py:101-201): planners and state builders read some of these
        # through getattr(env, ...) with a default, so a missing one is a silently different input (build_rah_state: `burst_mode`)
        c = _Cfg(config, flags)
        self.simulation_frame_rate, self.info = p.simulation_frame_rate, c.get("info", "No Info")
        self.render_speed, self.render_enabled, self.render_mode = -1, False, c.get("render_mode", "human")   # (nothing is rendered on this path)
        self.action_mode = "TaskAssign"
        self.agents_config, self.tasks_config = dict(c.get("agents")), dict(c.get("tasks"))
        self.threats_list = list(c.get("threats_list") or [])
        self.random_init_pos, self.num_obstacles, self.hidden_obstacles = bool(p.random_init_pos), p.num_obstacles, False
        self._multi_tasks = bool(p.multiple_tasks_per_agent)   # (read through the property below; multiple_agents_per_task has one live value)
        self.fail_rate = p.fail_rate
        self.early_terminate, self.capability_mask, self.saturate_mask = bool(p.early_terminate), bool(p.capability_mask), bool(p.saturate_mask)
        self.reward_weights = dict(zip(REWARD_KEYS, (float(w) for w in p.reward_weights)))
        self.arrival_rate, self.include_time_windows, self.dynamic_idle_penalty = p.arrival_rate, bool(p.include_time_windows), p.dynamic_idle_penalty
        self.hard_windows, self.window_length = bool(p.hard_windows), p.window_length
        self.burst_mode, self.burst_size, self.dual_region_bursts = bool(p.burst_mode), p.burst_size, bool(p.dual_region_bursts)
        self.miss_penalty, self.on_time_bonus = p.miss_penalty, p.on_time_bonus
        self.share_knowledge = bool(p.share_knowledge)
        self.escort_radius, self.escort_requirement = p.escort_radius, p.escort_requirement
        self.escort_intercept_radius, self.mutual_support_radius = p.escort_intercept_radius, p.mutual_support_radius
        self.escort_agent_types = tuple(c.get("escort_agent_types", ("F1", "F2")) or ("F1", "F2"))
        self.task_idle = TaskView(self, 0)
        self._tasks: Dict[int, TaskView] = {0: self.task_idle}
        self.agents_obj: List[UAVView] = []
        self._agent_names: List[str] = []
        self.agent_by_name: Dict[str, UAVView] = {}
        self._threats: Dict[int, ThreatView] = {}
        self._known: Dict[str, set] = {}
        self._views_upto, self._known_sig = 0, None
        fs = _Cfg(config, flags).get("fixed_seed", -1)
        self.fixed_seed = -1 if fs is None else int(fs)                            # DroneEnv.py:530-531: overrides every reset's seed
        self._steps = 0
        self.observations: Dict[str, dict] = {}
        self.last_tasks_info: Optional[List[TaskView]] = None

    @classmethod
    def batch(cls, config, n_envs: int, flags=None, backend=None, device: int = 0, **tiles) -> "MultiUAVEnvBatch":
        """`n_envs` env objects over ONE device handle: one launch per step for all of them (MultiUAVEnvBatch below)."""
        return MultiUAVEnvBatch(config, n_envs, flags=flags, backend=backend, device=device, **tiles)

    # ------------------------------------------------------------------ helpers
    def _task(self, tid: int) -> TaskView:
        t = self._tasks.get(tid)
        if t is None:
            t = self._tasks[tid] = TaskView(self, tid)
        return t

    def _scalar(self, col: int) -> float:
        return float(self._snap["SCALARS"][col])

    def _refresh(self, new_step: bool = True):
        self._snap.clear()
        ids = self._snap["TASK_ID"]
        n_ids = int(self._scalar(27))
        for tid in range(self._views_upto + 1, n_ids + 1):  # every id handed out so far has a view (incl. tasks created and retired between two observations)
            self._task(tid)
        self._views_upto = max(self._views_upto, n_ids)
        if ids.max(initial=-1) > n_ids:  # (a resident id beyond the counter: not produced by the device, kept for safety)
            for tid in ids[ids > n_ids]:
                self._task(int(tid))
        # (the fields of a task whose slot a STEP recycles reach its view through the release log below — its final record; slots recycled by an
        # out-of-step call are captured by _capture_resident() in front of the call.  Round 4 re-read every field of every resident task here,
        # every step: 6 ms of Python per step, 4x the reference's whole step)
        self.last_tasks_info = [self._task(i) for i in self._snap["OPEN_IDS"].tolist() if i >= 0]
        _ = self.threats  # register this step's spawns now, so the order is right even if nobody looks every step
        # agent_known_tasks grows monotonically in the reference; bits of recycled slots are folded in here
        known = self._snap["KNOWN"]
        sig = (ids.tobytes(), known.tobytes())
        if sig != self._known_sig:  # (same residents and same bits as at the last refresh: the sets already hold them)
            self._known_sig = sig
            bits = (known[:, :, None] >> np.arange(32, dtype=known.dtype)) & 1           # [A, KW, 32] -> slot-major bool rows
            bits = bits.reshape(known.shape[0], -1)[:, :len(ids)].astype(bool) & (ids >= 0)
            for i, name in enumerate(self._agent_names):
                self._known.setdefault(name, set()).update(ids[bits[i]].tolist())
        # ... ids that left the device during this step, with the agents that knew them then (muavta_set_release_log)
        if self._steps != self._log_step:
            self._log_step = self._steps
            log = self._snap["RELEASE_LOG"]
            n = int(log[:1].view(np.int32)[0])
            if n > (len(log) - 1) // 29:
                raise RuntimeError("release log overflow")
            for r in log[1:1 + 29 * n].reshape(n, 29):
                tid, mask = int(r[0]), int(r[1]) | (int(r[2]) << 32)
                for a in self.agents_obj:
                    if (mask >> a.id) & 1:
                        self._known[a.name].add(tid)
                # the task's final record (the last observation may predate changes of its last step, or not exist)
                last = self._task(tid)._last
                for col, v in zip(range(8), (r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10])):
                    last[f"meta{col}"] = int(v)
                last.update(position=np.array(r[11:13]), org=float(r[13]), init=float(r[15]), done=float(r[16]),
                            cur=np.array(r[17:23]), alloc=np.array(r[23:29]),
                            details={("stale", k): (None, None) for k in range(int(r[8]))})
        # a threat drags its Int task along even after that task was retired (update_threats, DroneEnv.py:1725-1744)
        tpos = self._snap["THREAT_POS"]
        for h, row in enumerate(self._snap["THREAT_META"].tolist()):
            t = self._tasks.get(row[4]) if row[0] != -9 and row[4] > 0 else None
            if t is not None and t._slot() < 0:
                t._last["position"] = np.array(tpos[h], dtype=np.float64)
        # ... and the team-wide reveal of a task whose slot was released before it came due (_register_dynamic_task /
        # _wps_process_reveals, DroneEnv.py:1491-1541): every id joins every set at created_at + max(threat_delay, 0)
        if self._params.share_knowledge and (self.sense_radius or self.threat_delay):
            now, delay = self.time_steps, max(int(self.threat_delay), 0)
            for tid, t in self._tasks.items():
                if tid == 0 or tid in self._revealed:
                    continue
                born = self._born.setdefault(tid, t._last.get("meta2", now))  # never resident: created (and gone) this step
                if t._slot() >= 0:
                    continue  # still resident: the device's known bits carry its reveal when the step that processes it has run
                              # (a task created BETWEEN steps — _create_escort_for / _sync_escorts called by a planner — is revealed by
                              # the next step's _wps_process_reveals, not at creation)
                if now >= born + delay:
                    self._revealed.add(tid)
                    for s in self._known.values():
                        s.add(tid)

    def _build_observations(self):
        o = self._b.observe()
        rows = o["tasks"][0]
        # bulk conversions once per step (a row dict then holds VIEWS of these arrays, as the reference's rows hold their own arrays)
        r64 = rows.astype(np.float64)
        ids, status = rows[:, 0].astype(np.int64).tolist(), rows[:, 3].astype(np.int64).tolist()
        extra = rows[:, 16:21].tolist()   # init_time, end_time, type_idx, unmet, age as Python floats (float(np.float32))
        windows, have_open = bool(self._params.include_time_windows), len(self.last_tasks_info) > 0
        tasks_info = []
        for j, st in enumerate(status):
            if st == -1:
                tasks_info.append({"status": -1})
                continue
            d = {"id": ids[j], "position": r64[j, 1:3], "status": st, "current_reqs": r64[j, 4:10], "alloc_reqs": r64[j, 10:16]}
            if ids[j] != 0 or have_open:
                e = extra[j]
                if windows:
                    d.update(init_time=e[0], end_time=e[1], type_idx=e[2])
                d.update(unmet=e[3], age=e[4])
            tasks_info.append(d)
        mask = o["mask"][0].astype(bool).tolist()
        agents64 = o["agents"][0].astype(np.float64)
        alloc = o["agents"][0, :, 8].astype(np.int64).tolist()
        legal = o["legal_mask"][0].astype(bool).tolist()
        flags = np.array(o["event_flags"][0], dtype=np.float32)
        self.observations = {}
        for i, name in enumerate(self._agent_names):  # (agents_obj[i].id == i)
            self.observations[name] = {
                "agent_position": agents64[i, 0:2],
                "agent_caps": agents64[i, 2:8],
                "alloc_task": alloc[i],
                "tasks_info": tasks_info,
                "mask": mask,
                "legal_mask": legal[i],
                "event_flags": flags.copy(),
            }

    # ------------------------------------------------------------------ PettingZoo surface
    def seed(self, seed):
        self.reset(seed=seed)                                                       # DroneEnv.py:518-519

    def reset(self, seed=None, return_info=True, options=None):
        if seed is None:  # the reference draws from the GLOBAL `random` module (DroneEnv.py:525-526): random.seed(s) in front of an unseeded
            seed = random.randint(0, MAX_INT)  # reset() gives the episode it gives there
        if self.fixed_seed != -1:
            seed = self.fixed_seed
        self._seed = int(seed)
        self._b.reset(np.array([self._seed], dtype=np.uint64))
        return self._after_reset()

    def _after_reset(self):
        """everything reset() does behind the device launch (a MultiUAVEnvBatch launches once for all its views)"""
        self._steps = 0
        self._tasks = {0: self.task_idle}
        self._threats = {}
        self._known = {}
        self._born, self._revealed = {}, set()
        self._views_upto, self._known_sig = 0, None
        self._log_step = 0  # the reset itself releases nothing
        self._snap.clear()
        self.agents_obj = [UAVView(self, a) for a in range(self.n_agents)]
        self._agent_names = [a.name for a in self.agents_obj]  # (fixed for the episode: reset's shuffle)
        self.agent_by_name = dict(zip(self._agent_names, self.agents_obj))
        self.agents = list(self.possible_agents)
        self._refresh()
        self._build_observations()
        self.rewards = dict.fromkeys(self._agent_names, 0)
        self.terminations = dict.fromkeys(self._agent_names, False)
        self.truncations = dict.fromkeys(self._agent_names, False)
        self.infos = {n: {} for n in self._agent_names}
        return self.observations, self.infos

    def step(self, actions):
        aa, ai = self._b.pack_actions([self._action_items(actions)])
        self._b.step(aa, ai)
        return self._after_step()

    def _action_items(self, actions):
        """the ORDERED (UAV.id, index) items of an actions dict {agent_name: index | [indices]} (DroneEnv.py:813-825)"""
        if not isinstance(actions, dict):
            raise TypeError("actions must be a dict {agent_name: index | [indices]} (DroneEnv.py:810)")
        items = []
        for name, idxs in actions.items():
            a = self.agent_by_name[name]
            for i in (idxs if isinstance(idxs, list) else [idxs]):
                items.append((a.id, int(i)))
        return items

    def _after_step(self):
        """everything step() does behind the device launch"""
        self._steps += 1
        self._refresh()
        self._build_observations()
        reward, term, trunc = self._b.step_result()
        self.rewards = dict.fromkeys(self._agent_names, float(reward[0]))
        self.terminations = dict.fromkeys(self._agent_names, bool(term[0]))
        self.truncations = dict.fromkeys(self._agent_names, bool(trunc[0]))
        self.infos = {n: {} for n in self._agent_names}
        self.infos["selected"] = self.possible_agents[self._steps % len(self.possible_agents)]  # agent_selector.next()
        ev = self._snap["EVENTS"]
        self.infos["events"] = [[EVENT_TAGS[int(t)], int(arg)] for t, arg in ev if t >= 0]
        if term[0] or trunc[0]:
            self.infos["metrics"] = self.calculate_metrics()
        return self.observations, self.rewards, self.terminations, self.truncations, self.infos

    def observe(self, agent):
        o = self.observations[agent]
        o["agent_id"] = agent
        return o

    # ------------------------------------------------------------------ attributes callers read
    @property
    def tasks(self) -> List[TaskView]:
        return [self._tasks[k] for k in sorted(self._tasks) if k != 0]

    @property
    def threats(self) -> List[ThreatView]:
        # env.threats is in spawn order (DroneEnv.py:1601-1643): groups in config order, ascending ids inside a step,
        # so appending the newly spawned ids of each step in ascending order reproduces it (dict keeps insertion order)
        for h, status in enumerate(self._snap["THREAT_META"][:, 0].tolist()):
            if status != -9 and h not in self._threats:
                self._threats[h] = ThreatView(self, h)
        return list(self._threats.values())

    time_steps = property(lambda self: int(self._scalar(0)))
    F_Reward = property(lambda self: self._scalar(2))
    total_distance = property(lambda self: self._scalar(3))
    n_on_time = property(lambda self: int(self._scalar(4)))
    n_missed_windows = property(lambda self: int(self._scalar(5)))
    n_windowed_tasks = property(lambda self: int(self._scalar(6)))
    n_task_switches = property(lambda self: int(self._scalar(7)))
    n_reallocations = property(lambda self: int(self._scalar(8)))
    n_arrivals = property(lambda self: int(self._scalar(9)))
    conclusion_time = property(lambda self: int(self._scalar(11)))
    escort_requests = property(lambda self: int(self._scalar(12)))
    escort_completed = property(lambda self: int(self._scalar(13)))
    escort_failed = property(lambda self: int(self._scalar(14)))
    escort_required_steps = property(lambda self: int(self._scalar(15)))
    escort_covered_steps = property(lambda self: int(self._scalar(16)))
    protection_breaches = property(lambda self: int(self._scalar(17)))
    threats_intercepted = property(lambda self: int(self._scalar(18)))
    recon_losses = property(lambda self: int(self._scalar(19)))
    escort_losses = property(lambda self: int(self._scalar(20)))
    mutual_support_engagements = property(lambda self: int(self._scalar(21)))
    protected_rec_completed = property(lambda self: int(self._scalar(22)))
    current_agent = property(lambda self: self.possible_agents[self._steps % len(self.possible_agents)])

    # The two action-mode switches (DroneEnv.py:156-157; read by step at :842,877-882) are parameters of the device handle, fixed when it is created.
    # The reference's main.py:130-141 assigns them on the env object right after reset().  `multiple_agents_per_task` has one live value (False is dead
    # code in the reference, :935): True is accepted, False raises.  A write that CHANGES `multiple_tasks_per_agent` re-creates the handle with the new
    # parameter and resets it with the episode's seed — reset never reads the switch, so the new handle stands exactly where the old one stood — which is
    # only right while nothing has happened since reset(): the new handle's state is compared with the old one's, field by field, and the write raises if
    # they differ (the env was stepped or mutated: the reference would change behaviour mid-episode there, this path does not offer that).
    @staticmethod
    def _fields_that_differ(old, new) -> List[str]:
        """names of the device fields in which two handles differ — the resident part of each: task rows of slots that hold a task, threat rows of spawned
        threats, known-bits of resident slots (what a free slot or an unspawned threat's row holds is not part of the env's state)"""
        out = []
        g = lambda b, f: np.asarray(b.get(f))[0]  # noqa: E731
        for f in ("SCALARS", "AGENT_POS", "AGENT_STATE", "AGENT_QUEUE", "AGENT_NFT", "AGENT_NFP", "AGENT_CAPS", "AGENT_MISC", "TASK_ID", "OPEN_IDS", "ESCORTS"):
            if not np.array_equal(g(old, f), g(new, f)):
                out.append(f)
        ids = g(old, "TASK_ID")
        if "TASK_ID" not in out:
            live = ids >= 0
            for f in ("TASK_STATUS", "TASK_POS", "TASK_META", "TASK_TIMES", "TASK_CUR", "TASK_ALLOC"):
                if not np.array_equal(g(old, f)[live], g(new, f)[live]):
                    out.append(f)
            ka, kb = g(old, "KNOWN"), g(new, "KNOWN")
            bits = lambda k: ((k[:, :, None] >> np.arange(32, dtype=k.dtype)) & 1).reshape(k.shape[0], -1)[:, :len(ids)][:, live]  # noqa: E731
            if ka.shape != kb.shape or not np.array_equal(bits(ka), bits(kb)):
                out.append("KNOWN")
        ta, tb = g(old, "THREAT_META"), g(new, "THREAT_META")
        if ta.shape != tb.shape or not np.array_equal(ta[:, 0], tb[:, 0]):
            out.append("THREAT_META")
        else:
            act = ta[:, 0] != -9
            if not np.array_equal(ta[act], tb[act]):
                out.append("THREAT_META")
            if not np.array_equal(g(old, "THREAT_POS")[act], g(new, "THREAT_POS")[act]):
                out.append("THREAT_POS")
        return out

    @property
    def multiple_agents_per_task(self) -> bool:
        return True

    @multiple_agents_per_task.setter
    def multiple_agents_per_task(self, value):
        if not value:
            raise ValueError("multiple_agents_per_task=False is dead code in the reference (DroneEnv.py:935) and not a mode of the device handle")

    @property
    def multiple_tasks_per_agent(self) -> bool:
        return self._multi_tasks

    @multiple_tasks_per_agent.setter
    def multiple_tasks_per_agent(self, value):
        value = bool(value)
        if value == self._multi_tasks:
            return
        why = None
        if self._backend_factory is None:
            why = "this env was given a backend instance, not a way to make one (backend_factory)"
        elif self._steps != 0:
            why = "the env has been stepped since reset()"
        if why is None:
            import copy

            params = copy.copy(self._params)
            params.multiple_tasks_per_agent = int(value)
            new = self._backend_factory(params)
            new.set_release_log(True)
            if self._seed is not None:
                new.reset(np.array([self._seed], dtype=np.uint64))
                differs = self._fields_that_differ(self._b, new)
                if differs:
                    getattr(new, "close", lambda: None)()
                    why = f"the env was changed since reset() ({', '.join(differs)})"
        if why is not None:
            raise ValueError(f"multiple_tasks_per_agent is a parameter of the device handle and can be changed right after reset() only: {why}.  "
                             f"Construct the env with multiple_tasks_per_agent={value} in its configuration instead")
        # (settings made on the old handle through `env.backend` — an allocator mode, lanes — are not carried over: the facade itself makes none)
        old, self._b, self._params, self._multi_tasks = self._b, new, params, value
        self._snap.b = new
        self._snap.clear()
        getattr(old, "close", lambda: None)()

    def get_live_agents(self):  # DroneEnv.py:1484-1486
        return [a for a, st in zip(self.agents_obj, self._snap["AGENT_STATE"].tolist()) if st != -1]

    def agent_visibility_map(self):  # :1595-1599
        if not self.sense_radius and not self.threat_delay:
            return None
        return {name: set(ids) for name, ids in self._known.items()}

    def known_tasks_for(self, agent_name=None):  # :1582-1593
        if agent_name is not None:
            ids = self._known.get(agent_name, set())
            return [t for t in self.tasks if t.id in ids or t.id == 0]
        if not self.sense_radius and not self.threat_delay:
            return list(self.tasks)
        known = set().union(*self._known.values()) if self._known else set()
        return [t for t in self.tasks if t.id in known or t.id == 0]

    def calculate_metrics(self) -> Dict[str, Any]:  # :1231-1319, key order preserved
        row = self._b.metrics()[0]
        return {k: (int(v) if k in INT_METRICS else float(v)) for k, v in zip(METRIC_KEYS, row)}

    def compute_s_wps(self) -> float:  # :1321-1337
        return float(self._b.metrics()[0][METRIC_KEYS.index("S_WPS")])

    def compute_s_esc(self) -> float:  # :2002-2011
        return float(self._b.metrics()[0][METRIC_KEYS.index("S_ESC")])

    def close(self):
        """pettingzoo's ParallelEnv.close() (main.py:273 calls it when a case is done): frees the device handle"""
        c = getattr(self._b, "close", None)
        if c is not None:
            c()

    backend = property(lambda self: self._b)  # the handle behind the facade (BatchedMultiUAVEnv on the GPU): get_state() / get_rng() checkpoint it

    def get_initial_state(self):
        """DroneEnv.py:764-771: deep copies of the agent names and of every Task as they stand now, `quality_table` (None on every path: :249) and an
        empty event list.  The copies are detached `TaskSnapshot`s: they keep the values of this moment while the env moves on, as a deep copy does.
        (The device-side checkpoint of a whole handle is `backend.get_state()` / `get_rng()`: INTEGRATION.md section 2.)"""
        return {"agents": list(self.agents), "tasks": [TaskSnapshot(t) for t in self.tasks], "quality_table": None, "events": []}

    # ------------------------------------------------------------------ spaces (DroneEnv.py:298-308, 310-323)
    def observation_space(self, agent):
        """The reference's declared per-agent space (which its observations do not follow): a gymnasium
        `Dict` of `Box`es when gymnasium is importable, else objects with the same `shape` / `low` / `high` / `dtype`."""
        Dict_, Box_, _ = _spaces()
        f32 = np.float32
        return Dict_({
            "agent_position": Box_(low=0, high=1, shape=(2,), dtype=f32),
            "agent_state": Box_(low=0, high=1, shape=(5,), dtype=f32),
            "agent_type": Box_(low=0, high=1, shape=(6,), dtype=f32),
            "next_free_time": Box_(low=0, high=1, shape=(1,), dtype=f32),
            "position_after_last_task": Box_(low=0, high=1, shape=(2,), dtype=f32),
            "tasks_info": Box_(low=0, high=1, shape=(self.max_tasks * 12,), dtype=f32),
        })

    def action_space(self, agent):
        """`Discrete(max_tasks)` for action_mode 'TaskAssign' (DroneEnv.py:283-285)."""
        return _spaces()[2](self.max_tasks)

    # ------------------------------------------------------------------ out-of-step mutators (muavta_call)
    def _capture_resident(self):
        """Read every field of every resident task into its view's cache.  An out-of-step call may recycle a slot on demand, and only
        env.step writes the release log: called in FRONT of such a call, so the views of the tasks it displaces keep their last values."""
        for t in self._tasks.values():
            if t.id and t._slot() >= 0:
                _ = (t.position, t.typeIdx, t.currentReqs, t.allocatedReqs, t.orgReqs, t.created_at, t.required_agents,
                     t.kind, t._meta(6, -1), t._meta(7, -1), getattr(t, "hard_deadline", None), t.initTime, t.doneTime, t.allocationDetails)

    def _after_call(self):
        """A mutator changed the device state behind the cached views: re-read it (new tasks get their proxies)."""
        self._refresh(new_step=False)

    @property
    def _escort_by_recon(self) -> Dict[str, TaskView]:
        """{recon UAV name: escort Task} in insertion order (DroneEnv.py:232)."""
        out = {}
        for a, tid in self._snap["ESCORTS"]:
            if a >= 0:
                out[self.agents_obj[int(a)].name] = self._task(int(tid))
        return out

    def _is_task_action_valid(self, agent, task) -> bool:  # :341-363
        if task is None or task.status == 2:
            return False
        if task.id == 0:  # task_idle = Task(0, ..., "Hold", {"Hold": 0.0}) (:589): never resident on the device
            if agent.tasks and agent.tasks[0].id == 0:
                return True
            if self.capability_mask and agent.currentCap2Task[0] <= 0:
                return False
            return not self.saturate_mask  # allocatedReqs[Hold] (0.0) >= orgReqs[Hold] (0.0)
        return bool(self._b.call("action_valid", [agent.id, int(task.id)])[0])

    def _create_escort_for(self, recon_agent, rec_task):  # :1888-1917
        if not self.escort_enabled or recon_agent is None:
            return None
        self._capture_resident()
        out = self._b.call("create_escort", [recon_agent.id, int(rec_task.id) if rec_task is not None else 0])
        self._after_call()
        return None if out[0] < 0 else self._task(int(out[0]))

    def _sync_escorts(self):  # :1964-2000
        self._capture_resident()
        self._b.call("sync_escorts")
        self._after_call()

    def _retire_escort(self, escort_task, failed: bool = False):  # :1938-1950
        if escort_task is None or escort_task.status == 2:
            return
        self._capture_resident()
        self._b.call("retire_escort", [int(escort_task.id), int(bool(failed))])
        self._after_call()

    def _retire_escort_for(self, recon_agent, failed: bool = False):  # :1952-1957
        if recon_agent is not None:
            self._retire_escort(self._escort_by_recon.get(recon_agent.name), failed=failed)

    def _escort_fighters_near(self, protected_agent, radius=None):  # :1746-1764
        if protected_agent is None:
            return []
        out = self._b.call("escort_fighters_near", [protected_agent.id], -1.0 if radius is None else float(radius))
        return [self.agents_obj[int(i)] for i in out[1:1 + int(out[0])]]


class _Box:
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class _Dict(dict):
    @property
    def spaces(self):
        return self


class _Discrete:
    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.int64


def _spaces():
    try:
        from gymnasium.spaces import Box, Dict as GDict, Discrete
        return GDict, Box, Discrete
    except Exception:
        return _Dict, _Box, _Discrete


# ---------------------------------------------------------------------------------------------------------------------------------------
# Vectorised facade: n reference-shaped env objects over ONE handle
# ---------------------------------------------------------------------------------------------------------------------------------------
class _SharedBatch:
    """Whole-batch fetches of the one handle behind a MultiUAVEnvBatch, kept until the next launch / mutator call."""

    def __init__(self, backend):
        self.b = backend
        self.cache: Dict[Any, Any] = {}

    def invalidate(self):
        self.cache.clear()

    def fetch(self, key, fn):
        if key not in self.cache:
            self.cache[key] = fn()
        return self.cache[key]


class _RowBackend:
    """Row `i` of the shared handle behind the 1-env backend surface MultiUAVEnv talks to (every array keeps a leading axis of 1)."""

    def __init__(self, shared: _SharedBatch, i: int):
        self.S, self.i = shared, int(i)
        b = shared.b
        self.params, self.n_envs, self.n_agents, self.A_tile, self.T = b.params, 1, b.n_agents, b.A_tile, b.T
        self.max_tasks, self.possible_agents, self.dims, self.device_index = b.max_tasks, b.possible_agents, b.dims, getattr(b, "device_index", 0)

    def _row(self, x):
        return x[self.i:self.i + 1]

    def get(self, name):
        return self._row(self.S.fetch(("get", name), lambda: self.S.b.get(name)))

    def set(self, name, value):
        full = self.S.b.get(name)
        full[self.i] = np.asarray(value)[0]
        self.S.b.set(name, full)
        self.S.invalidate()

    def observe(self):
        o = self.S.fetch("observe", self.S.b.observe)
        return {k: self._row(v) for k, v in o.items()}

    def step_result(self):
        r, t, u = self.S.fetch("step_result", self.S.b.step_result)
        return self._row(r), self._row(t), self._row(u)

    def metrics(self):
        return self._row(self.S.fetch("metrics", self.S.b.metrics))

    def allocate(self, replan_interval: int = 20, use_visibility: bool = True, fetch: bool = True):
        """the device allocator's plan for this env (one launch for the whole batch, kept until the next step)"""
        aa, ai = self.S.fetch(("allocate", int(replan_interval), bool(use_visibility)), lambda: self.S.b.allocate(replan_interval, use_visibility))
        return self._row(aa), self._row(ai)

    def call(self, op, iargs=(), darg: float = -1.0, env_index: int = 0):
        out = self.S.b.call(op, iargs, darg, env_index=self.i)
        self.S.invalidate()
        return out

    def set_release_log(self, enable: bool = True):
        pass  # (the batch switched it on for the whole handle)

    def pack_actions(self, per_env):
        return self.S.b.pack_actions(per_env)

    def reset(self, seeds):
        raise RuntimeError("a view of a MultiUAVEnvBatch is reset through the batch (one launch for all envs): batch.reset(seeds)")

    def step(self, act_agent, act_index):
        raise RuntimeError("a view of a MultiUAVEnvBatch is stepped through the batch (one launch for all envs): batch.step([actions, ...])")

    def get_state(self):
        raise NotImplementedError("whole-handle snapshot: use MultiUAVEnvBatch.backend.get_state()")

    get_rng = get_state


class MultiUAVEnvBatch:
    """`n_envs` MultiUAVEnv objects — the reference's env surface, object views included — that share ONE device handle: a Python-side
    allocator loop written against the reference (experiments/wps_eval.py:112-133,273: `hung.allocate_tasks(env.get_live_agents(),
    _open_tasks(env), ...)`, `_apply_assign`, `env.step`) runs over `batch.envs[i]` unchanged, and the batch pays one launch, one state
    mirror and one observation copy per step for ALL of them instead of one each.

        batch = MultiUAVEnv.batch(config, 64, flags=...)
        outs = batch.reset(seeds)                                  # [(obs, infos)] per env
        while ...: outs = batch.step([actions_of(env) for env in batch.envs])   # [(obs, rewards, terms, truncs, infos)] per env

    An env whose episode has ended keeps receiving steps like its neighbours (pass {} for it); as with the reference, what a step after the
    end returns is the caller's to ignore.  `views=[...]` limits the Python-side refresh (object views, observation dicts) of a step to those
    envs — the others return None and catch up at their next refreshed step."""

    def __init__(self, config, n_envs: int, flags=None, backend=None, device: int = 0, **tiles):
        from .params import params_from_config as _pfc

        params = _pfc(config, flags, **tiles)
        if backend is None:
            from .batched import BatchedMultiUAVEnv

            backend = BatchedMultiUAVEnv(params, int(n_envs), device)
        if backend.n_envs != int(n_envs):
            raise ValueError("backend holds another number of envs")
        self.backend = backend
        backend.set_release_log(True)
        self._S = _SharedBatch(backend)
        self.envs: List[MultiUAVEnv] = [MultiUAVEnv(config, flags=flags, backend=_RowBackend(self._S, i), **tiles) for i in range(int(n_envs))]
        self.n_envs = int(n_envs)

    def __len__(self):
        return self.n_envs

    def __getitem__(self, i):
        return self.envs[i]

    def reset(self, seeds):
        seeds = np.asarray(seeds, dtype=np.uint64)
        self.backend.reset(seeds)
        self._S.invalidate()
        out = []
        for e, sd in zip(self.envs, seeds):
            e._seed = int(sd)
            out.append(e._after_reset())
        return out

    def step(self, actions_list, views=None):
        if len(actions_list) != self.n_envs:
            raise ValueError(f"one actions dict per env: {self.n_envs}")
        aa, ai = self.backend.pack_actions([e._action_items(a or {}) for e, a in zip(self.envs, actions_list)])
        self.backend.step(aa, ai)
        self._S.invalidate()
        todo = set(range(self.n_envs) if views is None else views)
        out = []
        for i, e in enumerate(self.envs):
            if i in todo:
                out.append(e._after_step())
            else:
                e._steps += 1   # (its release-log rows of this step are skipped: a view that sits out a step loses the ids released in it)
                out.append(None)
        return out

    def close(self):
        self.backend.close()
