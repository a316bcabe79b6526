"""ctypes mirror of ``MuavtaParams`` / ``MuavtaDims`` (include/muavta.h) and the packing of a
reference-style config into it.

``params_from_config`` accepts either an ``agentEnvOptions``-like object (attribute access, the
reference's mUAV_TA/MultiDroneEnvUtils.py:5-105) or a scenario spec dict + flag dict
(experiments/paper_eval.py:42-82 ``make_config``) and applies the same coercions
``MultiUAVEnv.__init__`` applies (mUAV_TA/DroneEnv.py:176-217: ``x or default``).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

ABI_VERSION = 1
MAX_GROUPS = 8
N_METRICS = 30
N_SCALARS = 28

UAV_TYPES = ["R1", "R2", "E1", "F1", "F2", "T1", "T2"]           # MultiDroneEnvData.py:15
TASK_TYPES = ["Hold", "Rec", "Att", "Def", "Int", "Det"]          # MultiDroneEnvData.py:18
EVENT_TAGS = ["Reset_Allocation", "New_Threat", "Agent_Fail", "Escort_Created", "Escort_Retired"]
REWARD_KEYS = ["action", "distance", "quality", "s_quality", "time", "alloc", "time_penaulty", "step"]
METRIC_KEYS = [
    "F_time", "F_distance", "F_quality", "F_Reward", "S_WPS", "S_ESC", "Losses", "Kills", "makespan",
    "total_distance", "n_reallocations", "n_task_switches", "n_arrivals", "n_tasks_final", "n_reached",
    "n_missed_windows", "n_on_time", "n_windowed_tasks", "on_time_rate", "reserve_idle_fraction",
    "escort_coverage_rate", "protected_rec_completed", "recon_losses", "escort_losses",
    "threats_intercepted", "mutual_support_engagements", "protection_breaches", "escort_requests",
    "escort_completed", "escort_failed",
]
INT_METRICS = {
    "Losses", "Kills", "n_reallocations", "n_task_switches", "n_arrivals", "n_tasks_final", "n_reached",
    "n_missed_windows", "n_on_time", "n_windowed_tasks", "protected_rec_completed", "recon_losses",
    "escort_losses", "threats_intercepted", "mutual_support_engagements", "protection_breaches",
    "escort_requests", "escort_completed", "escort_failed",
}


class MuavtaParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_agent_groups", C.c_int32),
        ("agent_type", C.c_int32 * MAX_GROUPS),
        ("agent_count", C.c_int32 * MAX_GROUPS),
        ("n_task_groups", C.c_int32),
        ("task_type", C.c_int32 * MAX_GROUPS),
        ("task_count", C.c_int32 * MAX_GROUPS),
        ("n_threat_groups", C.c_int32),
        ("threat_type", C.c_int32 * MAX_GROUPS),
        ("threat_count", C.c_int32 * MAX_GROUPS),
        ("max_time_steps", C.c_int32),
        ("multiple_tasks_per_agent", C.c_int32),
        ("early_terminate", C.c_int32),
        ("capability_mask", C.c_int32),
        ("saturate_mask", C.c_int32),
        ("include_time_windows", C.c_int32),
        ("threat_delay", C.c_int32),
        ("hard_windows", C.c_int32),
        ("window_length", C.c_int32),
        ("burst_mode", C.c_int32),
        ("burst_size", C.c_int32),
        ("dual_region_bursts", C.c_int32),
        ("share_knowledge", C.c_int32),
        ("commit_horizon", C.c_int32),
        ("escort_enabled", C.c_int32),
        ("escort_agent_type_mask", C.c_uint32),
        ("num_obstacles", C.c_int32),
        ("simulation_frame_rate", C.c_double),
        ("fail_rate", C.c_double),
        ("reward_weights", C.c_double * 8),
        ("arrival_rate", C.c_double),
        ("dynamic_idle_penalty", C.c_double),
        ("sense_radius", C.c_double),
        ("miss_penalty", C.c_double),
        ("on_time_bonus", C.c_double),
        ("reassign_penalty", C.c_double),
        ("escort_radius", C.c_double),
        ("escort_requirement", C.c_double),
        ("escort_intercept_radius", C.c_double),
        ("mutual_support_radius", C.c_double),
        ("tile_agents", C.c_int32),
        ("tile_tasks", C.c_int32),
        ("tile_threats", C.c_int32),
        ("random_init_pos", C.c_int32),
    ]

    @property
    def n_agents(self) -> int:
        return sum(self.agent_count[: self.n_agent_groups])

    @property
    def n_tasks(self) -> int:  # DroneEnv.py:145
        return sum(self.task_count[: self.n_task_groups]) + 1

    @property
    def max_tasks(self) -> int:  # DroneEnv.py:147
        return self.n_tasks + 28

    @property
    def possible_agents(self):  # DroneEnv.py:124-127
        names = []
        for g in range(self.n_agent_groups):
            for i in range(self.agent_count[g]):
                names.append(f"{UAV_TYPES[self.agent_type[g]][0:2]}_agent{i}")
        return names


class MuavtaDims(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_agents", C.c_int32), ("tile_agents", C.c_int32), ("tile_tasks", C.c_int32),
        ("tile_threats", C.c_int32), ("max_tasks", C.c_int32), ("obs_task_width", C.c_int32),
        ("obs_agent_width", C.c_int32), ("queue_cap", C.c_int32), ("event_cap", C.c_int32),
        ("action_cap", C.c_int32), ("state_bytes", C.c_int64), ("n_threats", C.c_int32), ("known_words", C.c_int32),
        ("lds_bytes", C.c_int32), ("legal_words", C.c_int32),
    ]


class MuavtaRecord(C.Structure):
    """include/muavta.h: MuavtaRecord (device pointers of the per-step rings of muavta_rollout_record)."""
    _fields_ = [("kind", C.c_int32), ("max_tasks", C.c_int32), ("max_agents", C.c_int32), ("reserved", C.c_int32),
                ("task_feats", C.c_void_p), ("task_mask", C.c_void_p), ("task_ids", C.c_void_p), ("agent_feats", C.c_void_p),
                ("agent_mask", C.c_void_p), ("agent_ids", C.c_void_p), ("edge_valid", C.c_void_p), ("n_urgent", C.c_void_p),
                ("expert_mask", C.c_void_p), ("replanned", C.c_void_p), ("s_wps", C.c_void_p),
                ("obs_tasks", C.c_void_p), ("obs_legal", C.c_void_p), ("obs_pad", C.c_void_p), ("obs_agents", C.c_void_p),
                ("obs_flags", C.c_void_p), ("obs_reward", C.c_void_p), ("obs_done", C.c_void_p)]


class _Cfg:
    """Uniform getattr-with-default view over an options object or a (spec, flags) dict pair."""

    def __init__(self, obj: Any, flags: Optional[Dict[str, Any]] = None):
        self.obj, self.flags = obj, flags or {}

    def get(self, key: str, default: Any = None) -> Any:
        if isinstance(self.obj, dict):
            if key in self.obj:
                return self.obj[key]
            return self.flags.get(key, default)
        return getattr(self.obj, key, default)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def params_from_config(config: Any, flags: Optional[Dict[str, Any]] = None, *, tile_agents: int = 0,
                       tile_tasks: int = 0, tile_threats: int = 0) -> MuavtaParams:
    c = _Cfg(config, flags)
    p = MuavtaParams()
    p.abi_version = ABI_VERSION

    def groups(d, table, n_attr, t_attr, c_attr):
        items = list(d.items()) if isinstance(d, dict) else list(d or [])
        if len(items) > MAX_GROUPS:
            raise ValueError(f"at most {MAX_GROUPS} groups supported, got {len(items)}")
        setattr(p, n_attr, len(items))
        for i, (name, cnt) in enumerate(items):
            getattr(p, t_attr)[i] = table.index(name)
            getattr(p, c_attr)[i] = int(cnt)

    groups(c.get("agents"), UAV_TYPES, "n_agent_groups", "agent_type", "agent_count")
    groups(c.get("tasks"), TASK_TYPES, "n_task_groups", "task_type", "task_count")
    groups(c.get("threats_list") or [], UAV_TYPES, "n_threat_groups", "threat_type", "threat_count")
    if str(c.get("action_mode", "TaskAssign")) != "TaskAssign":
        raise ValueError("only action_mode='TaskAssign' exists on this path (DroneEnv.py:794)")
    if not c.get("multiple_agents_per_task", True):
        raise ValueError("multiple_agents_per_task=False is dead code in the reference (DroneEnv.py:935)")
    if c.get("hidden_obstacles", False):
        raise ValueError("hidden_obstacles is not part of the batched path")
    p.max_time_steps = int(c.get("max_time_steps", 150))
    p.multiple_tasks_per_agent = int(bool(c.get("multiple_tasks_per_agent", False)))
    p.random_init_pos = int(bool(c.get("random_init_pos", False)))
    p.num_obstacles = int(c.get("num_obstacles", 0) or 0)
    p.simulation_frame_rate = float(c.get("simulation_frame_rate", 0.01))
    p.fail_rate = float(c.get("fail_rate", 0.0))
    p.early_terminate = int(bool(c.get("early_terminate", False)))
    p.capability_mask = int(bool(c.get("capability_mask", False)))
    p.saturate_mask = int(bool(c.get("saturate_mask", False)))
    rw = c.get("reward_weights", None) or {
        "action": 0.0, "distance": 1.0, "quality": 1.0, "s_quality": 1.0, "time": 0.0, "alloc": 0.0,
        "time_penaulty": 0.0, "step": 0.0,
    }
    for i, k in enumerate(REWARD_KEYS):
        p.reward_weights[i] = float(rw[k])
    p.arrival_rate = float(c.get("arrival_rate", 0.0) or 0.0)
    p.include_time_windows = int(bool(c.get("include_time_windows", False)))
    p.dynamic_idle_penalty = float(c.get("dynamic_idle_penalty", 0.0) or 0.0)
    p.sense_radius = float(c.get("sense_radius", 0.0) or 0.0)
    p.threat_delay = int(c.get("threat_delay", 0) or 0)
    p.hard_windows = int(bool(c.get("hard_windows", False)))
    p.window_length = int(c.get("window_length", 30) or 30)
    p.burst_mode = int(bool(c.get("burst_mode", False)))
    p.burst_size = int(c.get("burst_size", 3) or 3)
    p.miss_penalty = float(c.get("miss_penalty", 25.0) or 0.0)
    p.on_time_bonus = float(c.get("on_time_bonus", 10.0) or 0.0)
    p.dual_region_bursts = int(bool(c.get("dual_region_bursts", False)))
    p.share_knowledge = int(bool(c.get("share_knowledge", True)))
    p.commit_horizon = int(c.get("commit_horizon", 0) or 0)
    p.reassign_penalty = float(c.get("reassign_penalty", 0.0) or 0.0)
    p.escort_enabled = int(bool(c.get("escort_enabled", False)))
    p.escort_radius = float(c.get("escort_radius", 70.0) or 70.0)
    p.escort_requirement = float(c.get("escort_requirement", 1.2) or 1.2)
    p.escort_intercept_radius = float(c.get("escort_intercept_radius", 100.0) or 100.0)
    p.mutual_support_radius = float(c.get("mutual_support_radius", 80.0) or 80.0)
    mask = 0
    for t in tuple(c.get("escort_agent_types", ("F1", "F2")) or ("F1", "F2")):
        mask |= 1 << UAV_TYPES.index(t)
    p.escort_agent_type_mask = mask
    n_threats = sum(p.threat_count[: p.n_threat_groups])
    p.tile_agents = int(tile_agents) or max(16, _round_up(p.n_agents, 8))
    p.tile_tasks = int(tile_tasks) or max(40, _round_up(p.n_tasks + p.n_threat_groups + n_threats + 8, 16))
    p.tile_threats = int(tile_threats) or max(16, _round_up(n_threats, 8))
    if p.n_agents > p.tile_agents or n_threats > p.tile_threats:
        raise ValueError("tile too small for this fleet / threat list")
    return p


def params_for_case(case: str, **over) -> MuavtaParams:
    """Params for a registry case under the WPS harness flags (experiments/wps_eval.py:91-98)."""
    from .scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS

    ta, tt, th = TILES.get(case, (0, 0, 0))
    return params_from_config(CASE_SPECS[case], dict(WPS_ENV_FLAGS), tile_agents=over.pop("tile_agents", ta),
                              tile_tasks=over.pop("tile_tasks", tt), tile_threats=over.pop("tile_threats", th))
