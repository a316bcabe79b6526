"""Drop-in import aliases for code written against the reference.

The reference's callers do ``from mUAV_TA.DroneEnv import MultiUAVEnv``, ``from mUAV_TA.MultiDroneEnvUtils import
agentEnvOptions`` and (inside DroneEnv.py) ``import core_sim``.  ``install()`` registers modules of those names that
resolve to this package, so that e.g. ``experiments/wps_eval.py::run_wps_episode``, ``experiments/escort_eval.py::
run_escort_episode`` and ``experiments/test_escort.py`` run UNCHANGED on the MI355X backend:

    import muavta_amd.compat as compat
    compat.install()                      # before the first `import mUAV_TA...`
    from experiments.wps_eval import run_wps_episode

``agentEnvOptions`` below restates the reference's option object (mUAV_TA/MultiDroneEnvUtils.py:5-105: same keywords,
same defaults, same ``x or default`` coercions) — a table of constants, the only way to accept the same constructor calls.
"""
from __future__ import annotations

import sys
import types
from typing import Callable, Optional

from . import env as _env

_SAVED = {}
_BACKEND_FACTORY: Optional[Callable] = None


class agentEnvOptions:  # noqa: N801 (the reference's spelling)
    def __init__(self, render_mode="human", render_speed=-1, simulation_frame_rate=0.01, action_mode="TaskAssign",
                 simulator_module="Internal", max_time_steps=150, agents=None, tasks=None, multiple_tasks_per_agent=False,
                 multiple_agents_per_task=True, random_init_pos=False, num_obstacles=0, hidden_obstacles=False, fail_rate=0.0,
                 threats_list=None, fixed_seed=-1, info="No Info", early_terminate=False, capability_mask=False,
                 saturate_mask=False, reward_weights=None, arrival_rate=0.0, include_time_windows=False,
                 dynamic_idle_penalty=0.0, sense_radius=0.0, threat_delay=0, hard_windows=False, window_length=30,
                 burst_mode=False, burst_size=3, miss_penalty=25.0, on_time_bonus=10.0, dual_region_bursts=False,
                 share_knowledge=True, commit_horizon=0, reassign_penalty=0.0, escort_enabled=False, escort_radius=70.0,
                 escort_requirement=1.2, escort_intercept_radius=100.0, mutual_support_radius=80.0,
                 escort_agent_types=("F1", "F2")):
        self.render_mode, self.render_speed = render_mode, render_speed
        self.simulation_frame_rate, self.action_mode, self.simulator_module = simulation_frame_rate, action_mode, simulator_module
        self.max_time_steps, self.random_init_pos = max_time_steps, random_init_pos
        self.agents = {"F1": 0, "F2": 0, "R1": 1, "R2": 1} if agents is None else agents
        self.tasks = {"Att": 0, "Rec": 2, "Hold": 0} if tasks is None else tasks
        self.multiple_tasks_per_agent, self.multiple_agents_per_task = multiple_tasks_per_agent, multiple_agents_per_task
        self.num_obstacles, self.hidden_obstacles, self.fail_rate = num_obstacles, hidden_obstacles, fail_rate
        self.threats_list = [("T1", 4), ("T2", 2)] if threats_list is None else threats_list
        self.fixed_seed, self.info = fixed_seed, info
        self.early_terminate, self.capability_mask, self.saturate_mask = early_terminate, capability_mask, saturate_mask
        self.reward_weights = reward_weights or {"action": 0.0, "distance": 1.0, "quality": 1.0, "s_quality": 1.0, "time": 0.0,
                                                 "alloc": 0.0, "time_penaulty": 0.0, "step": 0.0}
        self.arrival_rate, self.include_time_windows, self.dynamic_idle_penalty = arrival_rate, include_time_windows, dynamic_idle_penalty
        self.sense_radius, self.threat_delay, self.hard_windows, self.window_length = sense_radius, threat_delay, hard_windows, window_length
        self.burst_mode, self.burst_size, self.miss_penalty, self.on_time_bonus = burst_mode, burst_size, miss_penalty, on_time_bonus
        self.dual_region_bursts, self.share_knowledge = dual_region_bursts, share_knowledge
        self.commit_horizon = int(commit_horizon or 0)
        self.reassign_penalty = float(reassign_penalty or 0.0)
        self.escort_enabled = bool(escort_enabled)
        self.escort_radius = float(escort_radius or 70.0)
        self.escort_requirement = float(escort_requirement or 1.2)
        self.escort_intercept_radius = float(escort_intercept_radius or 100.0)
        self.mutual_support_radius = float(mutual_support_radius or 80.0)
        self.escort_agent_types = tuple(escort_agent_types or ("F1", "F2"))


class MultiUAVEnv(_env.MultiUAVEnv):
    """``mUAV_TA.DroneEnv.MultiUAVEnv(config=None)``: the facade with the reference's constructor signature."""

    def __init__(self, config=None):
        if config is None:
            config = agentEnvOptions()
        super().__init__(config, backend_factory=_BACKEND_FACTORY)  # (None: the HIP library)


class SceneData:
    """mUAV_TA/MultiDroneEnvData.py:8-85 — the scene tables other modules read by name."""
    UavTypes = ["R1", "R2", "E1", "F1", "F2", "T1", "T2"]
    TaskTypes = ["Hold", "Rec", "Att", "Def", "Int", "Det"]
    GameArea = (1200, 700)
    ContactLine = 550
    Bases = [(400, 680)]
    maxSpeeds = {"F1": 20.0, "F2": 15.0, "R1": 5.0, "R2": 8.0, "E1": 5.0, "T1": 14.0, "T2": 12.0}
    engage_range = {"F1": 40.0, "F2": 30.0, "R1": 0.0, "R2": 0.0, "E1": 0.0, "T1": 35.0, "T2": 25.0}
    TaskDuration = {"Hold": 1, "Rec": 10, "Att": 5, "Def": 5, "Int": 0, "Det": 1}


def install(backend_factory: Optional[Callable] = None, with_core_sim: bool = True) -> None:
    """Register ``mUAV_TA`` / ``mUAV_TA.DroneEnv`` / ``mUAV_TA.MultiDroneEnvUtils`` / ``mUAV_TA.MultiDroneEnvData`` (and
    ``core_sim``) in ``sys.modules``.  ``backend_factory(params) -> backend`` replaces the HIP backend (the CPU tests of
    this repository inject their oracle-backed stand-in that way; the product default is the MI355X library)."""
    global _BACKEND_FACTORY
    _BACKEND_FACTORY = backend_factory
    names = ["mUAV_TA", "mUAV_TA.DroneEnv", "mUAV_TA.MultiDroneEnvUtils", "mUAV_TA.MultiDroneEnvData"] + (["core_sim"] if with_core_sim else [])
    for n in names:
        if n not in _SAVED:
            _SAVED[n] = sys.modules.get(n)
    pkg = types.ModuleType("mUAV_TA")
    pkg.__path__ = []  # a package: `from mUAV_TA.DroneEnv import ...` resolves through sys.modules
    drone = types.ModuleType("mUAV_TA.DroneEnv")
    drone.MultiUAVEnv = MultiUAVEnv
    # the module-level names callers import beside the class (DroneEnv.py:35-68; main.py:11-12): the two factory functions hand out the parallel
    # env itself — pettingzoo's order-enforcing / AEC wrappers belong to the turn-based TBTA path (out of scope)
    drone.MAX_INT, drone.EPS = sys.maxsize, 1e-12
    drone.env = drone.raw_env = lambda config=None: MultiUAVEnv(config)
    utils = types.ModuleType("mUAV_TA.MultiDroneEnvUtils")
    utils.agentEnvOptions = agentEnvOptions
    data = types.ModuleType("mUAV_TA.MultiDroneEnvData")
    data.SceneData = SceneData
    pkg.DroneEnv, pkg.MultiDroneEnvUtils, pkg.MultiDroneEnvData = drone, utils, data
    for n, m in (("mUAV_TA", pkg), ("mUAV_TA.DroneEnv", drone), ("mUAV_TA.MultiDroneEnvUtils", utils), ("mUAV_TA.MultiDroneEnvData", data)):
        m._muavta_compat = True
        sys.modules[n] = m
    if with_core_sim:
        cs = types.ModuleType("core_sim")
        cs._muavta_compat = True

        def _simcore(*a, **k):  # resolved lazily: the HIP library is only needed once somebody calls it
            from .core_sim import SimCore
            return SimCore()

        cs.SimCore = _simcore
        sys.modules["core_sim"] = cs


def uninstall() -> None:
    global _BACKEND_FACTORY
    _BACKEND_FACTORY = None
    for n, m in list(_SAVED.items()):
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m
    _SAVED.clear()
