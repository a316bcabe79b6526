"""Import harness for the read-only reference checkout (THIS container only).

Test tooling, never product code: pre-seeds ``sys.modules`` with inert stand-ins for the
third-party packages the reference imports but this image lacks (gymnasium, pettingzoo,
seaborn, tianshou) and with a Python restatement of the 60-line Rust helper
``core_sim.SimCore.avoid_obstacles`` (core_sim/src/sim_core.rs:25-59; Rust ``%`` is a
truncated remainder, hence ``math.fmod``).  Nothing under /root/reference is written,
copied or byte-compiled.  Used only by ``tools/gen_golden.py`` to capture golden vectors.
"""
import math
import os
import sys
import types

REF = os.environ.get("MUAVTA_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Space:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k
        self.shape = k.get("shape")

    def __getitem__(self, k):
        return self.args[0][k]


class _ParallelEnv:
    def __init__(self, *a, **k):
        pass

    def close(self):  # pettingzoo.ParallelEnv.close is a no-op the env inherits (main.py:273 calls it)
        pass


class _AgentSelector:
    """Cyclic iterator used by DroneEnv.py:142-143,597-598,754,787."""

    def __init__(self, order):
        self.reinit(order)

    def reinit(self, order):
        self.agent_order = list(order)
        self._current_agent = 0
        self.selected_agent = 0

    def reset(self):
        self.reinit(self.agent_order)
        return self.next()

    def next(self):
        self._current_agent = (self._current_agent + 1) % len(self.agent_order)
        self.selected_agent = self.agent_order[self._current_agent - 1]
        return self.selected_agent


class _SimCore:
    def __init__(self, max_time_steps=0):
        self.time_steps, self.max_time_steps = 0, max_time_steps

    @staticmethod
    def avoid_obstacles(agent_pos, obstacles, movement):
        ax = ay = 0.0
        for ox, oy, osz in obstacles:
            dx, dy = ox - agent_pos[0], oy - agent_pos[1]
            d_zone = math.sqrt(dx * dx + dy * dy) - osz
            if d_zone < 40.0:
                nx, ny = dx / d_zone, dy / d_zone
                force = 0.5 / (1.0 - math.log(max(1.05, d_zone)))
                ang = math.atan2(movement[1], movement[0]) - math.atan2(dy, dx)
                ang = math.fmod(ang + math.pi, 2.0 * math.pi) - math.pi
                rx, ry = (ny, -nx) if ang > 0.0 else (-ny, nx)
                ax += rx * force
                ay += ry * force
        return [ax, ay]


def install():
    if "core_sim" in sys.modules and getattr(sys.modules["core_sim"], "_muavta_shim", False):
        return
    sys.dont_write_bytecode = True
    spaces = _mod("gymnasium.spaces", Dict=_Space, Box=_Space, Discrete=_Space, MultiDiscrete=_Space)
    _mod("gymnasium", spaces=spaces)
    wrappers = _mod("pettingzoo.utils.wrappers", OrderEnforcingWrapper=lambda e: e)
    sel = _mod("pettingzoo.utils.agent_selector", agent_selector=_AgentSelector)
    utils = _mod("pettingzoo.utils", parallel_to_aec=lambda e: e, wrappers=wrappers, agent_selector=sel)
    _mod("pettingzoo", ParallelEnv=_ParallelEnv, utils=utils)
    _mod("seaborn")
    data = _mod("tianshou.data", Batch=dict)
    _mod("tianshou", data=data)
    _mod("TaskAllocation.RL_Policies.Tianshou_Policy", _get_model=lambda *a, **k: None)
    _mod("core_sim", SimCore=_SimCore, _muavta_shim=True)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def available():
    return os.path.isdir(os.path.join(REF, "mUAV_TA"))
