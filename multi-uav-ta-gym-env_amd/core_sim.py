"""Compat module for the reference's PyO3 helper: `from muavta_amd import core_sim` in place of `import core_sim`
(core_sim/src/lib.rs, sim_core.rs:25-59; called at mUAV_TA/DroneEnv.py:1033,1047,1120 as
`self.sim_core.avoid_obstacles(list(pos), [[x, y, size], ...], list(movement)) -> [dx, dy]`).
Backed by the same device function the batched env uses (`muavta_avoid_obstacles`): it needs the HIP library and a
GPU like everything else on this path.  `avoid_obstacles_batch` takes n (position, movement) pairs per call."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .batched import avoid_obstacles as _avoid


class SimCore:
    def __init__(self, device: int = 0):
        self.device = int(device)

    def avoid_obstacles(self, agent_pos: Sequence[float], obstacles: Sequence[Sequence[float]], movement: Sequence[float]) -> List[float]:
        if len(obstacles) == 0:
            return [0.0, 0.0]  # sim_core.rs:25-59 with K = 0: the zero vector
        out = _avoid(np.asarray(agent_pos, dtype=np.float64)[None], obstacles, np.asarray(movement, dtype=np.float64)[None], self.device)
        return [float(out[0, 0]), float(out[0, 1])]

    def avoid_obstacles_batch(self, agent_pos, obstacles, movement) -> np.ndarray:
        return _avoid(agent_pos, obstacles, movement, self.device)
