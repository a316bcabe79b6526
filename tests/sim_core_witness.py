"""A third witness for core_sim's avoid_obstacles with K > 0 obstacles (test infrastructure).

The reference's helper is Rust (core_sim/src/sim_core.rs:25-59) and cannot be built here, so the CPU oracle and the device are pinned on OUR
reading of those 35 lines ("parity unpinned vs the Rust").  This module evaluates the same lines a third, independent way: every IEEE-754
operation of the source (`-`, `*`, `+`, `/`, `sqrt`, `max`, `%` = fmod) is done in Python floats — correctly rounded by the language — and the two
library functions the Rust calls into the platform libm for (`f64::ln`, `f64::atan2`) are evaluated in ARBITRARY PRECISION (mpmath, 300 bits) and
rounded to nearest once.  Wherever the host libm's `log` / `atan2` are themselves correctly rounded for the arguments met — and a correctly rounded
libm is what the Rust would call too — the oracle, the device and this witness must agree bit for bit; the arguments where the libm is NOT
correctly rounded are listed (they are the only places where a Rust build on another libm could differ from the oracle, by one ulp of the force)."""
import math
from fractions import Fraction

import mpmath

mpmath.mp.prec = 300


def _to_f64(x):
    """round-to-nearest-even of an mpf to a Python float (mpmath's own float() truncates)"""
    sign, man, exp, _bc = x._mpf_
    if not man:
        return 0.0
    v = Fraction(int(man)) * (Fraction(2) ** int(exp))
    return float(-v if sign else v)  # int / int true division in CPython is correctly rounded


def log_cr(x):
    return _to_f64(mpmath.log(mpmath.mpf(x)))


def atan2_cr(y, x):
    if x == 0.0 and y == 0.0:
        return math.atan2(y, x)  # signed-zero conventions: no rounding involved
    return _to_f64(mpmath.atan2(mpmath.mpf(y), mpmath.mpf(x)))


def avoid_cr(agent_pos, obstacles, movement, log=log_cr, atan2=atan2_cr):
    """sim_core.rs:25-59 line by line.  Returns ([ax, ay], info) — info per obstacle inside the zone: the logarithm's argument, whether the
    host libm rounds it (and both atan2 calls) correctly, and the wrapped angle difference whose SIGN picks the rotation."""
    ax = ay = 0.0
    info = []
    for ox, oy, size in obstacles:
        dx = ox - agent_pos[0]
        dy = oy - agent_pos[1]
        distance_to_obstacle = math.sqrt(dx * dx + dy * dy)
        distance_to_zone = distance_to_obstacle - size
        if distance_to_zone < 40.0:
            nx, ny = dx / distance_to_zone, dy / distance_to_zone
            arg = max(1.05, distance_to_zone)
            avoid_force = log(arg)
            avoid_force = 0.5 / (1.0 - avoid_force)
            angle_mov = atan2(movement[1], movement[0])
            angle_obs = atan2(dy, dx)
            angle_between = angle_mov - angle_obs
            angle_between = math.fmod(angle_between + math.pi, 2.0 * math.pi) - math.pi  # Rust `%` on f64 is fmod (truncated), exact
            if angle_between > 0.0:
                rx, ry = ny, -nx
            else:
                rx, ry = -ny, nx
            ax += rx * avoid_force
            ay += ry * avoid_force
            info.append({"log_arg": arg, "libm_log_cr": math.log(arg) == log_cr(arg), "angle_between": angle_between,
                         "libm_atan2_cr": math.atan2(movement[1], movement[0]) == atan2_cr(movement[1], movement[0]) and math.atan2(dy, dx) == atan2_cr(dy, dx)})
    return [ax, ay], info

