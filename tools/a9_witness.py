#!/usr/bin/env python3
"""Builder's tool (container or GPU box): the K > 0 obstacle path against the arbitrary-precision witness of tests/sim_core_witness.py.
Prints how many of the oracle's results equal the witness bit for bit, and every logarithm / atan2 argument where the host libm is not
correctly rounded.   usage: a9_witness.py [n_pairs]"""
import ctypes as C
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from sim_core_witness import avoid_cr, log_cr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(3)
obst = np.array([[300.0, 300.0, 50.0], [700.0, 200.0, 80.0], [500.0, 500.0, 30.0]])
pos = rng.uniform(100, 900, (n, 2)); mov = rng.uniform(-1, 1, (n, 2))
# half of the positions inside some obstacle's 40-unit zone, so that the branch is actually taken
k = n // 2
which = rng.integers(0, 3, k); ang = rng.uniform(0, 2 * math.pi, k); rad = obst[which, 2] + rng.uniform(0.2, 39.9, k)
pos[:k, 0] = obst[which, 0] + rad * np.cos(ang); pos[:k, 1] = obst[which, 1] + rad * np.sin(ang)
L = orc.lib()
same = in_zone = log_args = 0
bad_log, bad_atan, unexplained = [], 0, []
for i in range(n):
    out = np.zeros(2)
    L.orc_avoid_obstacles(obst.ctypes.data_as(C.c_void_p), 3, pos[i].ctypes.data_as(C.c_void_p), mov[i].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    want, info = avoid_cr(pos[i].tolist(), obst.tolist(), mov[i].tolist())
    in_zone += bool(info)
    log_args += len(info)
    ok = out[0] == want[0] and out[1] == want[1]
    same += ok
    cr = all(r["libm_log_cr"] for r in info)
    for r in info:
        if not r["libm_log_cr"]:
            bad_log.append(r["log_arg"])
        bad_atan += not r["libm_atan2_cr"]
    if not ok and cr:
        unexplained.append((i, out.tolist(), want))
print(f"{n} (position, movement) pairs x 3 obstacles; {in_zone} pairs inside at least one zone ({log_args} logarithms taken)")
print(f"oracle == arbitrary-precision witness, bit for bit: {same} of {n}")
print(f"logarithm arguments where the host libm's log is not correctly rounded: {len(bad_log)} of {log_args}" + (f" (first: {[float.hex(a) for a in bad_log[:6]]})" if bad_log else ""))
print(f"atan2 calls where the host libm is not correctly rounded: {bad_atan} (only the SIGN of the wrapped difference is used)")
print(f"differences NOT explained by a non-correctly-rounded libm log: {len(unexplained)}")
for u in unexplained[:5]:
    print("   ", u)
