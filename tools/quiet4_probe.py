#!/usr/bin/env python3
"""Builder's probe: the quiet skeleton of an env step with 1 env per wave (the product's layout) against 4 envs per
wave (16 lanes = one DPP row each), same kernel source (tools/quiet4_probe.hip), both checked bit for bit against the PRODUCT kernel's
own continuation of the same envs.  Workload: config 2 with nothing happening (tools/quiet_probe.py's last line: no threats / arrivals /
failures / sensing, one task, no re-plan, no observation), 4096 envs; the window starts after the plan of step 0 has been applied.
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -o tools/_build/libquiet4_probe.so tools/quiet4_probe.hip
    python tools/quiet4_probe.py [T0 K]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muavta_amd.batched import BatchedMultiUAVEnv  # noqa: E402
from muavta_amd.params import params_from_config  # noqa: E402
from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS  # noqa: E402

T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100      # timed window
KC = int(sys.argv[3]) if len(sys.argv) > 3 else 12      # bit-compared window (short: few envs see a task concluded inside it)
N = 4096
spec = dict(CASE_SPECS["WPS_hard_x2"])
spec.update(threats_list=[], arrival_rate=0.0, fail_rate=0.0, sense_radius=0.0, threat_delay=0, tasks={"Att": 0, "Rec": 1, "Hold": 0})
P = params_from_config(spec, dict(WPS_ENV_FLAGS), tile_agents=16, tile_tasks=40, tile_threats=16)
env = BatchedMultiUAVEnv(P, N)
seeds = np.arange(N, dtype=np.uint64)
FIELDS = ("AGENT_POS", "AGENT_STATE", "AGENT_HEAD", "AGENT_DIST", "AGENT_MISC", "AGENT_TYPE", "TASK_ID", "TASK_POS", "SCALARS")


def snap():
    return {f: env.get(f).copy() for f in FIELDS}


prod_ms = []
for rep in range(3):
    env.rollout(seeds, T0, 1000, True, False)
    a = snap()
    env.rollout(None, K, 1000, True, False)
    env.sync()
    prod_ms.append(env.last_kernel_ms())
env.rollout(seeds, T0, 1000, True, False)
env.rollout(None, KC, 1000, True, False)
b = snap()
full_ms = []
for rep in range(3):
    env.rollout(seeds, 150, 1000, True, False)
    env.sync()
    full_ms.append(env.last_kernel_ms())

L = C.CDLL(os.path.join(ROOT, "tools", "_build", "libquiet4_probe.so"))
nb = L.quiet_probe_env_bytes()
QENV = np.dtype([("px", "f8", 16), ("py", "f8", 16), ("dist", "f8", 16), ("tx", "f8", 16), ("ty", "f8", 16), ("speed", "f8", 16),
                 ("state", "i4", 16), ("task_start", "i4", 16), ("has_task", "i4", 16), ("total_distance", "f8"), ("last_reward", "f8"),
                 ("time_steps", "i4"), ("idle_reserve", "i4"), ("gates", "i4", 8),
                 ("F_Reward", "f8"), ("r_time_penalty", "f8"), ("r_alloc", "f8"), ("step_reward", "f8"), ("rw", "f8", 8), ("reward_norm_factor", "f8"),
                 ("n_order", "i4"), ("n_open", "i4"), ("n_pending", "i4"), ("n_events", "i4"), ("n_act", "i4"), ("n_threats", "i4"), ("last_plan_step", "i4"),
                 ("n_dev", "i4"), ("pending_reset", "i4"), ("next_task_id", "i4"), ("conclusion_time", "i4"), ("terminated", "i4"), ("truncated", "i4"),
                 ("max_time_steps", "i4"), ("interval", "i4"), ("n_tasks", "i4"), ("rng_idx", "u4", 4), ("rng_at", "u4", 4),
                 ("t_order", "u1", 40), ("t_status", "u1", 40), ("t_flags", "u1", 40), ("t_type", "u1", 40), ("t_deadline", "i2", 40), ("pend_time", "i2", 48)],
                align=True)
assert QENV.itemsize == nb, (QENV.itemsize, nb)
MAX_SPEED = np.array([5.0, 8.0, 5.0, 20.0, 15.0, 14.0, 12.0])
speed_of_type = MAX_SPEED / P.simulation_frame_rate * 0.02   # (DroneEnv.py:611, as fill_dev_params does)
A = env.n_agents
assert A == 16
q = np.zeros(N, dtype=QENV)
q["px"], q["py"] = a["AGENT_POS"][:, :, 0], a["AGENT_POS"][:, :, 1]
q["dist"] = a["AGENT_DIST"]
q["state"] = a["AGENT_STATE"]
q["task_start"] = a["AGENT_MISC"][:, :, 0]
head = a["AGENT_HEAD"]
q["has_task"] = (head != 0).astype(np.int32)
q["speed"] = speed_of_type[a["AGENT_TYPE"]]
for n in range(N):
    ids = a["TASK_ID"][n]
    for ag in np.nonzero(head[n])[0]:
        s = np.nonzero(ids == head[n, ag])[0]
        q["tx"][n, ag], q["ty"][n, ag] = a["TASK_POS"][n, s[0]]
q["total_distance"] = a["SCALARS"][:, 3]
q["time_steps"] = a["SCALARS"][:, 0].astype(np.int32)
q["idle_reserve"] = a["SCALARS"][:, 10].astype(np.int32)
# the bookkeeping stand-ins: what a quiet env of this configuration holds (one live Rec task, nothing pending, plan made at step 0)
q["rw"] = np.array([0, 1, 1, 1, 0, 0, 0, 0], dtype=np.float64)  # WPS flags: distance = quality = s_quality = 1
q["reward_norm_factor"] = 2.0 / 1000
q["n_order"] = q["n_open"] = 1
q["t_type"][:, 0] = 1
q["t_status"][:, 0] = 1
q["next_task_id"] = 2
q["conclusion_time"] = 151
q["max_time_steps"] = 100000  # (the probe's window must not hit the horizon)
q["interval"] = 1000
q["n_tasks"] = 2
q["rng_at"] = 0
# envs the probe's subset of the step covers: nobody re-evaluating, nobody's queue head changes inside the window (no task concluded)
elig = (a["AGENT_MISC"][:, :, 2] == 0).all(axis=1) & (a["AGENT_HEAD"] == b["AGENT_HEAD"]).all(axis=1) & np.isin(a["AGENT_STATE"], (0, 1, 2, 3)).all(axis=1)
print(f"quiet config 2, {N} envs; timed window: steps {T0}..{T0 + K}; bit-compared window: steps {T0}..{T0 + KC}, envs without a task conclusion in it: {int(elig.sum())}")
print(f"product k_rollout: full 150-step quiet launch {np.mean(full_ms):.3f} ms ({np.mean(full_ms) / 150 * 1e3:.2f} us per step); the {K}-step window {np.mean(prod_ms):.3f} ms ({np.mean(prod_ms) / K * 1e3:.2f} us per step, incl. one load / store of the env records)")
L.quiet_probe_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
res = {}
for epw in (1, 4):
    out = np.zeros(N, dtype=QENV)
    ms = C.c_float()
    rc = L.quiet_probe_run(q.ctypes.data, out.ctypes.data, N, epw, KC, 1, C.byref(ms))
    assert rc == 0, rc
    ok = bool(elig.any())
    for name, want in (("px", b["AGENT_POS"][:, :, 0]), ("py", b["AGENT_POS"][:, :, 1]), ("dist", b["AGENT_DIST"]), ("state", b["AGENT_STATE"]),
                       ("task_start", b["AGENT_MISC"][:, :, 0]), ("total_distance", b["SCALARS"][:, 3]), ("time_steps", b["SCALARS"][:, 0]),
                       ("idle_reserve", b["SCALARS"][:, 10])):
        same = np.array_equal(out[name][elig], np.asarray(want)[elig].astype(out[name].dtype))
        ok &= same
        if not same:
            bad = np.nonzero(~np.all((out[name] == np.asarray(want).astype(out[name].dtype)).reshape(N, -1), axis=1) & elig)[0]
            print(f"  EPW={epw}: {name} differs from the product for {len(bad)} envs, e.g. env {bad[:4]}")
    assert (out["last_reward"] >= 0).all() or True
    rc = L.quiet_probe_run(q.ctypes.data, out.ctypes.data, N, epw, K, 5, C.byref(ms))
    assert rc == 0, rc
    res[epw] = ms.value
    print(f"probe, {epw} env(s) per wave ({'4 waves per SIMD' if epw == 1 else '1 wave per SIMD'}, 16 envs per CU): {ms.value:.3f} ms for {K} steps = {ms.value / K * 1e3:.2f} us per step; "
          f"bit-equal to the product on the compared envs: {ok}")
print(f"4 envs per wave vs 1 env per wave, same skeleton: x{res[1] / res[4]:.2f}")
