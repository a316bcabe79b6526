#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_rollout from in-kernel s_memtime stamps (lane 0).
Builds a SEPARATE library with -DMUAVTA_PROF (never the shipped one); shares are meaningful, the
absolute run time of this build is not (guide: 'In-kernel stamps')."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muavta_amd import native
so = os.path.join(ROOT, "tools", "_build", "libmuavta_prof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + native.HIPCC_FLAGS + ["-DMUAVTA_PROF"] + [a for a in sys.argv if a.startswith("-D")] + ["-o", so, os.path.join(native.CSRC, "muavta_kernels.hip")])
    sys.exit(0)
native.SO_PATH = so
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
interval = 12 if "escort" in case else 20
obs = True
if case == "QUIET":  # tools/quiet_probe.py's last line: config 2 with nothing happening (no threats / arrivals / failures / sensing, one task, no re-plan, no observation)
    from muavta_amd.params import params_from_config
    from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS
    spec = dict(CASE_SPECS["WPS_hard_x2"]); spec.update(threats_list=[], arrival_rate=0.0, fail_rate=0.0, sense_radius=0.0, threat_delay=0, tasks={"Att": 0, "Rec": 1, "Hold": 0})
    env = BatchedMultiUAVEnv(params_from_config(spec, dict(WPS_ENV_FLAGS), tile_agents=16, tile_tasks=40, tile_threats=16), n)
    interval, obs = 1000, False
else:
    env = BatchedMultiUAVEnv(params_for_case(case), n)
L = native.lib()
buf = (C.c_ulonglong * 64)()
env.rollout(np.arange(n, dtype=np.uint64), 150, interval, True, obs); env.sync()
L.muavta_prof_read(buf, 1)
per_env = 1
if "--slowest" in sys.argv:  # (build with --build -DMUAVTA_DIAG_TIMES) only the env whose wave ended last: what the launch's critical path spends its time on
    times = np.zeros((3, n), dtype=np.uint32)
    L.muavta_diag_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    assert L.muavta_diag_times(env.h, times.ctypes.data_as(C.c_void_p), n) == 0
    dur = (times[1] - times[0]).astype(np.int64)
    slow = int(np.argmax(dur))
    print(f"slowest env {slow}: {dur[slow] * 1e-5:.3f} ms; mean env {dur.mean() * 1e-5:.3f} ms")
    L.muavta_prof_target(slow)
    per_env = n  # the sums below are of ONE env: undo the division by n
env.rollout(np.arange(n, dtype=np.uint64), 150, interval, True, obs); env.sync()
L.muavta_prof_read(buf, 1)
names = {0: "(loop gap)", 1: "rng_refill", 2: "drain+release", 3: "actions", 4: "movement", 5: "dist", 6: "serial_b threats/arrivals/escorts",
         7: "sense", 8: "serial_c reveals/expire/reward", 9: "finish gc+open", 10: "(pre-alloc)", 11: "alloc gate", 12: "cost build",
         13: "lsap", 14: "accept", 15: "(pre-obs)", 16: "obs rows->LDS", 17: "obs rows stream", 18: "obs legal->LDS", 19: "obs legal stream", 20: "obs agents/flags/result", 21: "b: np.sum + penalty terms", 22: "b: generate_threat", 23: "b: threats parallel", 24: "b: threats serial replay", 25: "b: arrivals", 6: "b: escorts+sync", 26: "move: parallel pass", 4: "move: serial replay", 28: "c: prechecks", 29: "c: lists", 8: "c: serial_c (reward)", 32: "c: ballots", 33: "actions: precompute", 3: "actions: serial_a", 34: "move: compute (in 26)", 35: "finish: gc loop (slow path)", 36: "c: serial_c head", 37: "c: serial_c weighted sum", 38: "c: serial_c divisions", 30: "(count x1000) end-of-step slow path entered", 31: "(count x1000) ... with retired slots present", 40: "(count x1000) movement passes", 41: "(count x1000) movement events", 39: "reset: (entry)", 42: "reset: master init_by_array", 43: "reset: agent stream twist+tape", 44: "reset: seed draws + 2-3 init_by_array (parallel lanes)", 45: "reset: tgt/mission streams twist+tape", 46: "reset: zero blob", 47: "reset: serial entity creation"}
v = np.array(list(buf), dtype=np.float64) * per_env
counts = v[48:].copy()
extra = {59: "escorts: arrivals + creation (in b: escorts+sync)", 60: "escorts: segment passes", 61: "escorts: retirements"}
v = v[:48]
for i in (30, 31, 40, 41):  # event counters that live among the cycle slots
    counts = np.append(counts, v[i]); v[i] = 0
v[10] = 0  # (pre-alloc) only holds the first stamp's absolute clock
tot = v.sum()
print(f"{case} n={n}: kernel {env.last_kernel_ms():.2f} ms; cycles/env-step (lane 0) {tot / n / 150:.0f} (each stamp costs ~500 cycles of s_memtime latency)")
for i in range(48):
    if v[i] == 0: continue
    print(f"  {str(names.get(i, i)):40s} {100 * v[i] / tot:6.2f} %   {v[i] / n / 150:9.0f} cyc/step")
cnames = ["LSAP solves", "LSAP scan steps (inner iterations)", "LSAP rows (sum)", "LSAP columns (sum)", "replans (allocate ran)", "releaseAllTasks calls",
          "threat passes", "threat events (serial replays)", "escort syncs with entries", "escort map entries (sum)", "escort sync loop iterations", "", "", "", "cost-build rows (sum over rounds)", "... of rounds that turned out infeasible (no solve)",
          "end-of-step slow path entered", "... with retired slots present", "movement passes", "movement events"]
for i, nm in extra.items():
    if counts[i - 48]: print(f"  {nm:40s} {100 * counts[i - 48] / tot:6.2f} %   {counts[i - 48] / n / 150:9.0f} cyc/step   (cycles; part of the slot named in brackets)"); counts[i - 48] = 0
print("  per env-step event counts:")
for nm, c in zip(cnames, counts):
    if nm and c: print(f"    {nm:40s} {c / 1000 / n / 150:8.3f}")
