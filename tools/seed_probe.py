#!/usr/bin/env python3
"""How the seeding kernel of launch i+1 overlaps launch i: per-launch k_rollout durations, k_seed's own event pair, and the wall
time per launch for launches queued back to back versus launches separated by a sync."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
interval = 12 if "escort" in case else 20
env = BatchedMultiUAVEnv(params_for_case(case), n)
seeds = np.arange(n, dtype=np.uint64)
for _ in range(3):
    env.rollout(seeds, 150, interval, True, True); env.sync()
print(case, n, "standalone: kernel %.3f ms, k_seed %.3f ms" % (env.last_kernel_ms(), env.last_seed_ms()))
for K in (1, 2, 10, 20):
    env.sync(); t0 = time.perf_counter()
    for _ in range(K):
        env.rollout(seeds, 150, interval, True, True)
    env.sync(); wall = (time.perf_counter() - t0) * 1e3
    km = env.kernel_ms_history(K)
    print(f"  {K:2d} queued: wall/launch {wall / K:.3f} ms, k_rollout mean {km.mean():.3f} (min {km.min():.3f} max {km.max():.3f}), last k_seed span {env.last_seed_ms():.3f} ms")
import json
res = []
for rep in range(12):
    env.sync(); t0 = time.perf_counter()
    for _ in range(20):
        env.rollout(seeds, 150, interval, True, True)
    env.sync(); wall = (time.perf_counter() - t0) * 1e3
    km = env.kernel_ms_history(20)
    res.append((round(wall / 20, 3), round(float(km.mean()), 3), round(env.last_seed_ms(), 3)))
print("  12 x 20 queued (wall/launch, k_rollout mean, last k_seed span):", res)
