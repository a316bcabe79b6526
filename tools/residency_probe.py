#!/usr/bin/env python3
"""Diagnostic: k_rollout time against the number of envs per launch (= resident waves per CU), headline workload."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
interval = 12 if "escort" in case else 20
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 1024, 2048, 3072, 4096, 8192]
for n in sizes:
    env = BatchedMultiUAVEnv(params_for_case(case), n)
    seeds = np.arange(n, dtype=np.uint64)
    for obs in (True, False):
        env.rollout(seeds, 150, interval, True, obs); env.sync()
        ms = []
        for _ in range(4):
            env.rollout(seeds, 150, interval, True, obs); ms.append(env.last_kernel_ms())
        print(f"{case} {n:6d} envs ({n / 256:5.1f} per CU) obs={int(obs)}: {np.mean(ms):7.3f} ms  {n * 150 / np.mean(ms) / 1e3:7.1f} M env-steps/s", flush=True)
    env.close()
