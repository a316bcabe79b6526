#!/usr/bin/env python3
"""Secondary throughput figures (SURVEY §8d configs 4 and 5, and the registry WPS_hard): not the headline bench."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
for case, n, interval in (("WPS_hard_x2", 4096, 20), ("WPS_hard", 4096, 20), ("WPS_escort24", 4096, 12), ("WPS_escort", 4096, 12), ("WPS_burst64", 1024, 20)):
    env = BatchedMultiUAVEnv(params_for_case(case), n)
    seeds = np.arange(n, dtype=np.uint64)
    env.rollout(seeds, 150, interval, True, True); env.sync()
    ms = []
    for _ in range(3):
        env.rollout(seeds, 150, interval, True, True); ms.append(env.last_kernel_ms())
    assert not env.get("ERROR").any()
    print(f"{case:14s} {n:5d} envs  {np.mean(ms):8.2f} ms/launch  {n * 150 / np.mean(ms) / 1e3:7.2f} M env-steps/s  LDS {env.dims.lds_bytes} B", flush=True)
