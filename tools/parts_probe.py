#!/usr/bin/env python3
"""Per-step path with sub-batches: host enqueue time vs total time for 150 env steps, by number of parts."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_TORCH_FIRST"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
from muavta_amd.batched import BatchedMultiUAVEnv
if os.environ.get("PROBE_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize(); _x = torch.zeros(8, device="cuda")
from muavta_amd.params import params_for_case
case, n = "WPS_hard_x2", 4096
env = BatchedMultiUAVEnv(params_for_case(case), n)
seeds = np.arange(n, dtype=np.uint64)
PRE = os.environ.get("PROBE_PRELUDE", "").split(",")
if "rollouts" in PRE:
    for _ in range(int(os.environ.get("PROBE_ROLLOUTS", "10"))):
        env.rollout(seeds, 150, 20, True, True)
    env.sync()
if "history" in PRE:
    print("history", env.kernel_ms_history(4), env.last_seed_ms(), env.last_kernel_ms())
if "stepapi" in PRE:
    env.reset(seeds)
    for _ in range(150):
        env.allocate(20, True, fetch=False); env.step_staged()
    env.sync()
if "get" in PRE:
    env.reset(seeds); env.get("ERROR"); env.rollout_metrics()
if "torchops" in PRE:
    import torch
    torch.cuda.synchronize(); t = torch.tensor([1.0, 2.0], dtype=torch.float64, device="cuda"); print(float(t[0]))
for parts in (0, 2, 3, 4):
    env.set_parts(parts)
    for rep in range(2):
        env.reset(seeds); env.sync()
        t0 = time.perf_counter()
        for _ in range(150):
            if parts:
                for p in range(parts):
                    env.rollout_part(p, 1, 20, True, True)
            else:
                env.rollout(None, 1, 20, True, True)
        t1 = time.perf_counter()
        env.sync()
        t2 = time.perf_counter()
    print(f"parts {parts}: host enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms -> {n * 150 / (t2 - t0) / 1e6:.1f} M env-steps/s")
