"""Workload for PC sampling: fused rollouts only.  Usage: python3 tools/pcs_driver.py [case] [envs] [reps]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case

case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
interval = 12 if "escort" in case else 20
env = BatchedMultiUAVEnv(params_for_case(case), n, device=0)
seeds = np.arange(n, dtype=np.uint64)
for _ in range(reps):
    env.rollout(seeds, 150, interval, True, True)
env.sync()
print("done", env.last_kernel_ms())
