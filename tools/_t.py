import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
for n in (1024, 4096):
    env = BatchedMultiUAVEnv(params_for_case("WPS_hard_x2"), n)
    seeds = np.arange(n, dtype=np.uint64)
    for steps in (0, 150):
        ts = []
        for r in range(4):
            env.rollout(seeds, steps, 20, True, True); ts.append(env.last_kernel_ms())
        print(n, steps, "ms", min(ts))
