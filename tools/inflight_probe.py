#!/usr/bin/env python3
"""Two (or more) handles in flight (builder's probe): handle i owns its own stream and blobs; launches alternate A, B, A, B ... so that
batch i+1's workgroups take the wave slots batch i's early finishers free.  usage: inflight_probe.py [case envs interval launches]"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muavta_amd.batched import BatchedMultiUAVEnv  # noqa: E402
from muavta_amd.params import params_for_case  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    interval = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    launches = int(sys.argv[4]) if len(sys.argv) > 4 else 24
    for k in (1, 2, 3):
        envs = [BatchedMultiUAVEnv(params_for_case(case), n) for _ in range(k)]
        seeds = [np.arange(i * n, (i + 1) * n, dtype=np.uint64) for i in range(k)]
        for _ in range(3):
            for e, s in zip(envs, seeds):
                e.rollout(s, 150, interval, True, True)
        for e in envs:
            e.sync()
        t0 = time.perf_counter()
        for i in range(launches):
            envs[i % k].rollout(seeds[i % k], 150, interval, True, True)
        for e in envs:
            e.sync()
        dt = time.perf_counter() - t0
        ms = np.mean([e.kernel_ms_history(min(8, launches // k)).mean() for e in envs])
        flagged = sum(int(np.count_nonzero(e.get("ERROR"))) for e in envs)
        print(f"{case} {n} envs x {k} in flight: {n * 150 * launches / dt / 1e6:.1f} M env-steps/s, {dt / launches * 1e3:.3f} ms per launch (wall), kernel {ms:.3f} ms each, flagged {flagged}", flush=True)
        for e in envs:
            e.close()


if __name__ == "__main__":
    main()
