"""Diagnostic: where the one-launch-per-env-step path (muavta_rollout(h, NULL, 1, ...)) spends its time.
Prints per-launch kernel durations (HIP events on the handle's stream) over one episode batch next to the fused rollout's
kernel time, and the wall-clock gap between queued launches.  Usage: python tools/step_launch_probe.py [case] [envs]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case

case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
interval = 12 if "escort" in case else 20
env = BatchedMultiUAVEnv(params_for_case(case), n, device=0)
seeds = np.arange(n, dtype=np.uint64)
env.rollout(seeds, 150, interval, True, True); env.sync()
fused = env.last_kernel_ms()
for rep in range(2):
    env.reset(seeds); env.sync()
    ms = []
    for t in range(150):
        env.rollout(None, 1, interval, True, True)
        ms.append(env.last_kernel_ms())
ms = np.array(ms)
env.reset(seeds); env.sync()
t0 = time.perf_counter()
for t in range(150):
    env.rollout(None, 1, interval, True, True)
t_queue = time.perf_counter() - t0
env.sync()
t_all = time.perf_counter() - t0
print(f"{case} {n} envs: fused 150-step kernel {fused:.3f} ms")
print(f"1-step launches: sum of kernel times {ms.sum():.3f} ms, mean {ms.mean()*1e3:.1f} us, min {ms.min()*1e3:.1f} us, max {ms.max()*1e3:.1f} us")
print("replan steps (t % interval == 0):", np.round(ms[::interval] * 1e3, 1), "us")
print("other steps, mean: %.1f us" % (np.delete(ms, np.arange(0, 150, interval)).mean() * 1e3))
print(f"queueing 150 launches from Python took {t_queue*1e3:.2f} ms ({t_queue/150*1e6:.1f} us per call); until drained {t_all*1e3:.2f} ms "
      f"-> {n*150/t_all/1e6:.1f} M env-steps/s")
