#!/usr/bin/env python3
"""Diagnostic (build with -DMUAVTA_DIAG_TIMES into tools/_build/libmuavta_times.so): when do the waves of one k_rollout launch
end?  Prints the distribution of per-env end times and, per SIMD, when its last wave ends — the launch ends on the slowest
SIMD, so the gap between the mean and the max SIMD is what placement / pacing could still recover."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MUAVTA_SO", os.path.join(ROOT, "tools", "_build", "libmuavta_times.so"))
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
interval = 12 if "escort" in case else 20
env = BatchedMultiUAVEnv(params_for_case(case), n, device=0)
seeds = np.arange(n, dtype=np.uint64)
for _ in range(3):
    env.rollout(seeds, 150, interval, True, True)
env.sync()
out = np.zeros((3, n), dtype=np.uint32)
env.L.muavta_diag_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
assert env.L.muavta_diag_times(env.h, out.ctypes.data_as(C.c_void_p), n) == 0
t0 = out[0].min()
start, end, hw = (out[0] - t0) * 1e-5, (out[1] - t0) * 1e-5, out[2]   # ms
simd = (hw >> 4) & 0xFFF | ((hw >> 16) << 12)
print(f"{case} {n} envs: kernel {env.last_kernel_ms():.3f} ms; mean makespan metric (conclusion_time: max_time_steps + 1 when the mission never concluded; every env ran all 150 steps) {env.rollout_metrics()[:, 8].mean():.1f}")
print("wave start  ms: min %.3f  median %.3f  max %.3f" % (start.min(), np.median(start), start.max()))
q = np.percentile(end, [0, 5, 25, 50, 75, 95, 99, 100])
print("wave end    ms: min %.3f  p5 %.3f  p25 %.3f  median %.3f  p75 %.3f  p95 %.3f  p99 %.3f  max %.3f" % tuple(q))
ids = np.unique(simd)
last = np.array([end[simd == s].max() for s in ids]); cnt = np.array([(simd == s).sum() for s in ids])
first = np.array([end[simd == s].min() for s in ids])
print(f"SIMDs seen: {len(ids)}, waves per SIMD: min {cnt.min()} max {cnt.max()}")
q = np.percentile(last, [0, 5, 25, 50, 75, 95, 100])
print("SIMD's last wave ends  ms: min %.3f  p5 %.3f  p25 %.3f  median %.3f  p75 %.3f  p95 %.3f  max %.3f   mean %.3f" % (tuple(q) + (last.mean(),)))
print("spread inside a SIMD (last - first wave end) ms: mean %.3f  p95 %.3f  max %.3f" % ((last - first).mean(), np.percentile(last - first, 95), (last - first).max()))
print("mean busy fraction of wave slots until the launch ends: %.3f" % ((end - start).sum() / (n * end.max())))
