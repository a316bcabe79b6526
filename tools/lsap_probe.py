#!/usr/bin/env python3
"""Micro-benchmark of the stand-alone LSAP kernels (k_lsap): n problems of one shape per launch, one wave per problem.
Run under `rocprofv3 --kernel-trace --stats` — the kernel durations are the figure (the C ABI call also copies over PCIe).
usage: lsap_probe.py [n_problems]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muavta_amd.batched import lsap

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
for (nr, nc, impl) in ((7, 35, "registers"), (21, 60, "registers"), (16, 23, "registers"), (7, 35, "lds"), (21, 60, "lds")):
    c = rng.uniform(0.0, 1.0, size=(n, nr, nc))
    c[rng.uniform(size=c.shape) < 0.2] = 1e6
    lsap(c, impl=impl)
    t0 = time.perf_counter()
    for _ in range(3):
        lsap(c, impl=impl)
    print(f"{impl:9s} {nr}x{nc} n={n}: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per call (incl. copies)")
