#!/usr/bin/env python3
"""Builder's probe: the policy-in-the-loop path consulted per GATE (muavta_rl_run_device) against per STEP (muavta_rl_step_device), fixed
score tensor on the device (times the env side).  usage: run_ahead_probe.py [case] [envs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MUAVTA_EAGER_PART_STREAMS", "8")
import torch
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case

case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard_x2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
interval = 12 if "escort" in case else 20
H = 150
dev = torch.device("cuda", 0)
e = BatchedMultiUAVEnv(params_for_case(case), N, device=0)
seeds = np.arange(N, dtype=np.uint64)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
scores = ((torch.rand((N, 16, 32), generator=gen, device=dev) * 2 - 1) * 0.35).contiguous()
tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}
mk = lambda: {k: torch.empty(sh, dtype=tdt[dt], device=dev) for k, (sh, dt) in e.token_shapes("pair", 32, 16).items()}
bufs, nxt = [mk(), mk()], mk()
sel = torch.empty((N, 16, 32), dtype=torch.float32, device=dev)
rep = torch.empty((N,), dtype=torch.int32, device=dev)
sw = torch.empty((2, N), dtype=torch.float64, device=dev)
dn = torch.empty((N,), dtype=torch.uint8, device=dev)
nst = torch.empty((N,), dtype=torch.int32, device=dev)
prk = torch.empty((N,), dtype=torch.uint8, device=dev)
rs = torch.empty((N,), dtype=torch.float64, device=dev)

def per_step():
    for _ in range(2):
        e.reset(seeds); e.tokens("pair", 32, 16, out=bufs[0]); e.sync()
        t1 = time.perf_counter()
        for t in range(H):
            e.rl_step("pair", 32, 16, edge_scores=scores, gate="trainer", replan_interval=interval, selected=sel, replanned=rep, next_tok=bufs[(t + 1) & 1], s_wps=sw, done=dn)
        e.sync()
        dt = time.perf_counter() - t1
    return N * H / dt, e.metrics()

def run_ahead(max_steps, check_every, parts=0, with_next=True, with_park=True, with_sel=True, with_scores=True):
    e.set_parts(parts)
    res = None
    for _ in range(2):
        e.reset(seeds); e.tokens("pair", 32, 16, out=bufs[0]); e.sync()
        t1 = time.perf_counter()
        k = 0
        planned = 0
        while True:
            for p in ([None] if parts <= 1 else range(parts)):
                e.rl_run("pair", 32, 16, edge_scores=scores if with_scores else None, gate="trainer", replan_interval=interval, selected=sel if with_sel else None, replanned=rep, next_tok=nxt if with_next else None, s_wps=sw, done=dn,
                         park_tok=bufs[(k + 1) & 1] if with_park else None, n_stepped=nst, park=prk, reward_sum=rs, max_steps=max_steps, part=p)
            k += 1
            if k % check_every == 0:
                e.sync()
                if bool(((prk & 3) != 0).all()):
                    break
        dt = time.perf_counter() - t1
        res = (N * H / dt, k)
    e.set_parts(0)
    return res + (e.metrics(),)

r0, m0 = per_step()
print(f"{case} {N} envs: per-step launches (muavta_rl_step_device): {r0 / 1e6:.1f} M env-steps/s, 150 launches")
for ms in (0, 2, 3, 4, 5, 6, 8, 12):
    for ce in (1, 4):
        r, k, m = run_ahead(ms, ce)
        same = bool(np.array_equal(m, m0))
        print(f"  run-ahead max_steps={ms:2d} check every {ce}: {r / 1e6:6.1f} M env-steps/s, {k} launches ({k / H:.3f} policy calls per env step), metrics equal per-step path: {same}")
for parts in (2, 4):
    for ms in (0, 4, 6):
        r, k, m = run_ahead(ms, 4, parts)
        print(f"  run-ahead {parts} parts max_steps={ms:2d} check every 4: {r / 1e6:6.1f} M env-steps/s, {k} launches per part, metrics equal: {bool(np.array_equal(m, m0))}")

print("what the outputs cost (max_steps 5, check every 4):")
for label, kw in (("all outputs", {}), ("no next_tok", dict(with_next=False)), ("no next_tok, no park_tok", dict(with_next=False, with_park=False)),
                  ("no tokens, no selected", dict(with_next=False, with_park=False, with_sel=False))):
    r, k, m = run_ahead(5, 4, **kw)
    print(f"  {label:28s} {r / 1e6:6.1f} M env-steps/s, {k} launches")
r, k, m = run_ahead(5, 4, with_scores=False)
print(f"  all outputs, NO score tensor (the plan is the plain Hungarian's: other episodes)  {r / 1e6:6.1f} M env-steps/s, {k} launches")
