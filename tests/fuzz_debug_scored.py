#!/usr/bin/env python3
"""Debug helper (GPU): the scored-allocator leg of tests/fuzz_device.py for one config, printing the state at the first mismatch.
    python tests/fuzz_debug_scored.py k"""
import sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import orc
from fuzz_reference import wide_config
from fuzz_device import TILES, SCORED, PADS, params
from muavta_amd.batched import BatchedMultiUAVEnv
from test_gpu_parity import Snapshot, compare, GATE

k = int(sys.argv[1])
w = wide_config(k); cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
gate, kw, kname, kind, oflags = SCORED[k % len(SCORED)]
mt, ma = PADS[(k // len(SCORED)) % len(PADS)]
tile = TILES[int(sys.argv[2])] if len(sys.argv) > 2 else TILES[(k // 2) % 3]
p = params(cfg, tile)
print(cfg, "interval", interval, "seed", seed, gate, kw, kname, mt, ma, tile)
n = 2
seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
env = BatchedMultiUAVEnv(p, n); A = env.n_agents
oracles = [orc.OracleEnv(p) for _ in range(n)]
env.reset(seeds)
for i, o in enumerate(oracles): o.reset(int(seeds[i]))
rng = np.random.default_rng(1000 + k)
for t in range(p.max_time_steps):
    if any(bool(o.dims()["terminated"] or o.dims()["truncated"]) for o in oracles): break
    sc = (rng.uniform(-1, 1, (n, ma, mt)) * (0.35 if kind != 2 else 1.0)).astype(np.float32)
    pri = rng.uniform(-0.5, 1, (n, mt))
    res = rng.integers(0, 1 << A, n, dtype=np.uint64) & rng.integers(0, 1 << A, n, dtype=np.uint64)
    vis = bool((t // 3) % 2)
    out = env.allocate_scored(kname, mt, ma, edge_scores=sc, task_pri=pri, reserved=res, gate=gate, replan_interval=interval, use_visibility=vis, **kw)
    for i, o in enumerate(oracles):
        oa, oi, osel = o.allocate_scored(interval, int(vis), GATE[gate], kind, mt, ma, oflags, scores=sc[i], pri=pri[i], reserved=int(res[i]))
        kk = len(oa)
        if not (np.array_equal(out["act_agent"][i][:kk], oa) and np.all(out["act_agent"][i][kk:] == -1) and np.array_equal(out["act_index"][i][:kk], oi)):
            print("PLAN MISMATCH t", t, "env", i, out["act_agent"][i], out["act_index"][i], oa, oi); sys.exit(0)
        o.step(oa, oi)
    pre_esc = env.get("ESCORTS").copy(); pre_q = env.get("AGENT_QUEUE").copy(); pre_state = env.get("AGENT_STATE").copy()
    staged = env.get("STAGED_ACTIONS").copy()
    env.step_staged()
    snap = Snapshot(env)
    for i, o in enumerate(oracles):
        try:
            compare(snap, i, o, f"seed {seeds[i]} t={t+1}")
        except AssertionError as exc:
            print("MISMATCH", exc, "ERROR", snap.ERROR)
            trow, reqs = o.tasks(); ids = snap.TASK_ID[i]
            print(" device slots (id,status):", [(int(x), int(snap.TASK_STATUS[i, s])) for s, x in enumerate(ids) if x >= 0])
            print(" device OPEN_IDS:", snap.OPEN_IDS[i].tolist())
            print(" oracle open_ids:", o.open_ids().tolist())
            print(" oracle open:", [kk for kk in range(1, trow.shape[0]) if int(trow[kk, 0]) != 2], "n tasks", trow.shape[0])
            print(" oracle rows of missing:", [(kk, trow[kk, :13].tolist()) for kk in range(1, trow.shape[0]) if int(trow[kk, 0]) != 2 and kk not in set(ids.tolist())])
            rows, caps, q = o.agents()
            print(" queues dev", snap.AGENT_QUEUE[i][:, :5].tolist()); print(" queues orc", q[:, :5].tolist())
            print(" states", snap.AGENT_STATE[i], rows[:, 2].astype(int))
            print(" events", o.events().tolist(), "dev events", snap.EVENTS[i][:8].tolist())
            print(" staged actions (agent, task id...)", staged[i][:10].tolist())
            print(" before the step: escorts", pre_esc[i][:6].tolist(), "queues", pre_q[i][:, :4].tolist(), "states", pre_state[i].tolist())
            print(" after: escorts", env.get("ESCORTS")[i][:6].tolist(), "types", snap.AGENT_TYPE[i].tolist())
            print(" scalars dev", snap.SCALARS[i].tolist()); print(" scalars orc", o.scalars().tolist(), o.dims())
            sys.exit(0)
print("no mismatch")
