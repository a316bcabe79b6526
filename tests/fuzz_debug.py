#!/usr/bin/env python3
"""Debug helper for tests/fuzz_device.py (GPU): re-run one config stepwise with a given tile / mode and print the state around the
first mismatch.    python tests/fuzz_debug.py k tile_index mode_index [seed_offset]"""
import sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import orc
from fuzz_reference import wide_config
from fuzz_device import TILES, MODES, params
from muavta_amd.batched import BatchedMultiUAVEnv
from test_gpu_parity import Snapshot, compare

k, ti, mi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w = wide_config(k); cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
mode, name = MODES[mi]; tile = TILES[ti]
p = params(cfg, tile)
n = 2
seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
env = BatchedMultiUAVEnv(p, n); env.set_allocator(name)
oracles = [orc.OracleEnv(p) for _ in range(n)]
env.reset(seeds)
for i, o in enumerate(oracles): o.reset(int(seeds[i]))
for t in range(p.max_time_steps):
    if any(bool(o.dims()["terminated"] or o.dims()["truncated"]) for o in oracles): break
    aa, ai = env.allocate(interval, True)
    for i, o in enumerate(oracles):
        oa, oi = o.allocate_mode(interval, 1, mode)
        o.step(oa, oi)
    env.step(aa, ai)
    snap = Snapshot(env)
    for i, o in enumerate(oracles):
        try:
            compare(snap, i, o, f"seed {seeds[i]} t={t+1}")
        except AssertionError as exc:
            print("MISMATCH", exc)
            trow, reqs = o.tasks()
            ids = snap.TASK_ID[i]
            for s in np.nonzero(ids >= 0)[0]:
                kk = int(ids[s])
                if not np.array_equal(snap.TASK_POS[i, s], trow[kk, 1:3]) or snap.TASK_STATUS[i, s] != int(trow[kk, 0]):
                    print(" task", kk, "slot", s, "dev pos", snap.TASK_POS[i, s], "orc pos", trow[kk, 1:3], "status dev/orc", snap.TASK_STATUS[i, s], int(trow[kk, 0]),
                          "meta dev", snap.TASK_META[i, s], "orc row", trow[kk, 5:13])
            rows, caps, q = o.agents()
            print(" agents dev pos", snap.AGENT_POS[i].tolist()); print(" agents orc pos", rows[:, 0:2].tolist())
            print(" states dev", snap.AGENT_STATE[i], "orc", rows[:, 2].astype(int)); print(" queues dev", snap.AGENT_QUEUE[i][:, :4].tolist(), "orc", q[:, :4].tolist())
            print(" events", o.events().tolist(), "escorts", env.get("ESCORTS")[i].tolist() if "ESCORTS" in getattr(env, "FIELDS", {"ESCORTS": 1}) else None)
            ti_, legal, pad, ag, fl = o.observe()
            d = snap.obs["tasks"][i] != ti_
            print(" obs rows differing", np.unique(np.nonzero(d)[0]).tolist(), "n_open oracle", len(o.open_ids()), "max_tasks", p.max_tasks)
            sys.exit(0)
print("no mismatch")
