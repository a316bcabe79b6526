#!/usr/bin/env python3
"""Debug helper for tests/fuzz_device.py (GPU): run ONE leg on ONE configuration and, at the first state mismatch, print what differs —
tasks (position / status / times / allocationDetails), queues, states, open lists, observation cells, scalars.

    python tests/fuzz_debug.py <leg> <k>        # leg: stepwise | scored | lists | rl | rings | mutators | resume | ilrings"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import fuzz_device as FD  # noqa: E402
import test_gpu_parity as TG  # noqa: E402
from fuzz_reference import wide_config  # noqa: E402

leg, k = sys.argv[1], int(sys.argv[2])
plain = TG.compare


def verbose_compare(snap, i, o, tag, check_obs=True):
    try:
        plain(snap, i, o, tag, check_obs)
    except AssertionError as exc:
        print("MISMATCH", exc, "ERROR", snap.ERROR.tolist())
        trow, reqs = o.tasks()
        ids = snap.TASK_ID[i]
        for s in np.nonzero(ids >= 0)[0]:
            kk = int(ids[s])
            same = (np.array_equal(snap.TASK_POS[i, s], trow[kk, 1:3]) and snap.TASK_STATUS[i, s] == int(trow[kk, 0]) and
                    (int(trow[kk, 0]) == 2 or (np.array_equal(snap.TASK_TIMES[i, s], trow[kk, 3:5]) and snap.TASK_META[i, s][5] == int(trow[kk, 5]))))
            if not same:
                print(" task", kk, "slot", int(s), "dev pos", snap.TASK_POS[i, s], "orc", trow[kk, 1:3], "status", int(snap.TASK_STATUS[i, s]), int(trow[kk, 0]),
                      "times", snap.TASK_TIMES[i, s], trow[kk, 3:5], "meta dev", snap.TASK_META[i, s].tolist(), "orc", trow[kk, 5:13].tolist())
        print(" device slots (id, status):", [(int(x), int(snap.TASK_STATUS[i, s])) for s, x in enumerate(ids) if x >= 0])
        print(" device OPEN_IDS:", [int(x) for x in snap.OPEN_IDS[i] if x >= 0])
        print(" oracle open_ids:", o.open_ids().tolist())
        rows, caps, q = o.agents()
        print(" queues dev", snap.AGENT_QUEUE[i][:, :6].tolist())
        print(" queues orc", q[:, :6].tolist())
        print(" states dev", snap.AGENT_STATE[i].tolist(), "orc", rows[:, 2].astype(int).tolist())
        print(" scalars dev", snap.SCALARS[i].tolist())
        print(" scalars orc", o.scalars().tolist(), o.dims())
        if check_obs:
            ti = o.observe()[0]
            d = np.argwhere(snap.obs["tasks"][i] != ti)
            print(" observation cells differing (row, col):", d[:12].tolist())
        raise


FD.compare = verbose_compare
fn = getattr(FD, leg)
kw = {"verbose": True} if leg == "mutators" else {}
print(fn(k, wide_config(k), print, **kw))
