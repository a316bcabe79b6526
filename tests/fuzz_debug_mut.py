#!/usr/bin/env python3
"""Debug helper (GPU): the mutators leg of tests/fuzz_device.py for one config with details at the first mismatch."""
import sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import fuzz_device as FD
from fuzz_reference import wide_config
import test_gpu_parity as TG
k = int(sys.argv[1])
orig = TG.compare
def dbg(snap, i, o, tag, check_obs=True):
    try:
        orig(snap, i, o, tag, check_obs)
    except AssertionError as exc:
        print("MISMATCH", exc)
        trow, reqs = o.tasks(); ids = snap.TASK_ID[i]
        for s in np.nonzero(ids >= 0)[0]:
            kk = int(ids[s])
            if int(trow[kk, 0]) != 2 and (not np.array_equal(snap.TASK_TIMES[i, s], trow[kk, 3:5]) or snap.TASK_META[i, s][5] != int(trow[kk, 5])):
                print(" task", kk, "dev times", snap.TASK_TIMES[i, s], "orc", trow[kk, 3:5], "ndet dev/orc", snap.TASK_META[i, s][5], int(trow[kk, 5]), "type", int(trow[kk, 6]), "status", int(trow[kk,0]))
        rows, caps, q = o.agents()
        print(" queues dev", snap.AGENT_QUEUE[i][:, :5].tolist()); print(" queues orc", q[:, :5].tolist())
        print(" nft dev", snap.AGENT_NFT[i].tolist()); print(" nft orc", rows[:, 5].tolist())
        print(" states", snap.AGENT_STATE[i].tolist(), "time", o.dims()["time_steps"])
        raise
FD.compare = dbg
print(FD.mutators(k, wide_config(k), print, verbose=True))
