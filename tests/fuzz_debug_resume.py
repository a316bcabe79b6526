#!/usr/bin/env python3
"""Debug helper (GPU): the resume leg of tests/fuzz_device.py for one config, printing which observation cells differ."""
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
        ti, legal, pad, ag, fl = o.observe()
        d = np.argwhere(snap.obs["tasks"][i] != ti)
        print(" differing (row, col):", d[:20].tolist(), "n_open", len(o.open_ids()), "open", o.open_ids().tolist())
        for r, c in d[:8]:
            print("  row", r, "col", c, "dev", snap.obs["tasks"][i][r, c], "orc", ti[r, c])
        print(" SCALARS time", snap.SCALARS[i][0], "term/trunc", o.dims()["terminated"], o.dims()["truncated"], "time_steps", o.dims()["time_steps"])
        raise
FD.compare = dbg
print(FD.resume(k, wide_config(k), print))
