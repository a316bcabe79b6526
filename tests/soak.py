#!/usr/bin/env python3
"""Ad-hoc soak (not part of the test suite): fused device rollouts vs the oracle on many seeds per (case, allocator);
reports mismatching seeds and tile-capacity overflows (flagged envs are skipped in the comparison)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/soak.py: lives under tests/ because it uses the oracle as its checker
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_for_case
PLAN = [("WPS_easy", "hungarian", 0, 20, 8192), ("WPS_hard", "hungarian", 0, 20, 16384), ("WPS_burst", "hungarian", 0, 20, 8192),
        ("WPS_attn", "hungarian", 0, 20, 4096), ("WPS_attn_AWACS", "hungarian", 0, 20, 4096), ("D2_popup_threats", "hungarian", 0, 20, 4096),
        ("WPS_hard_x2", "hungarian", 0, 20, 32768), ("WPS_escort", "hungarian", 0, 12, 8192), ("WPS_escort24", "hungarian", 0, 12, 4096),
        ("WPS_burst64", "hungarian", 0, 20, 1024), ("WPS_hard", "urgency_pair", 1, 20, 8192), ("WPS_attn", "urgency_pair", 1, 20, 4096),
        ("WPS_escort", "urgency_coalition", 2, 12, 8192), ("WPS_escort24", "urgency_coalition", 2, 12, 2048), ("WPS_hard", "hungarian_gated", 3, 20, 8192),
        ("WPS_attn_XL", "hungarian", 0, 20, 1024), ("WPS_attn_L", "hungarian", 0, 20, 1024), ("WPS_burst64", "urgency_coalition", 2, 12, 512),
        ("WPS_burst64", "urgency_pair", 1, 20, 512), ("WPS_attn_XL", "urgency_pair", 1, 20, 512)]
base = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for case, name, mode, interval, n in PLAN:
    env = BatchedMultiUAVEnv(params_for_case(case), n)
    env.set_allocator(name)
    seeds = np.arange(base, base + n, dtype=np.uint64)
    env.rollout(seeds, 150, interval, True, False)
    got, err = env.rollout_metrics(), env.get("ERROR")
    o = orc.OracleEnv(params_for_case(case))
    t0 = time.time(); bad = []
    for i, s in enumerate(seeds):
        if err[i]:
            continue
        o.rollout_mode(int(s), 150, interval, 1, mode)
        if not np.array_equal(got[i], o.metrics()):
            bad.append(int(s))
    print(f"{case:18s} {name:18s} {n:6d} seeds: mismatches {len(bad)} {bad[:5]}  capacity-flagged {int((err != 0).sum())} codes {np.unique(err[err != 0]).tolist()}  (oracle {time.time() - t0:.1f}s)", flush=True)
