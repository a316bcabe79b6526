#!/usr/bin/env python3
"""Wide differential fuzz of the policy-in-the-loop path, THIS container only (not part of the test suite): random configurations
(fuzz_reference.wide_config; fleets of at most 16 UAVs, the trainers' token pad) are run through the REFERENCE's run_rl_episode loop
with PairCostHybrid.plan(scores=<seeded matrix>) (tools/gen_golden.py::rl_episode) and checked against the oracle's scored allocator
with the comparison the committed rl_*.npz traces go through (test_oracle_golden.check_rl): tokens, edge_valid, every scored LSAP
cost matrix and assignment, _selected_mask, pairs, actions, step rewards, next tokens, done flags, final metrics — bit for bit; (r5) and the
oracle's run-to-the-next-gate (OracleEnv.rl_run, the checker of muavta_rl_run_device) replays the same reference episode launch by launch.

    python tests/fuzz_reference_rl.py [first_k [n_configs [procs]]]
    python tests/fuzz_reference_rl.py --pin k [k ...]     # commit the reference's RL episode of those configurations as fixtures

--pin writes tests/golden/rlterm_WIDE<k>.npz (the format of rl_<case>.npz) and the configuration into tests/golden/rlterm_configs.json:
episodes that END EARLY (every task done before max_time_steps; about 4 % of the draws) — the registry cases' rl_*.npz all run to
their truncation step.  test_oracle_golden.py replays them through the oracle's per-step and run-to-the-gate checkers.

Lives under tests/ because it uses the oracle as its checker."""
import os
import sys
import time
import traceback
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from fuzz_reference import wide_config  # noqa: E402


def run_one(k: int):
    import gen_golden as G
    from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions
    from muavta_amd.params import params_from_config
    import test_oracle_golden as TOG

    w = wide_config(k)
    cfg, seed = w["cfg"], w["seed"]
    raw = bool(k & 1)
    try:
        env = G.MultiUAVEnv(agentEnvOptions(render_speed=-1, action_mode="TaskAssign", multiple_agents_per_task=True, fixed_seed=-1, **cfg))
        tr = G.rl_episode(env, seed, raw)
    except Exception as exc:  # a combination the reference itself cannot run
        return k, "skip", f"{type(exc).__name__}: {str(exc)[:200]}"
    try:
        P = params_from_config(dict(cfg), None, tile_agents=64, tile_tasks=128, tile_threats=48)
        TOG.check_rl(tr, f"WIDE{k}", P)
        if len(tr["replanned"]) and not bool(tr["ep_done"][:-1].any() if len(tr["ep_done"]) else False):
            TOG.check_rl_run(tr, f"WIDE{k} run-ahead", P, (0, 3, 1)[k % 3])  # (r5) the run-to-the-next-gate checker, launch by launch on the same reference episode
    except AssertionError as exc:
        return k, "MISMATCH", str(exc)[:400]
    except Exception as exc:
        return k, "ERROR", "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-700:]
    return k, "ok", f"plans {len(tr['step'])} S_WPS {tr['metrics'][4]:.3f}"


def pin(ks):
    import json

    import numpy as np
    import gen_golden as G
    from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions

    path = os.path.join(HERE, "golden", "rlterm_configs.json")
    configs = json.load(open(path)) if os.path.exists(path) else {}
    for k in ks:
        w = wide_config(k)
        env = G.MultiUAVEnv(agentEnvOptions(render_speed=-1, action_mode="TaskAssign", multiple_agents_per_task=True, fixed_seed=-1, **w["cfg"]))
        tr = G.rl_episode(env, w["seed"], bool(k & 1))
        del tr["interval"]
        out = os.path.join(HERE, "golden", f"rlterm_WIDE{k}.npz")
        np.savez_compressed(out, **tr)
        configs[f"WIDE{k}"] = w["cfg"]
        print(out, os.path.getsize(out) // 1024, "KiB", "steps", len(tr["replanned"]), "of", w["cfg"]["max_time_steps"], "plans", len(tr["step"]),
              "ep_done", tr["ep_done"].tolist()[-3:], "S_WPS", tr["metrics"][4])
    with open(path, "w") as f:
        json.dump(configs, f, indent=1)  # key order is semantic: groups are created in dict order


if __name__ == "__main__" and "--pin" in sys.argv:
    pin([int(a) for a in sys.argv[1:] if not a.startswith("--")])
    sys.exit(0)

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if len(args) > 0 else 0
    n = int(args[1]) if len(args) > 1 else 100
    procs = int(args[2]) if len(args) > 2 else 4
    counts = {}
    t0 = time.time()
    with Pool(procs) as pool:
        for k, status, msg in pool.imap_unordered(run_one, range(first, first + n)):
            counts[status] = counts.get(status, 0) + 1
            if status != "ok" or "--verbose" in sys.argv:
                print(f"k={k} {status}: {msg}", flush=True)
    print(f"configs {first}..{first + n - 1}: {counts}  ({time.time() - t0:.0f} s)", flush=True)
