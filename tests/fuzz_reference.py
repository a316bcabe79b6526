#!/usr/bin/env python3
"""Wide differential fuzz, THIS container only (not part of the test suite): random configurations are run through the REFERENCE
(imported from /root/reference by tools/refshim.py, driven by tools/gen_golden.py::run_episode) and the resulting full trace is
checked step by step against the CPU oracle with the same comparison the committed traces go through
(test_oracle_golden.check_trace): plans, every LSAP call, drained events, state, observation, final metrics — all bit for bit.

    python tests/fuzz_reference.py [first_k [n_configs [procs]]]      # e.g. 1000 400 6

    python tests/fuzz_reference.py [first_k [n [procs]]] --lists      # random LIST-VALUED actions instead of the allocator
    python tests/fuzz_reference.py --pin k [k ...]                    # commit those configs as fixtures

Lives under tests/ because it uses the oracle as its checker.  A configuration the reference itself cannot run is skipped; a
configuration the oracle disagrees on is printed with its key.  --pin writes the reference's full trace of the named configurations
as tests/golden/trace_WIDE<k>_s<seed>.npz and their configs into tests/golden/wide_configs.json: from then on they are part of
the test suite (test_oracle_golden.test_trace_bit_exact; on the GPU test_fuzzed_config_stepwise_vs_oracle_and_reference_metrics)."""
import json
import os
import random
import sys
import time
import traceback
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


BIG = 1_000_000  # keys from here on draw the LARGE family: fleets of up to 64 UAVs, up to 48 static tasks, up to 40 threats
EDGE = 2_000_000  # ... and from here on the EDGE family: small-family draws with a few knobs at extreme values


def wide_config(k: int) -> dict:
    """A wider net than tools/gen_golden.py::fuzz_config: fleets up to 16 UAVs (k >= BIG: up to 64 — the 24- and 64-agent tiles),
    every knob agentEnvOptions has on this path, odd radii / windows / horizons; the replan interval and the episode seed are part
    of the draw."""
    r = random.Random(0x5EED0000 + k)
    pick = r.choice
    big = r.random() < 0.35
    hi = 4 if big else 2
    large = BIG <= k < EDGE
    if large:
        hi = pick([5, 6, 8, 12, 16])
    agents = {t: r.randint(0, hi) for t in ("F1", "F2", "R1", "R2")}
    if agents["F1"] + agents["F2"] == 0:
        agents[pick(["F1", "F2"])] = 1
    if agents["R1"] + agents["R2"] == 0:
        agents[pick(["R1", "R2"])] = 1
    if r.random() < 0.3:  # group order is semantic (ids, names, shuffles): permute it
        keys = list(agents)
        r.shuffle(keys)
        agents = {t: agents[t] for t in keys}
    agents = {t: n for t, n in agents.items() if n > 0 or r.random() < 0.5}
    threats = []
    if r.random() < 0.85:
        threats.append(("T1", r.randint(1, 24 if large else 6)))
    if r.random() < 0.7:
        threats.append(("T2", r.randint(1, 16 if large else 5)))
    if len(threats) == 2 and r.random() < 0.3:
        threats.reverse()
    cfg = {
        "agents": agents, "tasks": {"Att": r.randint(0, 16 if large else 5), "Rec": r.randint(1, 32 if large else 6), "Hold": pick([0, 0, 0, 1, 2])},
        "threats_list": threats, "max_time_steps": pick([40, 60, 97, 150, 150, 200, 260]),
        "simulation_frame_rate": pick([0.01, 0.01, 0.02, 0.015, 0.008]),
        "multiple_tasks_per_agent": pick([True, True, True, False]), "random_init_pos": pick([False, False, True]),
        "num_obstacles": pick([0, 0, 0, 0, 2, 3]), "fail_rate": pick([0.0, 0.0, 0.05, 0.2, 0.5]),
        "early_terminate": pick([False, False, True]), "capability_mask": pick([False, True]), "saturate_mask": pick([False, True]),
        "reward_weights": pick([None, None, {"action": 0.5, "distance": 1.0, "quality": 0.7, "s_quality": 1.0, "time": 0.1, "alloc": 0.2,
                                             "time_penaulty": 0.25, "step": 0.3},
                                {"action": 1.0, "distance": 0.0, "quality": 1.0, "s_quality": 0.0, "time": 1.0, "alloc": 1.0,
                                 "time_penaulty": 0.0, "step": 1.0}]),
        "arrival_rate": pick([0.0, 0.03, 0.08, 0.2, 0.45]), "include_time_windows": pick([False, True]),
        "dynamic_idle_penalty": pick([0.0, 0.05, 0.5]), "sense_radius": pick([0.0, 40.0, 90.0, 250.0, 2000.0]),
        "threat_delay": pick([0, 1, 6, 20, 45]), "hard_windows": pick([False, True, True]), "window_length": pick([3, 12, 25, 40, 90]),
        "burst_mode": pick([False, True]), "burst_size": pick([1, 2, 3, 4]), "miss_penalty": pick([0.0, 25.0, 30.0]),
        "on_time_bonus": pick([0.0, 10.0, 12.0]), "dual_region_bursts": pick([False, True]),
        "share_knowledge": pick([True, True, False]), "commit_horizon": pick([0, 0, 10, 25]), "reassign_penalty": pick([0.0, 2.0, 0.5]),
        "escort_enabled": pick([False, True]), "escort_radius": pick([40.0, 70.0, 120.0]), "escort_requirement": pick([1.2, 2.5, 0.8, 3.1]),
        "escort_intercept_radius": pick([100.0, 60.0, 200.0]), "mutual_support_radius": pick([80.0, 150.0, 20.0]),
        "escort_agent_types": pick([("F1", "F2"), ("F2",), ("F1", "F2", "R2"), ("F1",)]),
    }
    if cfg["reward_weights"] is None:
        del cfg["reward_weights"]
    interval = pick([1, 5, 12, 12, 20, 20, 33])
    seed = pick([0, 1, 2, 3, 4, 7, 123456789, 2 ** 32 + 5, 2 ** 63 - 1])
    if k >= EDGE:  # the EDGE family: a handful of knobs pushed to values at the rim of what the options accept
        for _ in range(r.randint(2, 5)):
            key, values = pick([("max_time_steps", [1, 2, 3, 5, 11]), ("fail_rate", [0.9, 1.0]), ("arrival_rate", [0.9, 1.0]),
                                ("simulation_frame_rate", [0.002, 0.05, 0.1]), ("sense_radius", [1.0, 5000.0]), ("threat_delay", [200, 3]),
                                ("window_length", [1, 2, 300]), ("burst_size", [8, 20]), ("escort_requirement", [0.1, 9.9]),
                                ("escort_radius", [1.0, 2000.0]), ("escort_intercept_radius", [1.0, 2000.0]), ("mutual_support_radius", [0.5, 3000.0]),
                                ("miss_penalty", [1000.0]), ("on_time_bonus", [1000.0]), ("reassign_penalty", [50.0]), ("dynamic_idle_penalty", [10.0]),
                                ("commit_horizon", [1, 500]), ("agents", [{"F1": 1, "R1": 1}, {"F2": 1, "R2": 1}, {"R2": 2, "F1": 1}, {"F1": 8, "R1": 8}]),
                                ("tasks", [{"Att": 0, "Rec": 1, "Hold": 0}, {"Att": 1, "Rec": 1, "Hold": 3}, {"Att": 12, "Rec": 1, "Hold": 0}]),
                                ("threats_list", [[], [("T1", 1)], [("T2", 16)], [("T1", 8), ("T2", 8)]])])
            cfg[key] = pick(values)
        interval = pick([1, 2, 3, 50, 1000])
    return {"cfg": cfg, "interval": interval, "seed": seed}


def run_one(k: int):
    import numpy as np  # noqa: F401
    import gen_golden as G
    from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions
    from muavta_amd.params import params_from_config
    import test_oracle_golden as TOG

    w = wide_config(k)
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]

    def make_env(case, _cfg=cfg):
        return G.MultiUAVEnv(agentEnvOptions(render_speed=-1, action_mode="TaskAssign", multiple_agents_per_task=True, fixed_seed=-1, **_cfg))

    G.make_env = make_env
    t0 = time.time()
    lists = "--lists" in sys.argv
    try:
        if lists:  # no allocator: random list-valued actions drive the reference (repeated tasks, indices beyond the list, dead agents)
            rng = np.random.default_rng(2000 + k)

            def next_actions(env, t):
                actions, items = {}, []
                names = [a.name for a in env.agents_obj]
                for j in rng.permutation(len(names))[:int(rng.integers(0, len(names) + 1))]:
                    idxs = [int(rng.integers(0, 6)) if rng.random() < 0.9 else int(rng.integers(20, 140)) for _ in range(int(rng.integers(1, 7)))]
                    actions[names[j]] = idxs if (len(idxs) > 1 or rng.random() < 0.5) else idxs[0]
                    items += [(env.agent_by_name[names[j]].id, i) for i in idxs]
                return actions, items

            tr = G.drive_with_actions(make_env(None), seed, min(cfg["max_time_steps"], 100), next_actions)
        else:
            tr = G.run_episode(f"WIDE{k}", seed, interval, True)
    except Exception as exc:  # a combination the reference itself cannot run
        return k, "skip", f"{type(exc).__name__}: {exc}", 0.0
    t_ref = time.time() - t0
    try:
        P = params_from_config(dict(cfg), None, tile_agents=64, tile_tasks=128, tile_threats=48)
        if lists:
            TOG.check_lists(tr, f"WIDE{k}", P)
        else:
            TOG.check_trace(tr, f"WIDE{k}", P, seed)
    except AssertionError as exc:
        if "--save" in sys.argv:
            np.savez_compressed(os.path.join(HERE, "golden", f"trace_WIDE{k}_s{seed}.npz"), **tr)
        return k, "MISMATCH", str(exc)[:400], t_ref
    except Exception as exc:
        return k, "ERROR", "".join(traceback.format_exception_only(type(exc), exc))[:400], t_ref
    return k, "ok", f"steps {tr['pos'].shape[0] - 1} tasks {int(tr['n_task_ids'])}", t_ref


def pin(ks):
    import numpy as np
    import gen_golden as G
    from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions

    path = os.path.join(HERE, "golden", "wide_configs.json")
    configs = json.load(open(path)) if os.path.exists(path) else {}
    for k in ks:
        w = wide_config(k)
        G.make_env = lambda case, _cfg=w["cfg"]: G.MultiUAVEnv(agentEnvOptions(render_speed=-1, action_mode="TaskAssign", multiple_agents_per_task=True,
                                                                                 fixed_seed=-1, **_cfg))
        tr = G.run_episode(f"WIDE{k}", w["seed"], w["interval"], True)
        out = os.path.join(HERE, "golden", f"trace_WIDE{k}_s{w['seed']}.npz")
        np.savez_compressed(out, **tr)
        configs[f"WIDE{k}"] = w["cfg"]
        print(out, os.path.getsize(out) // 1024, "KiB", "steps", tr["pos"].shape[0] - 1, "tasks", int(tr["n_task_ids"]), "S_WPS", tr["metrics"][4])
    with open(path, "w") as f:
        json.dump(configs, f, indent=1)  # key order is semantic: groups are created in dict order


def pin_scored(ks):
    """The scored-allocator leg of tests/fuzz_device.py for config k (pseudo-random scores / priorities / reserved agents through the
    ORACLE's scored allocator) gives an action sequence; the REFERENCE is driven with exactly those actions and its trace is
    committed as tests/golden/lists_WIDE<k>_s<seed>.npz (the format of the list-valued action traces: oracle and device replay
    it in test_list_valued_actions_trace_bit_exact / test_reference_list_valued_action_traces_on_the_device)."""
    import numpy as np
    import gen_golden as G
    import orc
    from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions
    from muavta_amd.params import params_from_config

    SCORED = (("allocator", 0, 3), ("force", 2, 4), ("escort", 1, 1), ("trainer", 0, 1), ("trainer", 0, 2), ("escort", 2, 4))  # (gate, token kind, oracle flags): fuzz_device.SCORED
    PADS = ((32, 16), (6, 3), (48, 16), (12, 8))
    GATE = {"force": 0, "trainer": 1, "escort": 2, "allocator": 3}
    path = os.path.join(HERE, "golden", "wide_configs.json")
    configs = json.load(open(path)) if os.path.exists(path) else {}
    for k in ks:
        k, env_i = (int(x) for x in (str(k).split(":") + ["0"])[:2])  # "k:i" = env i of the fuzz leg's batch (seed + i, the i-th rows of its inputs)
        w = wide_config(k)
        cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
        seed = seed - env_i if seed > 2 ** 62 else seed + env_i
        gate, kind, oflags = SCORED[k % len(SCORED)]
        mt, ma = PADS[(k // len(SCORED)) % len(PADS)]
        o = orc.OracleEnv(params_from_config(dict(cfg), None, tile_agents=64, tile_tasks=128, tile_threats=48))
        o.reset(seed)
        rng = np.random.default_rng(1000 + k)
        n, A = 2, o.A

        def next_actions(env, t):
            sc = (rng.uniform(-1, 1, (n, ma, mt)) * (0.35 if kind != 2 else 1.0)).astype(np.float32)
            pri = rng.uniform(-0.5, 1, (n, mt))
            res = rng.integers(0, 1 << A, n, dtype=np.uint64) & rng.integers(0, 1 << A, n, dtype=np.uint64)
            oa, oi, _ = o.allocate_scored(interval, int(bool((t // 3) % 2)), GATE[gate], kind, mt, ma, oflags, scores=sc[env_i], pri=pri[env_i], reserved=int(res[env_i]))
            o.step(oa, oi)
            actions, items = {}, []
            for a, i in zip(oa, oi):
                actions.setdefault(env.agents_obj[int(a)].name, []).append(int(i))
                items.append((int(a), int(i)))
            return {name: (v if len(v) > 1 else v[0]) for name, v in actions.items()}, items

        env = G.MultiUAVEnv(agentEnvOptions(render_speed=-1, action_mode="TaskAssign", multiple_agents_per_task=True, fixed_seed=-1, **cfg))
        tr = G.drive_with_actions(env, seed, cfg["max_time_steps"], next_actions)
        tr["multi"] = np.int64(bool(cfg["multiple_tasks_per_agent"]))
        out = os.path.join(HERE, "golden", f"lists_WIDE{k}_s{seed}.npz")
        np.savez_compressed(out, **tr)
        configs[f"WIDE{k}"] = cfg
        print(out, os.path.getsize(out) // 1024, "KiB", "steps", tr["pos"].shape[0] - 1, "items", len(tr["actions"]))
    with open(path, "w") as f:
        json.dump(configs, f, indent=1)


if __name__ == "__main__" and "--pin-scored" in sys.argv:
    pin_scored([a for a in sys.argv[1:] if not a.startswith("--")])
    sys.exit(0)

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
        for k, status, msg, t_ref in pool.imap_unordered(run_one, range(first, first + n)):
            counts[status] = counts.get(status, 0) + 1
            if status != "ok" or "--verbose" in sys.argv:
                print(f"k={k} {status}: {msg}", flush=True)
                if status in ("MISMATCH", "ERROR"):
                    print("   ", json.dumps(wide_config(k)), flush=True)
    print(f"configs {first}..{first + n - 1}: {counts}  ({time.time() - t0:.0f} s)", flush=True)
