"""Host-side logic on CPU: the PettingZoo-style facade over a (test-only) oracle backend, the config
packing, and — when the read-only reference checkout is present in this container — the reference's
OWN allocator + harness glue driving the facade unchanged (drop-in proof)."""
import os
import sys

import numpy as np
import pytest

import orc
from oracle_backend import OracleBackend
from muavta_amd.env import MultiUAVEnv
from muavta_amd.params import METRIC_KEYS, params_for_case, params_from_config
from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _facade(case):
    p = params_for_case(case)
    return MultiUAVEnv(CASE_SPECS[case], backend=OracleBackend(p), flags=dict(WPS_ENV_FLAGS)), p


@pytest.mark.parametrize("case,seed", [("WPS_easy", 0), ("WPS_hard", 1), ("WPS_escort", 0)])
def test_facade_episode_matches_golden(case, seed):
    g = np.load(os.path.join(GOLDEN, f"trace_{case}_s{seed}.npz"))
    env, p = _facade(case)
    obs, infos = env.reset(seed=seed)
    assert set(obs) == set(env.possible_agents) and infos == {n: {} for n in env.possible_agents}
    interval = int(g["interval"])
    T = int(g["max_tasks"])
    t = 0
    done = False
    while not done:
        # golden observation of this step (what the reference returned)
        first = obs[env.agents_obj[0].name]
        assert len(first["tasks_info"]) == T and len(first["mask"]) == T
        want = g["obs_tasks"][t]
        for j, info in enumerate(first["tasks_info"]):
            if want[j, 3] == -1 and want[j, 0] == 0 and not want[j].any() is False:
                pass
            if "id" not in info:
                assert info == {"status": -1} and want[j, 3] == -1
            else:
                assert info["id"] == int(want[j, 0]) and info["status"] == int(want[j, 3])
                assert np.allclose(info["position"], want[j, 1:3], rtol=0, atol=1e-7)
                assert np.array_equal(np.float32(info["current_reqs"]), want[j, 4:10])
                assert set(info) >= {"id", "position", "status", "current_reqs", "alloc_reqs", "unmet", "age", "init_time", "end_time", "type_idx"}
        legal = np.unpackbits(g["obs_legal"][t], axis=-1)[:, :T].astype(bool)
        for a in env.agents_obj:
            assert obs[a.name]["legal_mask"] == list(legal[a.id])
            assert obs[a.name]["alloc_task"] == int(g["head"][t][a.id])
        assert [x.id for x in env.last_tasks_info] == list(g["open_ids"][g["open_ptr"][t]:g["open_ptr"][t + 1]])
        # allocator decisions come from the oracle backend; the facade turns them into the reference's actions dict
        aa, ai = env._b.allocate(interval, True)
        actions = {env.agents_obj[int(a)].name: int(i) for a, i in zip(aa[0], ai[0]) if a >= 0}
        ga = g["actions"][g["actions"][:, 0] == t]
        assert [env.agent_by_name[n].id for n in actions] == list(ga[:, 1]) and list(actions.values()) == list(ga[:, 3])
        obs, rew, term, trunc, infos = env.step(actions)
        t += 1
        assert rew[env.agents_obj[0].name] == g["reward"][t] and len(set(rew.values())) == 1
        ev = g["events"][g["events"][:, 0] == t][:, 1:]
        from muavta_amd.params import EVENT_TAGS
        assert infos["events"] == [[EVENT_TAGS[int(x)], int(y)] for x, y in ev]
        assert env.time_steps == t and env.total_distance == g["scalars"][t][1]
        vis = env.agent_visibility_map()  # exact, ids of retired (slot-released) tasks included
        known = np.unpackbits(g["known"][t], axis=-1)[:, :int(g["n_task_ids"])].astype(bool)
        for a in env.agents_obj:
            want_known = set(np.nonzero(known[a.id])[0].tolist())
            assert vis[a.name] == want_known, f"t={t} {a.name}: {sorted(vis[a.name] ^ want_known)}"
        done = all(term.values()) or all(trunc.values())
    assert t == 150 and list(infos["metrics"].keys()) == METRIC_KEYS
    assert np.array_equal(np.array([float(infos["metrics"][k]) for k in METRIC_KEYS]), g["metrics"])
    assert isinstance(infos["metrics"]["n_on_time"], int) and isinstance(infos["metrics"]["S_WPS"], float)
    assert env.compute_s_wps() == g["metrics"][4] and env.compute_s_esc() == g["metrics"][5]


def test_facade_takes_list_valued_actions_of_any_length():
    """env.step({name: [index, ...]}) as the reference takes it (DroneEnv.py:813-838), more items per step than the tile's
    action_cap: the facade flattens the dict in order and the episode equals the reference's (tests/golden/lists_*.npz)."""
    import itertools
    from test_oracle_golden import lists_params
    path = os.path.join(GOLDEN, "lists_WPS_hard_s0.npz")
    g, p = lists_params(path)
    flags = dict(WPS_ENV_FLAGS, multiple_tasks_per_agent=bool(int(g["multi"])))
    env = MultiUAVEnv(CASE_SPECS["WPS_hard"], backend=OracleBackend(p), flags=flags)
    env.reset(seed=int(g["seed"]))
    acts = g["actions"]
    longest = 0
    for t in range(g["pos"].shape[0] - 1):
        ga = acts[acts[:, 0] == t]
        actions = {}
        for aid, grp in itertools.groupby(ga, key=lambda r: int(r[1])):  # (an agent appears once per step in these traces)
            idxs = [int(r[2]) for r in grp]
            actions[env.agents_obj[aid].name] = idxs if len(idxs) > 1 else idxs[0]
        longest = max(longest, len(ga))
        obs, rew, term, trunc, infos = env.step(actions)
        assert rew[env.agents_obj[0].name] == g["reward"][t + 1]
        assert np.array_equal(np.array([a.position for a in env.agents_obj]), g["pos"][t + 1])
        assert [x.id for x in env.last_tasks_info] == list(g["open_ids"][g["open_ptr"][t + 1]:g["open_ptr"][t + 2]])
    assert longest > env._b.A_tile


def test_object_views_identity_and_visibility():
    env, p = _facade("WPS_hard")
    env.reset(seed=3)
    t1 = env.tasks[0]
    assert t1 is env.tasks[0] and t1 in env.last_tasks_info and env.last_tasks_info.index(t1) == 0
    assert getattr(t1, "hard_deadline", None) is None and t1.kind is None and t1.status == 0
    a0 = env.agents_obj[0]
    assert a0.tasks == [env.task_idle] and a0.tasks[0].id == 0 and env.agent_by_name[a0.name] is a0
    assert a0.name in env.possible_agents and a0.type in ("F1", "F2", "R1", "R2")
    vis = env.agent_visibility_map()
    assert set(vis) == set(env.possible_agents) and all(v == {t.id for t in env.tasks} for v in vis.values())
    assert [a.id for a in env.get_live_agents()] == list(range(env.n_agents))
    env2, _ = _facade("WPS_attn_AWACS")
    env2.reset(seed=0)
    assert env2.agent_visibility_map() is None  # sense_radius == 0 and threat_delay == 0 (DroneEnv.py:1597)
    with pytest.raises(TypeError):
        env.step([0, 1])


def test_params_packing_matches_reference_defaults():
    p = params_for_case("WPS_escort")
    assert p.n_agents == 14 and p.n_tasks == 9 and p.max_tasks == 37
    assert p.possible_agents[:3] == ["F1_agent0", "F1_agent1", "F1_agent2"] and p.possible_agents[-1] == "R2_agent1"
    assert p.escort_agent_type_mask == (1 << 3) | (1 << 4) and p.escort_enabled == 1 and p.share_knowledge == 0
    assert p.window_length == 28 and p.reassign_penalty == 2.0 and p.multiple_tasks_per_agent == 1
    q = params_from_config({"agents": {"R1": 1}, "tasks": {"Rec": 2}, "window_length": 0, "burst_size": 0,
                            "miss_penalty": 0.0, "escort_radius": 0.0})
    assert q.window_length == 30 and q.burst_size == 3 and q.miss_penalty == 0.0 and q.escort_radius == 70.0  # `x or default`
    with pytest.raises(ValueError):
        params_from_config({"agents": {"R1": 1}, "tasks": {"Rec": 1}, "action_mode": "Other"})


@pytest.mark.skipif(not os.path.isdir("/root/reference/mUAV_TA"), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("case,interval", [("WPS_hard", 20), ("WPS_escort", 12)])
def test_reference_allocator_and_harness_glue_drive_the_facade(case, interval):
    """The reference's HungarianAllocator, _open_tasks and _apply_assign, imported unmodified, run against
    our MultiUAVEnv facade and reproduce the metrics the reference env produced itself."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refshim
    refshim.install()
    from TaskAllocation.OptimizationBased.HungarianAllocator import HungarianAllocator
    from experiments.paper_eval import _events, _open_tasks
    from experiments.wps_eval import _apply_assign

    g = np.load(os.path.join(GOLDEN, f"metrics_{case}.npz"))
    for seed in (0, 1):
        env, _ = _facade(case)
        obs, info = env.reset(seed=seed)
        hung = HungarianAllocator(replan_interval=interval, max_coord=env.max_coord)
        done = {a: False for a in env.agents}
        trunc = {a: False for a in env.agents}
        while not all(done.values()) and not all(trunc.values()):
            result = hung.allocate_tasks(env.get_live_agents(), _open_tasks(env), time_step=env.time_steps,
                                         events=_events(info), agent_known_ids=env.agent_visibility_map())
            obs, reward, done, trunc, info = env.step(_apply_assign(env, result))
        got = np.array([float(info["metrics"][k]) for k in METRIC_KEYS])
        assert np.array_equal(got, g["metrics"][seed]), dict(zip(METRIC_KEYS, got - g["metrics"][seed]))
        assert hung.n_replans == int(g["n_replans"][seed])


@pytest.mark.skipif(not os.path.isdir("/root/reference/mUAV_TA"), reason="reference checkout not present (GPU box)")
def test_replay_document_equals_reference_generator(tmp_path):
    """muavta_amd.replay.generate (facade + planner through the backend) writes the same dashboard document as the
    reference's experiments/generate_simulation_replay.py for its WPS_escort / Urgency-Coalition replay."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refshim
    refshim.install()
    import json
    from experiments import generate_simulation_replay as ref
    from muavta_amd import replay

    seed = 3
    want = ref.generate(seed, tmp_path / "ref.json", "WPS_escort")
    env, _ = _facade("WPS_escort")
    got = replay.generate(seed, tmp_path / "ours.json", "WPS_escort", "urgency_coalition", env=env,
                          title="WPS_escort: protect recon with fighter coalitions")
    assert json.loads((tmp_path / "ours.json").read_text()) == json.loads(json.dumps(got))
    assert got["metadata"] == want["metadata"]
    assert len(got["frames"]) == len(want["frames"]) == 151
    for k, (a, b) in enumerate(zip(got["frames"], want["frames"])):
        for key in ("time", "agents", "threats", "events", "decision", "metrics", "tasks"):
            assert a[key] == b[key], f"frame {k} {key}: " + str([(x, y) for x, y in zip(a[key], b[key]) if x != y][:2] if isinstance(a[key], list) else (a[key], b[key]))
    assert got["events"] == want["events"] and got["final_metrics"] == want["final_metrics"]


@pytest.mark.parametrize("case,n,interval", [("WPS_hard", 4, 20), ("WPS_escort", 3, 12)])
def test_vectorised_facade_views_equal_single_env_facades(case, n, interval):
    """MultiUAVEnv.batch(n) — n reference-shaped env objects over ONE handle (env.py: _SharedBatch / _RowBackend / MultiUAVEnvBatch) — over the
    oracle stand-in of an n-env handle, against n single-env facades stepped side by side with the same actions: observation dicts, rewards,
    done flags, infos, last_tasks_info, the Task / UAV views and agent_visibility_map() after every step; an out-of-step mutator through ONE
    view (UAV.allocate, then a state write) leaves its neighbours untouched; and the batch pays one whole-handle call per step, not n.
    (The same comparison runs on the HIP backend under -m gpu.)"""
    from oracle_backend import OracleBatchBackend

    p = params_for_case(case)
    backend = OracleBatchBackend(p, n)
    batch = MultiUAVEnv.batch(CASE_SPECS[case], n, flags=dict(WPS_ENV_FLAGS), backend=backend)
    singles = [MultiUAVEnv(CASE_SPECS[case], backend=OracleBackend(p), flags=dict(WPS_ENV_FLAGS)) for _ in range(n)]
    seeds = [3 + 11 * i for i in range(n)]
    outs = batch.reset(seeds)
    refs = [s.reset(seed=sd) for s, sd in zip(singles, seeds)]

    def same_obs(a, b):
        assert a.keys() == b.keys()
        for name in a:
            for key in ("agent_position", "agent_caps", "event_flags"):
                assert np.array_equal(a[name][key], b[name][key])
            assert a[name]["alloc_task"] == b[name]["alloc_task"] and a[name]["mask"] == b[name]["mask"] and a[name]["legal_mask"] == b[name]["legal_mask"]
            assert len(a[name]["tasks_info"]) == len(b[name]["tasks_info"])
            for ra, rb in zip(a[name]["tasks_info"], b[name]["tasks_info"]):
                assert ra.keys() == rb.keys() and all(np.array_equal(ra[k], rb[k]) for k in ra)

    for i in range(n):
        same_obs(outs[i][0], refs[i][0])
    done_at, mutated = None, False
    for t in range(p.max_time_steps):
        acts = []
        for v, s in zip(batch.envs, singles):
            aa, ai = v._b.allocate(interval, True)
            sa, si = s._b.allocate(interval, True)
            k = sa.shape[1]
            assert np.array_equal(aa[:, :k], sa) and np.all(aa[:, k:] == -1) and np.array_equal(ai[:, :k], si)
            acts.append({v.agents_obj[int(a)].name: int(j) for a, j in zip(aa[0], ai[0]) if a >= 0})
        if t == 7:  # an out-of-step mutator through view 1 only (and the same on its single-env twin)
            rets = []
            for e in (batch.envs[1], singles[1]):
                uav = e.get_live_agents()[0]
                open_tasks = [x for x in e.tasks if x.status != 2]
                assert open_tasks
                rets.append((uav.allocate(open_tasks[-1], e.time_steps), [x.id for x in uav.tasks]))
            assert rets[0] == rets[1]
            mutated = True
        before = dict(backend.launches)
        outs = batch.step(acts)
        assert backend.launches["step"] == before["step"] + 1 and backend.launches["observe"] <= before["observe"] + 1
        for i, (v, s) in enumerate(zip(batch.envs, singles)):
            o, r, te, tr, info = s.step(acts[i])
            bo, br, bte, btr, binfo = outs[i]
            same_obs(bo, o)
            assert br == r and bte == te and btr == tr and binfo["events"] == info["events"] and binfo["selected"] == info["selected"]
            assert [x.id for x in v.last_tasks_info] == [x.id for x in s.last_tasks_info] and [x.id for x in v.tasks] == [x.id for x in s.tasks]
            assert v.agent_visibility_map() == s.agent_visibility_map() and v.time_steps == s.time_steps == t + 1
            for a, b in zip(v.agents_obj, s.agents_obj):
                assert a.name == b.name and a.state == b.state and np.array_equal(a.position, b.position) and [x.id for x in a.tasks] == [x.id for x in b.tasks]
            if t % 25 == 0:
                for x, y in zip(v.tasks, s.tasks):
                    assert x.status == y.status and np.array_equal(x.position, y.position) and np.array_equal(x.currentReqs, y.currentReqs)
            if all(tr.values()) or all(te.values()):
                assert binfo["metrics"] == info["metrics"]
        if all(all(o[3].values()) or all(o[2].values()) for o in outs):
            done_at = t
            break
    assert done_at is not None and mutated
    # whole-handle fetches per step stay O(fields), not O(fields x envs): the n views share them
    assert backend.launches["get"] <= 16 * (done_at + 2) + 16 * n
    batch.close()


@pytest.mark.skipif(not os.path.isdir("/root/reference/mUAV_TA"), reason="reference checkout not present (GPU box)")
def test_unseeded_reset_fixed_seed_and_seed_method_follow_the_reference():
    """DroneEnv.py:518-531: reset() without a seed draws one from the GLOBAL `random` module (so `random.seed(s)` in front of a trainer makes its
    unseeded episodes reproducible — the reference's trainers rely on nothing else), `fixed_seed != -1` overrides every reset's seed, and
    `seed(s)` is `reset(seed=s)`.  The facade must do the same: same seed drawn, same episode."""
    import random

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refshim
    refshim.install()
    from mUAV_TA.DroneEnv import MultiUAVEnv as RefEnv
    from experiments.paper_eval import make_config

    def pair(case, **over):
        cfg = make_config(CASE_SPECS[case], dict(WPS_ENV_FLAGS))
        for k, v in over.items():
            setattr(cfg, k, v)
        return RefEnv(cfg), MultiUAVEnv(cfg, backend=OracleBackend(params_from_config(cfg, None)))

    def same_state(ref, fac):
        assert ref._seed == fac._seed
        assert [a.name for a in ref.agents_obj] == [a.name for a in fac.agents_obj]
        assert np.array_equal(np.array([a.position for a in ref.agents_obj]), np.array([a.position for a in fac.agents_obj]))
        assert [(t.id, t.type, tuple(t.position)) for t in ref.tasks] == [(t.id, t.type, tuple(t.position)) for t in fac.tasks]

    ref, fac = pair("WPS_hard")
    random.seed(99); ref.reset(); ref.reset(); s_ref = ref._seed
    random.seed(99); fac.reset(); fac.reset()
    assert fac._seed == s_ref
    random.seed(99); ref.reset(); ref.reset()  # (ref and fac now sit behind the same two draws)
    same_state(ref, fac)
    ref.seed(31); fac.seed(31)
    same_state(ref, fac)
    assert ref._seed == 31
    ref, fac = pair("WPS_easy", fixed_seed=7)
    assert fac.fixed_seed == 7
    ref.reset(seed=3); fac.reset(seed=3)
    same_state(ref, fac)
    assert fac._seed == 7
    random.seed(1); ref.reset(); fac.reset()
    same_state(ref, fac)
    assert fac._seed == 7


@pytest.mark.skipif(not os.path.isdir("/root/reference/mUAV_TA"), reason="reference checkout not present (GPU box)")
def test_every_env_attribute_the_reference_callers_read_exists_and_echoes_the_configuration():
    """A planner written against the reference reads the env through `env.<name>` and `getattr(env, "<name>", default)`; a name the facade lacks
    is an AttributeError in the first form and a SILENTLY different input in the second (build_rah_state's `burst_mode`: found by running
    train_rah.py over the facade).  Every such name in the reference's experiments/ and TaskAllocation/ (turn-based TBTA / tianshou and the
    legacy genetic-algorithm attributes aside — those paths are out of scope) must exist on the facade, and the configuration the
    reference env echoes on itself (DroneEnv.py:101-201) must read the same there."""
    import re

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refshim
    refshim.install()
    from mUAV_TA.DroneEnv import MultiUAVEnv as RefEnv
    from experiments.paper_eval import make_config

    names = set()
    for top in ("experiments", "TaskAllocation"):
        for d, _, files in os.walk(os.path.join("/root/reference", top)):
            for f in files:
                if f.endswith(".py"):
                    text = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
                    names |= set(re.findall(r"\benv\.([A-Za-z_][A-Za-z_0-9]*)", text))
                    names |= set(re.findall(r"getattr\((?:self\.)?env, *['\"]([A-Za-z_0-9]+)['\"]", text))
    for f in ("main.py", "benchmark.py"):  # the two scripts at the root of the checkout (main.py calls its env `worldModel`)
        names |= set(re.findall(r"\b(?:worldModel|env)\.([A-Za-z_][A-Za-z_0-9]*)", open(os.path.join("/root/reference", f), encoding="utf-8", errors="replace").read()))
    assert {"close", "get_initial_state", "current_agent"} <= names
    out_of_scope = {"agent_selector", "agent_selection", "pettingzoo_env", "last", "NUM_DRONES", "NUM_TARGETS", "targets", "drone_tasks",
                    "unallocated_tasks"}  # (unallocated_tasks: read by main.py's CTBTA branch only, a tianshou policy)
    assert len(names) > 40
    for case in ("WPS_hard", "WPS_escort", "D3_combined"):
        cfg = make_config(CASE_SPECS[case], dict(WPS_ENV_FLAGS))
        ref, fac = RefEnv(cfg), MultiUAVEnv(cfg, backend=OracleBackend(params_from_config(cfg, None)))
        ref.reset(seed=4); fac.reset(seed=4)
        missing = sorted(n for n in names - out_of_scope if hasattr(ref, n) and not hasattr(fac, n))
        assert not missing, (case, missing)
        for n in ("max_time_steps", "simulation_frame_rate", "info", "action_mode", "agents_config", "tasks_config", "threats_list", "random_init_pos", "num_obstacles",
                  "hidden_obstacles", "multiple_tasks_per_agent", "multiple_agents_per_task", "fail_rate", "early_terminate", "capability_mask", "saturate_mask",
                  "reward_weights", "arrival_rate", "include_time_windows", "dynamic_idle_penalty", "sense_radius", "threat_delay", "hard_windows", "window_length",
                  "burst_mode", "burst_size", "miss_penalty", "on_time_bonus", "dual_region_bursts", "share_knowledge", "commit_horizon", "reassign_penalty",
                  "escort_enabled", "escort_radius", "escort_requirement", "escort_intercept_radius", "mutual_support_radius", "escort_agent_types", "fixed_seed",
                  "n_agents", "max_agents", "n_tasks", "max_tasks", "max_coord", "area_width", "area_height", "possible_agents", "render_enabled", "render_speed"):
            r, f = getattr(ref, n), getattr(fac, n)
            assert r == f and type(r) is type(f), (case, n, r, f)

    # the object views: every attribute the reference's planners read on an agent / task exists (swarm_gap.py:104 reads `has_capability`)
    a_ref, a_fac = ref.agents_obj[0], fac.agents_obj[0]
    for n in ("has_capability", "altitude", "task_finished", "fail_multiplier", "max_speed", "engage_range", "attackCap", "type", "typeIdx", "name", "id", "state", "commit_until"):
        assert getattr(a_ref, n) == getattr(a_fac, n), n
    assert a_fac.env is fac
    t_ref, t_fac = ref.tasks[1], fac.tasks[1]
    for n in ("info", "max_time_steps", "type", "typeIdx", "status", "id", "task_duration"):
        assert getattr(t_ref, n) == getattr(t_fac, n), n

    # get_initial_state (DroneEnv.py:764-771; benchmark.py:49 and main.py:89 call it after every reset): the same keys, the tasks as detached copies
    # that keep the values of the moment while the env moves on, as the reference's deep copies do
    i_ref, i_fac = ref.get_initial_state(), fac.get_initial_state()
    assert list(i_ref) == list(i_fac) and i_fac["agents"] == i_ref["agents"] and i_fac["agents"] is not fac.agents
    assert i_fac["quality_table"] is None and i_ref["quality_table"] is None and i_fac["events"] == [] == i_ref["events"]
    assert len(i_fac["tasks"]) == len(i_ref["tasks"])

    def frozen(tasks):
        return [(t.id, t.type, t.typeIdx, t.status, tuple(np.asarray(t.position, dtype=float)), tuple(t.orgReqs), tuple(t.currentReqs), tuple(t.allocatedReqs),
                 t.initTime, t.doneTime, t.created_at, t.task_duration, t.required_agents, t.kind, getattr(t, "hard_deadline", None)) for t in tasks]

    at_reset = frozen(i_ref["tasks"])
    assert frozen(i_fac["tasks"]) == at_reset
    for _ in range(30):
        acts = {a.name: 1 + (k % (len(ref.last_tasks_info) - 1)) for k, a in enumerate(ref.get_live_agents()[:3])}
        ref.step(dict(acts)); fac.step(dict(acts))
    assert frozen(i_fac["tasks"]) == at_reset == frozen(i_ref["tasks"])          # the copies did not move ...
    assert frozen(fac.tasks) == frozen(ref.tasks) and frozen(fac.tasks) != at_reset  # ... the env did


@pytest.mark.skipif(not os.path.isdir("/root/reference/mUAV_TA"), reason="reference checkout not present (GPU box)")
def test_writing_multiple_tasks_per_agent_after_reset_as_main_py_does_gives_the_reference_episode():
    """The reference's main.py:130-141 builds its env with multiple_tasks_per_agent=False and assigns True on the env object right after every reset()
    (Greedy / Swarm-GAP / CBBA hand whole task LISTS to an agent).  On the device the switch is a parameter of the handle: the write re-creates the handle and
    resets it with the episode's seed (reset never reads the switch), after checking field by field that nothing happened since reset.  The episode that
    follows must be the reference's, list-valued actions queued in full; the same write after a step, or on an env that was given a backend instance, raises
    instead of passing silently with the old behaviour; multiple_agents_per_task=False (dead code in the reference) raises."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refshim
    refshim.install()
    from mUAV_TA.DroneEnv import MultiUAVEnv as RefEnv
    from experiments.paper_eval import make_config

    spec = dict(CASE_SPECS["static_strike"])
    cfg = make_config(spec, {})
    cfg.multiple_tasks_per_agent = False
    ref, fac = RefEnv(cfg), MultiUAVEnv(cfg, backend_factory=OracleBackend)
    assert ref.multiple_tasks_per_agent is False and fac.multiple_tasks_per_agent is False
    made = [fac.backend]
    for seed in (5, 6):
        ref.reset(seed=seed); fac.reset(seed=seed)
        views = (fac.agents_obj[0], fac.tasks[1], fac.last_tasks_info)
        ref.multiple_tasks_per_agent = True; ref.multiple_agents_per_task = True   # main.py:131-132
        fac.multiple_tasks_per_agent = True; fac.multiple_agents_per_task = True
        assert fac.multiple_tasks_per_agent is True
        if fac.backend is not made[-1]:
            made.append(fac.backend)
        assert (fac.agents_obj[0], fac.tasks[1], fac.last_tasks_info) == views and fac.agents_obj[0] is views[0]  # (the views a planner holds stay the env's views)
        rng = np.random.default_rng(seed)
        multi = 0
        for t in range(60):
            n_open = len(ref.last_tasks_info)
            acts = {a.name: [int(x) for x in rng.integers(1, n_open, size=int(rng.integers(1, 4)))] for a in ref.get_live_agents()[: 1 + t % 3]} if n_open > 1 and t % 4 == 0 else {}
            r_ref, r_fac = ref.step({k: list(v) for k, v in acts.items()}), fac.step({k: list(v) for k, v in acts.items()})
            assert r_ref[1] == r_fac[1] and r_ref[2] == r_fac[2] and r_ref[3] == r_fac[3], (seed, t)
            assert [[x.id for x in a.tasks] for a in ref.agents_obj] == [[x.id for x in a.tasks] for a in fac.agents_obj], (seed, t)
            assert np.array_equal(np.array([a.position for a in ref.agents_obj]), np.array([a.position for a in fac.agents_obj])), (seed, t)
            multi += sum(len(a.tasks) > 1 for a in fac.agents_obj)
        assert multi > 0  # (queues longer than one task: the switch is on)
    assert len(made) == 2  # one re-creation, at the first write; the later writes keep the value

    with pytest.raises(ValueError, match="right after reset"):  # stepped since reset
        fac.multiple_tasks_per_agent = False
    fac.reset(seed=9)
    fac.agents_obj[0].position = np.array([10.0, 20.0])         # mutated since reset
    with pytest.raises(ValueError, match="changed since reset"):
        fac.multiple_tasks_per_agent = False
    assert fac.multiple_tasks_per_agent is True
    with pytest.raises(ValueError, match="dead code in the reference"):
        fac.multiple_agents_per_task = False
    inst = MultiUAVEnv(cfg, backend=OracleBackend(params_from_config(cfg, None)))
    inst.reset(seed=1)
    inst.multiple_tasks_per_agent = False                       # (keeps the value)
    with pytest.raises(ValueError, match="backend instance"):
        inst.multiple_tasks_per_agent = True

