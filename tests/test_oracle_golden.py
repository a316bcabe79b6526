"""Pins the CPU oracle (oracle/muavta_oracle.cpp) against vectors captured from the reference itself
(tools/gen_golden.py -> tests/golden/*.npz).  Integer/index/mask data must be identical; every f64
must be bit-identical too (the oracle reproduces the reference's operation order), which is stricter
than the 1e-5 the north-star asks for."""
import glob
import os

import numpy as np
import pytest

import orc
from cases import params_of
from muavta_amd.params import METRIC_KEYS, params_for_case

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TRACES = sorted(glob.glob(os.path.join(GOLDEN, "trace_*.npz")))
METRICS = sorted(glob.glob(os.path.join(GOLDEN, "metrics_*.npz")))


def _case_of(path, prefix):
    name = os.path.basename(path)[len(prefix):-4]
    return name


def check_state(e, g, s, tag):
    """Compare oracle state after step s with the golden snapshot row s."""
    rows, caps, q = e.agents()
    A = rows.shape[0]
    assert np.array_equal(rows[:, 0:2], g["pos"][s]), f"{tag} step {s}: agent positions"
    assert np.array_equal(rows[:, 2].astype(int), g["state"][s].astype(int)), f"{tag} step {s}: agent state"
    assert np.array_equal(rows[:, 3].astype(int), g["head"][s]), f"{tag} step {s}: head task"
    gq = g["queue"][s]
    assert np.array_equal(q[:, : gq.shape[1]], gq), f"{tag} step {s}: queues"
    assert np.array_equal(rows[:, 5], g["nft"][s]), f"{tag} step {s}: next_free_time"
    assert np.array_equal(rows[:, 6:8], g["nfp"][s]), f"{tag} step {s}: next_free_position"
    assert np.array_equal(caps, g["caps"][s]), f"{tag} step {s}: caps"
    assert np.array_equal(rows[:, 8].astype(int), g["attack_cap"][s]), f"{tag} step {s}: attackCap"
    assert np.array_equal(rows[:, 9].astype(int), g["task_start"][s]), f"{tag} step {s}: task_start"
    assert np.array_equal(rows[:, 10].astype(int), g["re_eval"][s].astype(int)), f"{tag} step {s}: re_eval"
    assert np.array_equal(rows[:, 11].astype(int), g["last_task"][s]), f"{tag} step {s}: last_task"
    assert np.array_equal(rows[:, 15], g["agent_dist"][s]), f"{tag} step {s}: agent_distances"
    trow, reqs = e.tasks()
    nt = trow.shape[0]
    gs = g["t_status"][s]
    assert np.all(gs[nt:] == -9), f"{tag} step {s}: oracle has fewer tasks than the reference"
    assert np.all(gs[1:nt] != -9), f"{tag} step {s}: oracle has more tasks than the reference"
    assert np.array_equal(trow[1:, 0].astype(int), gs[1:nt].astype(int)), f"{tag} step {s}: task status"
    assert np.array_equal(trow[1:, 1:3], g["t_pos"][s][1:nt]), f"{tag} step {s}: task positions"
    assert np.array_equal(reqs[1:, 0], g["t_cur"][s][1:nt]), f"{tag} step {s}: currentReqs"
    assert np.array_equal(reqs[1:, 1], g["t_alloc"][s][1:nt]), f"{tag} step {s}: allocatedReqs"
    assert np.array_equal(reqs[1:, 2], g["t_done"][s][1:nt]), f"{tag} step {s}: doneReqs"
    assert np.array_equal(trow[1:, 3], g["t_init"][s][1:nt]), f"{tag} step {s}: initTime"
    assert np.array_equal(trow[1:, 4], g["t_donetime"][s][1:nt]), f"{tag} step {s}: doneTime"
    assert np.array_equal(trow[1:, 5].astype(int), g["t_ndet"][s][1:nt].astype(int)), f"{tag} step {s}: allocationDetails"
    st = g["t_static"][1:nt]
    assert np.array_equal(trow[1:, 6:11].astype(int), st), f"{tag} step {s}: task type/deadline/created/required/escort"
    NT = int(g["n_task_ids"])
    known = np.unpackbits(g["known"][s], axis=-1)[:, :NT].astype(bool)
    assert np.array_equal(e.known()[:, 1:], known[:, 1:nt]), f"{tag} step {s}: known-task masks"
    th = e.threats()
    assert np.array_equal(th[:, 0].astype(int), g["h_status"][s].astype(int)), f"{tag} step {s}: threat status"
    act = th[:, 0] != -9
    assert np.array_equal(th[act, 1:3], g["h_pos"][s][act]), f"{tag} step {s}: threat positions"
    assert np.array_equal(th[act, 3].astype(int), g["h_target"][s][act]), f"{tag} step {s}: threat targets"
    assert np.array_equal(th[act, 5].astype(int), g["h_acap"][s][act]), f"{tag} step {s}: threat attackCap"
    sc = e.scalars()
    gsc = g["scalars"][s]
    # golden scalar order: SCALARS list of tools/gen_golden.py + pending_reset, n_reached, n_pending
    ours = [sc[2], sc[3], sc[4], sc[5], sc[6], sc[7], sc[8], sc[9], sc[10], sc[11], sc[12], sc[13], sc[14], sc[15],
            sc[16], sc[17], sc[18], sc[19], sc[20], sc[21], sc[22]]
    d = e.dims()
    ours += [d["pending_reset"], d["n_reached"], d["n_pending"]]
    assert np.array_equal(np.array(ours, dtype=np.float64), gsc), f"{tag} step {s}: scalar counters {ours} vs {gsc}"
    assert sc[1] == g["reward"][s], f"{tag} step {s}: reward"
    lo, hi = g["open_ptr"][s], g["open_ptr"][s + 1]
    assert np.array_equal(e.open_ids(), g["open_ids"][lo:hi]), f"{tag} step {s}: last_tasks_info order"
    ti, legal, pad, ag, fl = e.observe()
    T = int(g["max_tasks"])
    assert np.array_equal(ti, g["obs_tasks"][s]), f"{tag} step {s}: obs tasks_info"
    glegal = np.unpackbits(g["obs_legal"][s], axis=-1)[:, :T].astype(bool)
    assert np.array_equal(legal, glegal), f"{tag} step {s}: legal_mask"
    assert np.array_equal(ag, g["obs_agent"][s]), f"{tag} step {s}: obs agent rows"
    assert np.array_equal(fl, g["obs_flags"][s]), f"{tag} step {s}: event_flags"


def check_trace(g, name, params, seed):
    """One reference episode (the arrays tools/gen_golden.py::run_episode(full=True) returns) against the oracle: every step's
    plan, LSAP calls, drained events, state and observation, then the final metrics."""
    interval = int(g["interval"])
    e = orc.OracleEnv(params)
    e.reset(seed)
    rows, _, _ = e.agents()
    assert np.array_equal(rows[:, 12].astype(int), g["agent_type"])
    assert np.array_equal(rows[:, 13].astype(int), g["agent_name_idx"])
    assert np.array_equal(rows[:, 14].astype(int), g["fail_event"])
    check_state(e, g, 0, name)
    S = g["pos"].shape[0] - 1
    acts, evs = g["actions"], g["events"]
    lsap_i = 0
    lsap_off = 0
    rc_off = 0
    for s in range(S):
        aa, ai = e.allocate(interval, 1)
        ga = acts[acts[:, 0] == s]
        assert np.array_equal(aa, ga[:, 1]), f"{name} t={s}: assigned agents {aa} vs {ga[:, 1]}"
        assert np.array_equal(ai, ga[:, 3]), f"{name} t={s}: assigned open-list indices"
        assert np.array_equal(e.last_actions()[:, 1], ga[:, 2]), f"{name} t={s}: assigned task ids"
        shapes, costs, rws, cls = e.lsap_calls()
        off = ro = 0
        for k in range(len(shapes)):
            assert int(g["lsap_step"][lsap_i]) == s
            nr, nc = shapes[k]
            assert (nr, nc) == tuple(g["lsap_shape"][lsap_i]), f"{name} t={s}: LSAP shape"
            m = min(nr, nc)
            assert np.array_equal(costs[off:off + nr * nc], g["lsap_cost"][lsap_off:lsap_off + nr * nc]), f"{name} t={s}: LSAP cost matrix"
            assert np.array_equal(rws[ro:ro + m], g["lsap_row"][rc_off:rc_off + m])
            assert np.array_equal(cls[ro:ro + m], g["lsap_col"][rc_off:rc_off + m])
            off += nr * nc; ro += m; lsap_off += nr * nc; rc_off += m; lsap_i += 1
        e.step(aa, ai)
        ge = evs[evs[:, 0] == s + 1][:, 1:]
        assert np.array_equal(e.events(), ge), f"{name} t={s + 1}: drained events"
        check_state(e, g, s + 1, name)
    assert lsap_i == len(g["lsap_step"])
    assert np.array_equal(e.metrics(), g["metrics"]), f"{name}: final metrics"
    assert e.dims()["n_replans"] == int(g["n_replans"])


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[6:-4] for p in TRACES])
def test_trace_bit_exact(path):
    g = np.load(path)
    name = os.path.basename(path)[6:-4]
    case, seed = name.rsplit("_s", 1)
    check_trace(g, name, params_of(case), int(seed))


LIST_TRACES = sorted(glob.glob(os.path.join(GOLDEN, "lists_*.npz")))


def lists_params(path, **tiles):
    """Params of a list-valued-action trace (tools/gen_golden.py --lists): the registry case under the WPS flags, with the
    trace's multiple_tasks_per_agent."""
    from muavta_amd.params import params_from_config
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS
    g = np.load(path)
    case = os.path.basename(path)[6:-4].rsplit("_s", 1)[0]
    if case.startswith("WIDE"):  # an action-driven episode of a wide-fuzz configuration (tests/fuzz_reference.py --pin-scored)
        return g, params_of(case, **tiles)
    flags = dict(WPS_ENV_FLAGS)
    flags["multiple_tasks_per_agent"] = bool(int(g["multi"]))
    ta, tt, th = TILES[case]
    return g, params_from_config(CASE_SPECS[case], flags, tile_agents=tiles.get("tile_agents", ta), tile_tasks=tiles.get("tile_tasks", tt),
                                 tile_threats=tiles.get("tile_threats", th))


@pytest.mark.parametrize("path", LIST_TRACES, ids=[os.path.basename(p)[6:-4] for p in LIST_TRACES])
def test_list_valued_actions_trace_bit_exact(path):
    """env.step({agent: [index, ...]}) of the reference (DroneEnv.py:813-838; up to 76 items in one step, repeated tasks,
    indices beyond the open list, dead agents, with and without multiple_tasks_per_agent): the oracle fed the same flattened
    items reproduces every step's state, reward, events and observation."""
    g, p = lists_params(path)
    name = os.path.basename(path)[:-4]
    assert "WIDE" in name or np.bincount(g["actions"][:, 0]).max() > 32  # (the random-list traces go past every tile's action_cap)
    check_lists(g, name, p)


def check_lists(g, name, p):
    """an action-driven reference episode (tools/gen_golden.py::drive_with_actions) against the oracle fed the same flattened items"""
    e = orc.OracleEnv(p)
    e.reset(int(g["seed"]))
    check_state(e, g, 0, name)
    acts, evs = g["actions"], g["events"]
    for s in range(g["pos"].shape[0] - 1):
        ga = acts[acts[:, 0] == s]
        e.step(ga[:, 1].astype(np.int32), ga[:, 2].astype(np.int32))
        assert np.array_equal(e.events(), evs[evs[:, 0] == s + 1][:, 1:]), f"{name} t={s + 1}: drained events"
        check_state(e, g, s + 1, name)


@pytest.mark.parametrize("path", METRICS, ids=[os.path.basename(p)[8:-4] for p in METRICS])
def test_metrics_many_seeds(path):
    g = np.load(path)
    case = os.path.basename(path)[8:-4]
    assert list(g["keys"]) == METRIC_KEYS
    interval = int(g["interval"])
    e = orc.OracleEnv(params_for_case(case))
    for seed, want in enumerate(g["metrics"]):
        n = e.rollout(seed, 150, interval, 1)
        assert n == 150
        got = e.metrics()
        assert np.array_equal(got, want), f"{case} seed {seed}: {dict(zip(METRIC_KEYS, got - want))}"
        assert e.dims()["n_replans"] == int(g["n_replans"][seed])


def test_lsap_known_answers():
    g = np.load(os.path.join(GOLDEN, "lsap_cases.npz"))
    off = ro = 0
    for nr, nc in g["shape"]:
        c = g["cost"][off:off + nr * nc].reshape(nr, nc)
        m = min(nr, nc)
        r, cc = orc.lsap(c)
        assert np.array_equal(r, g["row"][ro:ro + m]) and np.array_equal(cc, g["col"][ro:ro + m]), (nr, nc)
        off += nr * nc; ro += m


def test_cpython_random_known_answers():
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, "mt_kat.npz"))
    L = orc.lib()
    for i, seed in enumerate(g["seeds"]):
        r = C.c_void_p(L.orc_rng_new(C.c_uint64(int(seed))))
        got = [L.orc_rng_random(r) for _ in range(8)]
        assert got == list(g["random"][i][:8])
        got = [L.orc_rng_randint(r, C.c_int64(0), C.c_int64(2**63 - 1)) for _ in range(4)]
        assert got == [int(x) for x in g["randint63"][i]]
        got = [L.orc_rng_randint(r, C.c_int64(1), C.c_int64(150)) for _ in range(8)] + \
              [L.orc_rng_randint(r, C.c_int64(120), C.c_int64(1080)) for _ in range(4)]
        assert got == list(g["randint"][i])
        got = [L.orc_rng_uniform(r, C.c_double(3.5), C.c_double(1196.25)) for _ in range(4)]
        assert got == list(g["uniform"][i])
        got = [L.orc_rng_randbelow(r, C.c_uint64(2)) for _ in range(8)] + [L.orc_rng_randbelow(r, C.c_uint64(3)) for _ in range(8)]
        assert got == list(g["choice"][i])
        x = np.arange(16, dtype=np.int64)
        L.orc_rng_shuffle(r, x.ctypes.data_as(C.c_void_p), 16)
        assert list(x) == list(g["shuffle16"][i])
        for _ in range(700):
            L.orc_rng_random(r)
        assert L.orc_rng_random(r) == g["random"][i][8]
        L.orc_rng_free(r)


def test_numpy_bit_patterns():
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, "numpy_kat.npz"))
    L = orc.lib()
    v = g["vec"]
    got = np.array([L.orc_norm2(C.c_double(x), C.c_double(y)) for x, y in v])
    assert np.array_equal(got, g["norm_1d"])          # np.linalg.norm(1-D) == sqrt(fma(y, y, x*x))
    assert np.array_equal(np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]), g["norm_axis1"])
    for n in (4, 8, 14, 16, 24, 40, 64):
        d = g[f"sum{n}_in"]
        got = np.array([L.orc_np_sum(np.ascontiguousarray(r).ctypes.data_as(C.c_void_p), n) for r in d])
        assert np.array_equal(got, g[f"sum{n}_out"]), n


# ---- next row: Urgency-Pair (engineered edge scores feeding the same Hungarian) --------------------------
URG_TRACES = sorted(glob.glob(os.path.join(GOLDEN, "urgpair_trace_*.npz")))
URG_METRICS = sorted(glob.glob(os.path.join(GOLDEN, "urgpair_metrics_*.npz")))


@pytest.mark.parametrize("path", URG_TRACES, ids=[os.path.basename(p)[14:-4] for p in URG_TRACES])
def test_urgency_pair_trace(path):
    g = np.load(path)
    name = os.path.basename(path)[14:-4]
    case, seed = name.rsplit("_s", 1)
    e = orc.OracleEnv(params_for_case(case))
    e.reset(int(seed))
    acts = g["actions"]
    li = off = 0
    for s in range(150):
        aa, ai = e.allocate_mode(20, 1, 1)
        ga = acts[acts[:, 0] == s]
        assert np.array_equal(aa, ga[:, 1]) and np.array_equal(ai, ga[:, 3]), f"{name} t={s}: {aa} vs {ga[:, 1]}"
        assert np.array_equal(e.last_actions()[:, 1], ga[:, 2])
        shapes, costs, _, _ = e.lsap_calls()
        o = 0
        for k in range(len(shapes)):
            assert int(g["lsap_step"][li]) == s and tuple(shapes[k]) == tuple(g["lsap_shape"][li])
            n = int(shapes[k][0] * shapes[k][1])
            assert np.array_equal(costs[o:o + n], g["lsap_cost"][off:off + n]), f"{name} t={s}: scored cost matrix"
            o += n; off += n; li += 1
        e.step(aa, ai)
    assert li == len(g["lsap_step"])
    assert np.array_equal(e.metrics(), g["metrics"]) and e.dims()["n_replans"] == int(g["n_replans"])


@pytest.mark.parametrize("path", URG_METRICS, ids=[os.path.basename(p)[16:-4] for p in URG_METRICS])
def test_urgency_pair_metrics(path):
    g = np.load(path)
    case = os.path.basename(path)[16:-4]
    e = orc.OracleEnv(params_for_case(case))
    for seed, want in enumerate(g["metrics"]):
        assert e.rollout_mode(seed, 150, 20, 1, 1) == 150
        assert np.array_equal(e.metrics(), want), f"{case} seed {seed}"
        assert e.dims()["n_replans"] == int(g["n_replans"][seed])


# ---- next row: Urgency-Coalition (AttentionEscort.py:714-767 under escort_eval._should_replan) ----
URGCOAL_TRACES = [("WPS_escort", 0), ("WPS_escort", 1), ("WPS_escort24", 0), ("WPS_hard", 0)]


@pytest.mark.parametrize("case,seed", URGCOAL_TRACES)
def test_urgency_coalition_stepwise_vs_reference(case, seed):
    g = np.load(os.path.join(GOLDEN, f"urgcoal_trace_{case}_s{seed}.npz"))
    e = orc.OracleEnv(params_for_case(case))
    e.reset(seed)
    interval = int(g["interval"])
    acts = g["actions"]
    lsap_step, lsap_shape, lsap_cost = g["lsap_step"], g["lsap_shape"], g["lsap_cost"]
    cost_off = np.concatenate([[0], np.cumsum(lsap_shape[:, 0] * lsap_shape[:, 1])])
    for t in range(150):
        aa, ai = e.allocate_mode(interval, True, 2)
        exp = acts[acts[:, 0] == t]
        assert sorted(zip(aa.tolist(), ai.tolist())) == sorted(zip(exp[:, 1].tolist(), exp[:, 3].tolist())), (case, seed, t)
        shapes, costs, _, _ = e.lsap_calls()
        k = np.nonzero(lsap_step == t)[0]
        assert len(shapes) == len(k), (case, seed, t)
        off = 0
        for j, kk in enumerate(k):
            nr, nc = shapes[j]
            assert (nr, nc) == tuple(lsap_shape[kk])
            np.testing.assert_array_equal(costs[off:off + nr * nc], lsap_cost[cost_off[kk]:cost_off[kk + 1]])
            off += nr * nc
        np.testing.assert_array_equal(e.agent_commit_until(), g["commit_until"][t])
        e.step(aa, ai)
    np.testing.assert_array_equal(e.metrics(), g["metrics"])


@pytest.mark.parametrize("case", ["WPS_escort", "WPS_escort24", "WPS_hard"])
def test_urgency_coalition_metrics_vs_reference(case):
    g = np.load(os.path.join(GOLDEN, f"urgcoal_metrics_{case}.npz"))
    e = orc.OracleEnv(params_for_case(case))
    for s in range(g["metrics"].shape[0]):
        e.rollout_mode(s, 150, int(g["interval"]), True, 2)
        np.testing.assert_array_equal(e.metrics(), g["metrics"][s], err_msg=f"{case} seed {s}")


# ---- next row: token builders (build_pair_tokens / raw / build_escort_tokens) ----------------------------------
TOKEN_FILES = sorted(glob.glob(os.path.join(GOLDEN, "tokens_*.npz")))


from tokcheck import check_tokens  # noqa: E402


@pytest.mark.parametrize("path", TOKEN_FILES, ids=[os.path.basename(p)[7:-4] for p in TOKEN_FILES])
def test_token_builders_vs_reference(path):
    g = np.load(path)
    case = os.path.basename(path)[7:-4]
    e = orc.OracleEnv(params_for_case(case))
    e.reset(int(g["seed"]))
    mode = 2 if str(g["driver"]) == "urgcoal" else 0
    steps = g["step"].tolist()
    for t in range(150):
        aa, ai = e.allocate_mode(int(g["interval"]), 1, mode)
        if t in steps:
            check_tokens(e.tokens, g, steps.index(t), f"{case} t={t}")
        e.step(aa, ai)


# ---- next row: the trainers' imitation-learning data loop (expert pairs, tokens, _expert_mask, RL step reward) ----
IL_FILES = sorted(glob.glob(os.path.join(GOLDEN, "il_*.npz")))


@pytest.mark.parametrize("path", IL_FILES, ids=[os.path.basename(p)[3:-4] for p in IL_FILES])
def test_il_loop_vs_reference(path):
    g = np.load(path)
    case = os.path.basename(path)[3:-4]
    e = orc.OracleEnv(params_for_case(case))
    e.reset(int(g["seed"]))
    steps, k = g["step"].tolist(), 0
    assert e.metrics()[4] == g["s_wps"][0]
    for t in range(150):
        aa, ai = e.allocate_mode(20, 0, 3)  # Global-Hungarian expert, force=True, under the trainer's gate
        want_pairs = g["pairs"][g["pairs"][:, 0] == t][:, 1:]
        if g["replanned"][t]:
            assert t == steps[k]
            tok = e.tokens(0, 32, 16)
            assert np.array_equal(tok["expert_mask"], g["mask"][k]), f"{case} t={t}: expert mask"
            assert np.array_equal(tok["task_feats"], g["tf"][k]) and np.array_equal(tok["agent_feats"], g["af"][k])
            assert np.array_equal(tok["edge_valid"], g["ev"][k]) and np.array_equal(tok["task_ids"], g["tid"][k]) and np.array_equal(tok["agent_ids"], g["aid"][k])
            k += 1
        else:
            assert len(aa) == 0
        assert sorted(map(tuple, e.last_actions().tolist())) == sorted(map(tuple, want_pairs.tolist())), f"{case} t={t}: expert pairs"
        e.step(aa, ai)
        assert e.metrics()[4] == g["s_wps"][t + 1]  # RL step reward = (s_wps[t+1] - s_wps[t]) / 20
    assert k == len(steps) and np.array_equal(e.metrics(), g["metrics"])


# ---- caller-supplied planner inputs (a25 / f3 RL half): reference planners driven with seeded network outputs -------------
GATE_FORCE, GATE_TRAINER, GATE_ESCORT, GATE_ALLOCATOR = 0, 1, 2, 3
SC_EDGE_VALID_ONLY, SC_FULL_TASK_LIST, SC_COMMIT = 1, 2, 4


def _check_lsap(e, g, t, where):
    """every LSAP call of the oracle's last allocate equals the reference's calls at step t: shapes, every cost, rows, cols"""
    shapes, costs, rows, cols = e.lsap_calls()
    idx = np.nonzero(g["lsap_step"] == t)[0]
    assert len(idx) == len(shapes), f"{where} t={t}: {len(shapes)} LSAP calls, reference {len(idx)}"
    if not len(idx):
        return
    sizes = g["lsap_shape"][:, 0] * g["lsap_shape"][:, 1]
    mins = g["lsap_shape"].min(axis=1)
    c0, r0 = int(sizes[:idx[0]].sum()), int(mins[:idx[0]].sum())
    assert np.array_equal(shapes, g["lsap_shape"][idx]), f"{where} t={t}: LSAP shapes"
    n, m = int(sizes[idx].sum()), int(mins[idx].sum())
    assert np.array_equal(costs, g["lsap_cost"][c0:c0 + n]), f"{where} t={t}: cost matrices"
    assert np.array_equal(rows, g["lsap_row"][r0:r0 + m]) and np.array_equal(cols, g["lsap_col"][r0:r0 + m]), f"{where} t={t}: assignment"


def _acts_at(g, t):
    a = g["actions"][g["actions"][:, 0] == t][:, 1:]
    return a[np.argsort(a[:, 0], kind="stable")]


def check_rl(g, case, params):
    """run_rl_episode (experiments/train_pair_cost.py:132-156) with PairCostHybrid.plan(scores=seeded): scored cost matrices,
    assignments, _selected_mask, actions, tokens / next tokens, step rewards, done, final metrics"""
    kind = 1 if int(g["raw"]) else 0
    interval = int(g["interval"]) if "interval" in g else 20
    e = orc.OracleEnv(params)
    e.reset(int(g["seed"]))
    steps, k = g["step"].tolist(), 0
    assert e.metrics()[4] == g["s_wps"][0]
    pending = None
    for t in range(len(g["replanned"])):
        tok = e.tokens(kind, 32, 16)
        if pending is not None:  # next_tok of the previous transition = tokens after its step
            assert np.array_equal(tok["task_feats"], g["ntf"][pending]) and np.array_equal(tok["agent_feats"], g["naf"][pending]), f"{case} t={t}: next tokens"
            assert np.array_equal(tok["task_ids"], g["ntid"][pending]), f"{case} t={t}: next token ids"
            pending = None
        planned = bool(g["replanned"][t])
        sc = g["scores"][k] if planned else np.zeros((16, 32), np.float32)
        aa, ai, sel = e.allocate_scored(interval, 1, GATE_TRAINER, kind, 32, 16, SC_EDGE_VALID_ONLY, scores=sc)
        assert (e.scalars_last_plan() == t) == planned, f"{case} t={t}: gate"
        if planned:
            assert t == steps[k]
            assert np.array_equal(tok["task_feats"], g["tf"][k]) and np.array_equal(tok["agent_feats"], g["af"][k]), f"{case} t={t}: token features"
            assert np.array_equal(tok["edge_valid"], g["ev"][k]) and np.array_equal(tok["task_ids"], g["tid"][k]) and np.array_equal(tok["agent_ids"], g["aid"][k]), f"{case} t={t}: edge_valid / ids"
            _check_lsap(e, g, t, case)
            assert np.array_equal(sel, g["selected"][k]), f"{case} t={t}: selected mask"
            want = g["pairs"][g["pairs"][:, 0] == t][:, 1:]
            assert sorted(map(tuple, e.last_actions().tolist())) == sorted(map(tuple, want.tolist())), f"{case} t={t}: pairs"
            pending = k
        else:
            assert len(aa) == 0 and not sel.any()
        want_a = _acts_at(g, t)
        order = np.argsort(aa, kind="stable")
        assert np.array_equal(np.stack([aa[order], ai[order]], axis=1).reshape(-1, 2), want_a.reshape(-1, 2)), f"{case} t={t}: actions"
        done = e.step(aa, ai)
        assert e.metrics()[4] == g["s_wps"][t + 1], f"{case} t={t}: S_WPS"
        if planned:
            assert (e.metrics()[4] - g["s_wps"][t]) / 20.0 == g["step_r"][k] and bool(done) == bool(g["ep_done"][k]), f"{case} t={t}: step reward / done"
            k += 1
    assert k == len(steps) and np.array_equal(e.metrics(), g["metrics"]), f"{case}: final metrics"
    assert e.dims()["n_replans"] == int(g["n_replans"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rl_*.npz"))))
def test_rl_loop_scored_plans_vs_reference(path):
    case = os.path.basename(path)[3:-4]
    check_rl(np.load(path), case, params_for_case(case))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rah_*.npz"))))
def test_priorities_and_reserved_agents_vs_reference(path):
    """AttentionRAH.plan (AttentionRAH.py:395-453) with seeded (rho, pri_vec): task_priorities + reserved_agent_names as the
    planner passed them, over build_att_tokens' UNTRUNCATED open list, under wps_eval._should_replan(15)"""
    g = np.load(path)
    case = os.path.basename(path)[4:-4]
    e = orc.OracleEnv(params_for_case(case))
    e.reset(int(g["seed"]))
    steps, k = g["step"].tolist(), 0
    for t in range(len(g["replanned"])):
        planned = bool(g["replanned"][t])
        pri = g["pri"][k] if planned else np.zeros(32)
        res = int(g["reserved"][k]) if planned else 0
        if planned:
            assert np.array_equal(e.tokens(0, 32, 16)["task_ids"], g["tid"][k])
        aa, ai, _ = e.allocate_scored(15, 1, GATE_TRAINER, 0, 32, 16, SC_FULL_TASK_LIST, pri=pri, reserved=res)
        assert (e.scalars_last_plan() == t) == planned
        if planned:
            assert t == steps[k]
            _check_lsap(e, g, t, case)
            want = g["pairs"][g["pairs"][:, 0] == t][:, 1:]
            assert sorted(map(tuple, e.last_actions().tolist())) == sorted(map(tuple, want.tolist())), f"{case} t={t}: pairs"
            k += 1
        want_a = _acts_at(g, t)
        order = np.argsort(aa, kind="stable")
        assert np.array_equal(np.stack([aa[order], ai[order]], axis=1).reshape(-1, 2), want_a.reshape(-1, 2)), f"{case} t={t}: actions"
        e.step(aa, ai)
    assert k == len(steps) and np.array_equal(e.metrics(), g["metrics"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "esc_*.npz"))))
def test_escort_scored_plans_vs_reference(path):
    """AttentionEscort.plan (AttentionEscort.py:472-522) with seeded scores: edges over build_escort_tokens' sorted task list,
    reserved = committed_names, apply_agent_commits, under escort_eval._should_replan"""
    g = np.load(path)
    case = os.path.basename(path)[4:-4]
    mt, ma, interval = int(g["max_tasks"]), int(g["max_agents"]), int(g["interval"])
    e = orc.OracleEnv(params_for_case(case))
    e.reset(int(g["seed"]))
    steps, k = g["step"].tolist(), 0
    for t in range(len(g["replanned"])):
        planned = bool(g["replanned"][t])
        sc = g["scores"][k] if planned else np.zeros((ma, mt), np.float32)
        if planned:
            tok = e.tokens(2, mt, ma)
            assert np.array_equal(tok["task_ids"], g["tid"][k]) and np.array_equal(tok["agent_ids"], g["aid"][k])
        aa, ai, sel = e.allocate_scored(interval, 1, GATE_ESCORT, 2, mt, ma, SC_COMMIT, scores=sc)
        assert (e.scalars_last_plan() == t) == planned
        if planned:
            assert t == steps[k]
            _check_lsap(e, g, t, case)
            assert np.array_equal(sel, g["selected"][k]), f"{case} t={t}: selected mask"
            assert np.array_equal(e.agent_commit_until(), g["commit"][k]), f"{case} t={t}: commit_until"
            k += 1
        want_a = _acts_at(g, t)
        order = np.argsort(aa, kind="stable")
        assert np.array_equal(np.stack([aa[order], ai[order]], axis=1).reshape(-1, 2), want_a.reshape(-1, 2)), f"{case} t={t}: actions"
        e.step(aa, ai)
    assert k == len(steps) and np.array_equal(e.metrics(), g["metrics"])


# ---- run to the next gate (muavta_rl_run_device's checker): the reference episodes hold every gate step ------------------------------
def check_rl_run(g, case, params, max_steps):
    """OracleEnv.rl_run — plan at a gate, step, then env.step({}) up to the next gate (experiments/train_pair_cost.py:139-145) — replays a reference
    run_rl_episode trace LAUNCH BY LAUNCH: one launch per gate step of the reference (its `step` list), each consuming the score matrix the reference's
    policy was given there and reproducing the pushed transition (tok, selected, step_r, next_tok, ep_done); the quiet stretch ends exactly at the
    reference's next gate.  max_steps > 0 cuts the quiet stretches short: the extra launches must plan nothing."""
    kind = 1 if int(g["raw"]) else 0
    e = orc.OracleEnv(params)
    e.reset(int(g["seed"]))
    steps, T = list(np.asarray(g["step"]).tolist()), len(g["replanned"])
    tok = e.tokens(kind, 32, 16)
    t, k, launches = 0, 0, 0
    while t < T:
        at = k < len(steps) and steps[k] == t
        sc = g["scores"][k] if at else np.full((16, 32), 0.3, np.float32)  # (scores handed to an env that is not at a gate are not read)
        r = e.rl_run(20, 1, GATE_TRAINER, kind, 32, 16, SC_EDGE_VALID_ONLY, scores=sc, max_steps=max_steps)
        launches += 1
        assert r["replanned"] == at, f"{case} t={t}: gate"
        if at:
            assert np.array_equal(tok["task_feats"], g["tf"][k]) and np.array_equal(tok["agent_feats"], g["af"][k]) and np.array_equal(tok["edge_valid"], g["ev"][k])
            assert np.array_equal(r["selected"], g["selected"][k]), f"{case} t={t}: selected"
            assert (r["s_after"] - r["s_before"]) / 20.0 == g["step_r"][k] and bool(r["done"]) == bool(g["ep_done"][k])
            nt = r["next_tok"]
            assert np.array_equal(nt["task_feats"], g["ntf"][k]) and np.array_equal(nt["agent_feats"], g["naf"][k]) and np.array_equal(nt["task_ids"], g["ntid"][k])
            k += 1
        else:
            assert not r["selected"].any()
        assert r["s_before"] == g["s_wps"][t] and r["s_after"] == g["s_wps"][t + 1]
        nxt_gate = steps[k] if k < len(steps) else T
        want_n = nxt_gate - t if max_steps == 0 else min(nxt_gate - t, max_steps)
        assert r["n_stepped"] == want_n, f"{case} t={t}: {r['n_stepped']} steps, next gate of the reference at {nxt_gate}"
        t += r["n_stepped"]
        assert bool(r["park"] & 4) == (t < T and t == nxt_gate) and bool(r["park"] & 3) == (t == T)
        tok = r["park_tok"]
    assert k == len(steps) and np.array_equal(e.metrics(), g["metrics"]) and e.dims()["n_replans"] == int(g["n_replans"])
    if max_steps == 0:
        assert launches == len(steps)
    r = e.rl_run(20, 1, GATE_TRAINER, kind, 32, 16, SC_EDGE_VALID_ONLY, scores=None)  # an ended episode is left alone
    assert r["n_stepped"] == 0 and not r["replanned"] and r["s_before"] == r["s_after"] == g["s_wps"][-1] and np.array_equal(e.metrics(), g["metrics"])


@pytest.mark.parametrize("max_steps", [0, 3])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rl_*.npz"))))
def test_rl_run_to_the_next_gate_vs_reference(path, max_steps):
    """muavta_rl_run_device's checker pinned on the reference's own episodes (tests/golden/rl_*.npz hold every gate step): see check_rl_run."""
    case = os.path.basename(path)[3:-4]
    check_rl_run(np.load(path), case, params_for_case(case), max_steps)


# ---- RL episodes that END EARLY (every task done before max_time_steps): tests/fuzz_reference_rl.py --pin ---------------------------
RLTERM_FILES = sorted(glob.glob(os.path.join(GOLDEN, "rlterm_WIDE*.npz")))


def _rlterm_params(path):
    """The configuration of a pinned early-ending episode (tests/golden/rlterm_configs.json; not in wide_configs.json: that table feeds device
    tests of its own) on the largest tile, as tests/fuzz_reference_rl.py checks it."""
    import json

    from muavta_amd.params import params_from_config

    case = os.path.basename(path)[len("rlterm_"):-4]
    cfg = dict(json.load(open(os.path.join(GOLDEN, "rlterm_configs.json")))[case])
    cfg["threats_list"] = [tuple(x) for x in cfg["threats_list"]]
    cfg["escort_agent_types"] = tuple(cfg["escort_agent_types"])
    return case, cfg, params_from_config(cfg, None, tile_agents=64, tile_tasks=128, tile_threats=48)


@pytest.mark.parametrize("path", RLTERM_FILES)
def test_rl_loop_on_episodes_that_end_early_vs_reference(path):
    """run_rl_episode (experiments/train_pair_cost.py:132-156) on configurations whose episode TERMINATES before max_time_steps — the registry
    cases' rl_*.npz all run to their truncation step (round-4 advisor note on rl_stream).  The reference's loop stops with the episode; the
    oracle's scored allocator + step reproduce every plan up to it, the done flag of the last pushed transition and the final metrics."""
    g = np.load(path)
    case, cfg, P = _rlterm_params(path)
    assert len(g["replanned"]) < int(cfg["max_time_steps"]), "fixture no longer ends early"
    check_rl(g, case, P)


@pytest.mark.parametrize("max_steps", [0, 1, 4])
@pytest.mark.parametrize("path", RLTERM_FILES)
def test_rl_run_to_the_next_gate_on_episodes_that_end_early(path, max_steps):
    """muavta_rl_run_device's checker on the same episodes: the last launch's quiet stretch ends with the EPISODE (park bit 0), not at a gate,
    and a further launch leaves the ended env alone (check_rl_run's closing assertions)."""
    g = np.load(path)
    case, cfg, P = _rlterm_params(path)
    check_rl_run(g, case, P, max_steps)


def test_rl_run_batch_with_envs_that_end_at_different_steps():
    """What a batched caller of muavta_rl_run_device sees when its envs end at different steps (the loop of muavta_amd/il.py::rl_run_stream over
    the oracle): every env keeps being launched until ALL have ended, the ended ones must report n_stepped 0 / no plan / unchanged S_WPS and
    metrics at every further launch, and the others must be unaffected by them — each env's launches equal the reference episode's gates."""
    if len(RLTERM_FILES) < 2:
        pytest.skip("needs two pinned early-ending episodes")
    eps = []
    for path in RLTERM_FILES:
        g = np.load(path)
        case, cfg, P = _rlterm_params(path)
        e = orc.OracleEnv(P)
        e.reset(int(g["seed"]))
        eps.append(dict(g=g, case=case, e=e, kind=1 if int(g["raw"]) else 0, t=0, k=0, T=len(g["replanned"]), steps=np.asarray(g["step"]).tolist(), launches_after_end=0))
    assert len({ep["T"] for ep in eps}) > 1
    while any(ep["t"] < ep["T"] for ep in eps):
        for ep in eps:  # one "launch" of the whole batch
            g, e = ep["g"], ep["e"]
            over = ep["t"] >= ep["T"]
            at = (not over) and ep["k"] < len(ep["steps"]) and ep["steps"][ep["k"]] == ep["t"]
            sc = g["scores"][ep["k"]] if at else np.full((16, 32), -0.2, np.float32)
            r = e.rl_run(20, 1, GATE_TRAINER, ep["kind"], 32, 16, SC_EDGE_VALID_ONLY, scores=sc)
            if over:
                ep["launches_after_end"] += 1
                assert r["n_stepped"] == 0 and not r["replanned"] and not r["selected"].any() and r["s_before"] == r["s_after"] == g["s_wps"][-1], ep["case"]
                assert bool(r["park"] & 3) and np.array_equal(e.metrics(), g["metrics"]), ep["case"]
                continue
            assert r["replanned"] == at, f"{ep['case']} t={ep['t']}"
            if at:
                assert np.array_equal(r["selected"], g["selected"][ep["k"]]) and bool(r["done"]) == bool(g["ep_done"][ep["k"]])
                ep["k"] += 1
            ep["t"] += r["n_stepped"]
            assert ep["t"] == (ep["steps"][ep["k"]] if ep["k"] < len(ep["steps"]) else ep["T"]), f"{ep['case']}: stopped at {ep['t']}"
    for ep in eps:
        assert ep["k"] == len(ep["steps"]) and np.array_equal(ep["e"].metrics(), ep["g"]["metrics"]), ep["case"]
    assert any(ep["launches_after_end"] > 0 for ep in eps)


def test_avoid_obstacles_on_an_obstacle_axis_is_decided_by_the_last_bit_of_atan2():
    """The pair the device fuzz found (tests/fuzz_device.py configuration 32517, scored leg, step 182): an agent whose task lies inside an
    obstacle's keep-out zone is steered onto the line through the obstacle's centre, where the two headings of sim_core.rs:46-47 are one ulp
    apart and the side it passes on is the last bit of each atan2.  Pins what the oracle (the host libm) does there — the value the device's
    restated atan2 (csrc/muavta_atan2.h) must reproduce, and does under -m gpu."""
    import ctypes as C
    import math
    from conftest import host_libm_note

    if host_libm_note():
        pytest.skip(host_libm_note())
    f = float.fromhex
    pos = np.array([f("0x1.e1186bb3bf6abp+8"), f("0x1.c51c9fc28c1a9p+7")])
    mov = np.array([f("-0x1.5c070de57194ep-6"), f("-0x1.ffe26cf27704ap-1")])
    obst = np.array([[f("0x1.027ff88080afap+8"), f("0x1.41d8385dcd497p+8"), f("0x1.3p+5")], [f("0x1.df7adf58ae2f0p+8"), f("0x1.2d0c686c71993p+7"), f("0x1.04p+6")],
                     [f("0x1.c2bf3701aaba9p+9"), f("0x1.66a2e1dda60d5p+7"), f("0x1.08p+5")]])
    a_mov, a_obs = math.atan2(mov[1], mov[0]), math.atan2(obst[1, 1] - pos[1], obst[1, 0] - pos[0])
    assert (a_mov.hex(), a_obs.hex()) == ("-0x1.978fec4a46805p+0", "-0x1.978fec4a46806p+0")  # one ulp apart: ang + PI rounds to PI, the wrapped angle is 0.0, not > 0
    out = np.zeros(2)
    orc.lib().orc_avoid_obstacles(obst.ctypes.data_as(C.c_void_p), 3, pos.ctypes.data_as(C.c_void_p), mov.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert [v.hex() for v in out.tolist()] == ["-0x1.3a0ecc607185ep+1", "0x1.ab0d70316c8b9p-5"]


def test_avoid_obstacles_oracle_vs_arbitrary_precision_witness():
    """a9 with K > 0 obstacles stays "parity unpinned vs the Rust" (core_sim/src/sim_core.rs:25-59 cannot be built here) — what CAN be pinned:
    a third, independent evaluation of those lines with every IEEE operation in Python floats and `ln` / `atan2` in 300-bit arithmetic rounded
    once (tests/sim_core_witness.py).  The oracle (host libm) must equal it bit for bit wherever the libm's log is correctly rounded for the
    argument met; arguments where it is not are the only places a Rust build could differ, by an ulp of the force (tools/a9_witness.py lists them)."""
    import ctypes as C
    import math
    from sim_core_witness import avoid_cr

    rng = np.random.default_rng(11)
    obst = np.array([[300.0, 300.0, 50.0], [700.0, 200.0, 80.0], [500.0, 500.0, 30.0]])
    n = 1500
    pos = rng.uniform(100, 900, (n, 2)); mov = rng.uniform(-1, 1, (n, 2))
    k = n * 2 // 3   # two thirds of the positions inside a 40-unit zone: the branch is taken
    which = rng.integers(0, 3, k); ang = rng.uniform(0, 2 * math.pi, k); rad = obst[which, 2] + rng.uniform(0.2, 39.9, k)
    pos[:k, 0] = obst[which, 0] + rad * np.cos(ang); pos[:k, 1] = obst[which, 1] + rad * np.sin(ang)
    mov[::7] = 0.0  # atan2(0, 0)
    L = orc.lib()
    taken = explained = 0
    for i in range(n):
        out = np.zeros(2)
        L.orc_avoid_obstacles(obst.ctypes.data_as(C.c_void_p), 3, pos[i].ctypes.data_as(C.c_void_p), mov[i].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        want, info = avoid_cr(pos[i].tolist(), obst.tolist(), mov[i].tolist())
        taken += len(info)
        if all(r["libm_log_cr"] for r in info) and all(abs(r["angle_between"]) > 1e-9 and abs(abs(r["angle_between"]) - math.pi) > 1e-9 for r in info):
            assert out[0] == want[0] and out[1] == want[1], f"pair {i}: oracle {out.tolist()} vs witness {want}"
        else:
            explained += 1
    assert taken > n // 2 and explained < n // 100, (taken, explained)


CONTEXT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "context_*.npz")))


@pytest.mark.parametrize("path", CONTEXT_FILES, ids=[os.path.basename(p)[8:-4] for p in CONTEXT_FILES])
def test_context_vector_vs_reference(path):
    """build_context_summary of the ContextPair hybrids (TaskAllocation/Hybrid/ContextPairHybrid.py:33-78) sampled along reference episodes
    (tools/gen_golden.py --context): token pads 32 x 16 and 12 x 6, and the raw variant — bit for bit."""
    g = np.load(path)
    case = os.path.basename(path)[8:-4]
    o = orc.OracleEnv(params_for_case(case))
    o.reset(int(g["seed"]))
    steps = g["step"].tolist()
    for t in range(150):
        oa, oi = o.allocate(int(g["interval"]), 1)
        if t in steps:
            k = steps.index(t)
            assert np.array_equal(o.context(0, 32), g["ctx"][k]) and np.array_equal(o.context(0, 12), g["ctx12"][k]) and np.array_equal(o.context(1, 32), g["ctx_raw"][k]), f"{case} t={t}"
        o.step(oa, oi)
