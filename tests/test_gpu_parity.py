"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's golden
vectors.  Bit-exact on every integer / index / mask AND on every f64 (positions, rewards, times) —
stricter than the north-star's 1e-5 — plus size-independent properties at BASELINE.json's full sizes."""
import glob
import os

import numpy as np
import pytest

import orc
from muavta_amd.params import METRIC_KEYS, params_for_case

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _env(case, n, **kw):
    from muavta_amd.batched import BatchedMultiUAVEnv
    return BatchedMultiUAVEnv(params_for_case(case, **kw), n)


class Snapshot:
    """All device fields of all envs, fetched once per step."""
    NAMES = ["AGENT_POS", "AGENT_STATE", "AGENT_HEAD", "AGENT_QUEUE", "AGENT_NFT", "AGENT_NFP", "AGENT_CAPS",
             "AGENT_ATTACK_CAP", "AGENT_TYPE", "AGENT_NAME_IDX", "AGENT_DIST", "AGENT_MISC", "TASK_ID", "TASK_STATUS",
             "TASK_POS", "TASK_CUR", "TASK_ALLOC", "TASK_ORG_DONE", "TASK_META", "TASK_TIMES", "KNOWN", "THREAT_POS",
             "THREAT_META", "SCALARS", "OPEN_IDS", "EVENTS", "ERROR"]

    def __init__(self, env):
        for n in self.NAMES:
            setattr(self, n, env.get(n))
        self.obs = env.observe()
        self.reward, self.term, self.trunc = env.step_result()


def compare(snap, i, o, tag, check_obs=True):
    """env i of the device snapshot vs oracle env o (check_obs=False: between steps the reference holds no fresh observation)."""
    assert snap.ERROR[i] == 0, f"{tag}: device tile overflow code {snap.ERROR[i]}"
    rows, caps, q = o.agents()
    A = rows.shape[0]
    assert np.array_equal(snap.AGENT_POS[i], rows[:, 0:2]), f"{tag}: agent positions"
    assert np.array_equal(snap.AGENT_STATE[i], rows[:, 2].astype(int)), f"{tag}: agent state"
    assert np.array_equal(snap.AGENT_HEAD[i], rows[:, 3].astype(int)), f"{tag}: head task id"
    Q = snap.AGENT_QUEUE.shape[2]
    assert np.array_equal(snap.AGENT_QUEUE[i], q[:, :Q]), f"{tag}: queues\n{snap.AGENT_QUEUE[i]}\n{q[:, :Q]}"
    assert np.array_equal(snap.AGENT_NFT[i], rows[:, 5]), f"{tag}: next_free_time"
    assert np.array_equal(snap.AGENT_NFP[i], rows[:, 6:8]), f"{tag}: next_free_position"
    assert np.array_equal(snap.AGENT_CAPS[i], caps), f"{tag}: caps"
    assert np.array_equal(snap.AGENT_ATTACK_CAP[i], rows[:, 8].astype(int)), f"{tag}: attackCap"
    misc = snap.AGENT_MISC[i]
    assert np.array_equal(misc[:, 0], rows[:, 9].astype(int)), f"{tag}: task_start"
    assert np.array_equal(misc[:, 1], rows[:, 14].astype(int)), f"{tag}: fail_event"
    assert np.array_equal(misc[:, 2], rows[:, 10].astype(int)), f"{tag}: re_eval"
    assert np.array_equal(misc[:, 3], rows[:, 11].astype(int)), f"{tag}: last_task"
    assert np.array_equal(snap.AGENT_TYPE[i], rows[:, 12].astype(int)), f"{tag}: agent type"
    assert np.array_equal(snap.AGENT_NAME_IDX[i], rows[:, 13].astype(int)), f"{tag}: agent names"
    assert np.array_equal(snap.AGENT_DIST[i], rows[:, 15]), f"{tag}: agent_distances"
    trow, reqs = o.tasks()
    ids = snap.TASK_ID[i]
    live = np.nonzero(ids >= 0)[0]
    assert len(set(ids[live])) == len(live), f"{tag}: duplicate task ids in slots"
    open_oracle = {k for k in range(1, trow.shape[0]) if int(trow[k, 0]) != 2}
    assert open_oracle <= set(ids[live].tolist()), f"{tag}: open task missing on device"
    known = o.known()
    org = o.task_org()
    for s in live:
        k = int(ids[s])
        assert snap.TASK_STATUS[i, s] == int(trow[k, 0]), f"{tag}: task {k} status"
        assert np.array_equal(snap.TASK_POS[i, s], trow[k, 1:3]), f"{tag}: task {k} position"
        assert np.array_equal(snap.TASK_CUR[i, s], reqs[k, 0]), f"{tag}: task {k} currentReqs"
        assert np.array_equal(snap.TASK_ALLOC[i, s], reqs[k, 1]), f"{tag}: task {k} allocatedReqs"
        ty = int(trow[k, 6])
        assert snap.TASK_ORG_DONE[i, s, 1] == reqs[k, 2, ty], f"{tag}: task {k} doneReqs"
        assert snap.TASK_ORG_DONE[i, s, 0] == org[k], f"{tag}: task {k} orgReqs"
        meta = snap.TASK_META[i, s]
        assert list(meta[0:5]) == [int(x) for x in trow[k, 6:11]], f"{tag}: task {k} meta {meta} vs {trow[k, 6:11]}"
        assert meta[6] == int(trow[k, 11]) and meta[7] == int(trow[k, 12]), f"{tag}: task {k} escort meta"
        if int(trow[k, 0]) != 2:
            assert np.array_equal(snap.TASK_TIMES[i, s], trow[k, 3:5]), f"{tag}: task {k} init/done time"
            assert meta[5] == int(trow[k, 5]), f"{tag}: task {k} len(allocationDetails)"
            bits = (snap.KNOWN[i][:, s >> 5] >> (s & 31)) & 1
            assert np.array_equal(bits.astype(bool), known[:, k]), f"{tag}: who knows task {k}"
    th = o.threats()
    tm = snap.THREAT_META[i]
    assert np.array_equal(tm[:, 0], th[:, 0].astype(int)), f"{tag}: threat status"
    act = th[:, 0] != -9
    assert np.array_equal(snap.THREAT_POS[i][act], th[act, 1:3]), f"{tag}: threat positions"
    assert np.array_equal(tm[act, 1], th[act, 3].astype(int)), f"{tag}: threat targets"
    assert np.array_equal(tm[act, 2], th[act, 4].astype(int)), f"{tag}: threat mission targets"
    assert np.array_equal(tm[act, 3], th[act, 5].astype(int)), f"{tag}: threat attackCap"
    assert np.array_equal(tm[act, 4], th[act, 6].astype(int)), f"{tag}: threat task ids"
    sc = o.scalars()
    d = o.dims()
    assert np.array_equal(snap.SCALARS[i][:24], sc), f"{tag}: scalars\n{snap.SCALARS[i][:24]}\n{sc}"
    assert list(snap.SCALARS[i][24:28]) == [d["pending_reset"], d["n_reached"], d["n_pending"], d["n_task_ids"] - 1], f"{tag}: counters"
    oi = o.open_ids()
    assert np.array_equal(snap.OPEN_IDS[i][: len(oi)], oi) and np.all(snap.OPEN_IDS[i][len(oi):] == -1), f"{tag}: open list"
    ev = o.events()
    dev = snap.EVENTS[i]
    n_ev = int((dev[:, 0] >= 0).sum())
    assert n_ev == len(ev) and np.array_equal(dev[:n_ev], ev), f"{tag}: drained events"
    if not check_obs:
        return
    ti, legal, pad, ag, fl = o.observe()
    assert np.array_equal(snap.obs["tasks"][i], ti), f"{tag}: obs tasks_info"
    assert np.array_equal(snap.obs["legal_mask"][i], legal), f"{tag}: legal_mask"
    assert np.array_equal(snap.obs["mask"][i], pad), f"{tag}: pad mask"
    assert np.array_equal(snap.obs["agents"][i], ag), f"{tag}: obs agent rows"
    assert np.array_equal(snap.obs["event_flags"][i], fl), f"{tag}: event flags"
    assert snap.reward[i] == sc[1], f"{tag}: reward"
    assert bool(snap.term[i]) == bool(d["terminated"]) and bool(snap.trunc[i]) == bool(d["truncated"]), f"{tag}: done flags"


CASES = [("WPS_easy", 20, 6), ("WPS_hard", 20, 8), ("WPS_burst", 20, 4), ("WPS_attn", 20, 4), ("WPS_attn_AWACS", 20, 3),
         ("D2_popup_threats", 20, 2), ("WPS_hard_x2", 20, 8), ("WPS_escort", 12, 6), ("WPS_escort24", 12, 4), ("WPS_burst64", 20, 2),
         ("WPS_commit", 20, 3), ("WPS_attn_OS24", 20, 2), ("WPS_attn_L", 20, 2), ("WPS_attn_XL", 20, 2)]


@pytest.mark.parametrize("case,interval,n", CASES, ids=[c[0] for c in CASES])
def test_stepwise_bit_exact_vs_oracle(case, interval, n):
    """reset + 150 x (allocate -> step) through the C ABI, every field of every env compared each step."""
    env = _env(case, n)
    seeds = np.arange(n, dtype=np.uint64)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    snap = Snapshot(env)
    for i, o in enumerate(oracles):
        compare(snap, i, o, f"{case} seed {i} after reset")
    for t in range(150):
        aa, ai = env.allocate(interval, True)
        staged = env.get("STAGED_ACTIONS")
        for i, o in enumerate(oracles):
            oa, oi = o.allocate(interval, 1)
            k = len(oa)
            assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1), f"{case} seed {i} t={t}: assigned agents {aa[i]} vs {oa}"
            assert np.array_equal(ai[i][:k], oi), f"{case} seed {i} t={t}: assigned indices"
            assert np.array_equal(staged[i][:k, 1], o.last_actions()[:, 1]), f"{case} seed {i} t={t}: assigned task ids"
            o.step(oa, oi)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"{case} seed {i} t={t + 1}")
    m = env.metrics()
    for i, o in enumerate(oracles):
        assert np.array_equal(m[i], o.metrics()), f"{case} seed {i}: final metrics"


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "metrics_*.npz"))),
                         ids=lambda p: os.path.basename(p)[8:-4])
def test_fused_rollout_matches_reference_metrics(path):
    """ONE kernel launch (reset + 150 x allocate/step) reproduces the metrics the reference itself produced."""
    g = np.load(path)
    case = os.path.basename(path)[8:-4]
    want = g["metrics"]
    n = want.shape[0]
    env = _env(case, n)
    env.rollout(np.arange(n, dtype=np.uint64), 150, int(g["interval"]), True, True)
    got = env.rollout_metrics()
    assert np.all(env.get("ERROR") == 0)
    bad = np.nonzero(~np.all(got == want, axis=1))[0]
    assert len(bad) == 0, f"{case}: seeds {bad[:8]} differ, e.g. {dict(zip(METRIC_KEYS, got[bad[0]] - want[bad[0]]))}"
    assert np.array_equal(env.get("SCALARS")[:, 23].astype(int), g["n_replans"])
    assert np.array_equal(env.metrics(), want)


@pytest.mark.parametrize("case,interval,n", [("WPS_hard_x2", 20, 4096), ("WPS_escort24", 12, 4096), ("WPS_burst64", 20, 1024), ("WPS_escort", 12, 1024),
                                              ("WPS_hard", 20, 2048), ("WPS_attn", 20, 512)],
                         ids=["cfg2b", "cfg4b", "cfg5", "escort", "hard", "attn"])
def test_fused_rollout_vs_oracle_many_seeds(case, interval, n):
    """BASELINE configs 2, 4 and 5 at their full per-GPU sizes (4096 / 4096 / 1024 envs): every env's 30 metrics."""
    env = _env(case, n)
    seeds = np.arange(1000, 1000 + n, dtype=np.uint64)
    env.rollout(seeds, 150, interval, True, False)
    got = env.rollout_metrics()
    assert np.all(env.get("ERROR") == 0)
    want = orc.parallel_metrics(case, seeds, interval)
    bad = np.nonzero(~np.all(got == want, axis=1))[0]
    assert len(bad) == 0, f"{case}: seeds {seeds[bad[:8]]} differ, e.g. {dict(zip(METRIC_KEYS, got[bad[0]] - want[bad[0]]))}"


def test_config3_seed_range_0_32767_no_overflow_bit_equal():
    """BASELINE config 3: global env indices 0..32767 of WPS_hard_x2 in 4096-env shards (what each of the 8 GPUs runs).
    Every env must produce a result (ERROR == 0: no tile overflow) and all 30 metrics must equal the oracle's."""
    case, shard = "WPS_hard_x2", 4096
    env = _env(case, shard)
    want = orc.parallel_metrics(case, np.arange(32768), 20)
    for r in range(8):
        seeds = np.arange(r * shard, (r + 1) * shard, dtype=np.uint64)
        env.rollout(seeds, 150, 20, True, True)
        got = env.rollout_metrics()
        err = env.get("ERROR")
        assert not err.any(), f"shard {r}: envs {seeds[np.nonzero(err)[0][:8]]} overflowed the tile (codes {np.unique(err[err != 0])})"
        bad = np.nonzero(~np.all(got == want[r * shard:(r + 1) * shard], axis=1))[0]
        assert len(bad) == 0, f"shard {r}: seeds {seeds[bad[:8]]} differ"


SOAK = [("WPS_easy", "hungarian", 0, 20, 1024), ("WPS_burst", "hungarian", 0, 20, 1024), ("WPS_attn_AWACS", "hungarian", 0, 20, 512),
        ("D2_popup_threats", "hungarian", 0, 20, 512), ("WPS_hard", "urgency_pair", 1, 20, 1024), ("WPS_hard_x2", "urgency_pair", 1, 20, 2048),
        ("WPS_escort", "urgency_coalition", 2, 12, 1024), ("WPS_escort24", "urgency_coalition", 2, 12, 512), ("WPS_hard", "hungarian_gated", 3, 20, 1024)]


@pytest.mark.parametrize("case,name,mode,interval,n", SOAK, ids=[f"{c[0]}-{c[1]}" for c in SOAK])
def test_soak_slice(case, name, mode, interval, n):
    """A bounded slice of tests/soak.py: other cases / allocator modes on a seed range no other test uses."""
    env = _env(case, n)
    env.set_allocator(name)
    seeds = np.arange(200000, 200000 + n, dtype=np.uint64)
    env.rollout(seeds, 150, interval, True, False)
    got, err = env.rollout_metrics(), env.get("ERROR")
    assert not err.any(), f"{case} {name}: envs {seeds[np.nonzero(err)[0][:8]]} overflowed the tile"
    want = orc.parallel_metrics(case, seeds, interval, 1, mode)
    bad = np.nonzero(~np.all(got == want, axis=1))[0]
    assert len(bad) == 0, f"{case} {name}: seeds {seeds[bad[:8]]} differ"


@pytest.mark.parametrize("case,interval", [("WPS_hard_x2", 20), ("WPS_escort24", 12), ("WPS_burst64", 20)])
def test_observation_after_fused_rollout(case, interval):
    """The observation tensors a fused rollout leaves behind (written by its last step) equal the oracle's, per-step
    observation write on and off; so do the step result and the whole device state."""
    n = 6
    seeds = np.arange(40, 40 + n, dtype=np.uint64)
    for write_obs, steps in ((True, 150), (False, 150), (True, 67)):
        env = _env(case, n)
        env.rollout(seeds, steps, interval, True, write_obs)
        snap = Snapshot(env)
        for i in range(n):
            o = orc.OracleEnv(params_for_case(case))
            o.rollout(int(seeds[i]), steps, interval, 1)
            compare(snap, i, o, f"{case} seed {seeds[i]} after a fused rollout of {steps} steps (obs write {write_obs})")


def test_global_hungarian_and_split_rollout():
    """use_visibility=0 (Global-Hungarian) and a rollout split into 3 launches equal the oracle's single run."""
    case, n = "WPS_hard", 32
    env = _env(case, n)
    seeds = np.arange(n, dtype=np.uint64)
    env.rollout(seeds, 50, 20, False, False)
    env.rollout(None, 60, 20, False, True)
    env.rollout(None, 40, 20, False, False)
    got = env.metrics()
    o = orc.OracleEnv(params_for_case(case))
    for i in range(n):
        o.rollout(i, 150, 20, 0)
        assert np.array_equal(got[i], o.metrics()), f"seed {i}"


def test_full_size_properties_cfg2():
    """BASELINE config 2 (4096 envs, 16-agent tile): determinism, checkpoint/resume, and counter invariants."""
    case, n = "WPS_hard_x2", 4096
    env = _env(case, n)
    seeds = np.arange(n, dtype=np.uint64)
    env.rollout(seeds, 150, 20, True, True)
    m1 = env.rollout_metrics()
    assert np.all(env.get("ERROR") == 0)
    sc = env.get("SCALARS")
    assert np.all(sc[:, 0] == 150)
    K = {k: j for j, k in enumerate(METRIC_KEYS)}
    assert np.all(m1[:, K["n_on_time"]] + m1[:, K["n_missed_windows"]] <= m1[:, K["n_windowed_tasks"]])
    assert np.all(m1[:, K["n_tasks_final"]] <= env.max_tasks + 11)            # arrivals stop at max_tasks-1, threats always spawn
    assert np.all(m1[:, K["Losses"]] <= env.n_agents) and np.all(m1[:, K["total_distance"]] > 0)
    s_wps = 12.0 * m1[:, K["n_on_time"]] - 30.0 * m1[:, K["n_missed_windows"]] - 0.01 * m1[:, K["total_distance"]] / 1200.0
    assert np.array_equal(s_wps, m1[:, K["S_WPS"]])
    # spot-check 64 of the 4096 against the oracle
    o = orc.OracleEnv(params_for_case(case))
    for s in range(0, n, 64):
        o.rollout(s, 150, 20, 1)
        assert np.array_equal(m1[s], o.metrics()), f"seed {s}"
    # same seeds again -> identical bits; checkpoint at t=70 then resume -> identical bits
    env.rollout(seeds, 70, 20, True, False)
    state, rng = env.get_state(), env.get_rng()
    env.rollout(None, 80, 20, True, False)
    m2 = env.rollout_metrics()
    assert np.array_equal(m1, m2)
    env.rollout(seeds[::-1].copy(), 10, 20, True, False)  # scramble
    env.set_state(state); env.set_rng(rng)
    env.rollout(None, 80, 20, True, False)
    assert np.array_equal(m1, env.rollout_metrics())


def test_domain_sqrt_div_bit_exact():
    """fsqrt / fdiv (csrc/muavta_math.h): the compiler's sqrt and division sequences without range scaling and fix-ups must be
    the IEEE-754 correctly rounded results (numpy's) on everything the simulation can feed them: coordinates and their
    differences, squared distances, distances from 1e-12 up, speeds, zero numerators and radicands."""
    from muavta_amd.batched import domain_math

    rng = np.random.default_rng(20261004)
    n = 1 << 20
    pos = rng.uniform(0.0, 1200.0, size=(4, n))
    dx, dy = pos[0] - pos[1], pos[2] - pos[3]
    sq = dy * dy + dx * dx
    xs = [sq, dx, dy, np.sqrt(sq), rng.uniform(0, 2000, n), np.ldexp(rng.uniform(1, 2, n), rng.integers(-200, 40, n)),  # radicands / numerators
          np.zeros(64), np.array([1e-24, 1e-12, 1.0, 4.0, 1200.0 ** 2 * 2, np.inf]), np.ldexp(1.0, np.arange(-240, 240, dtype=np.int64))]
    ys = [np.sqrt(sq) + 1e-12, np.sqrt(sq) + 1e-12, rng.uniform(1e-12, 1e-6, n), rng.choice([0.1, 0.16, 0.4, 0.3, 0.28, 0.24, 10.0, 16.0], n),  # divisors
          rng.uniform(1e-3, 2000, n), np.ldexp(rng.uniform(1, 2, n), rng.integers(-40, 40, n)),
          rng.uniform(0.1, 50, 64), np.array([1e-12, 3.0, 7.0, 0.1, 1e-6, 1.0]), np.ldexp(1.0, -np.arange(-240, 240, dtype=np.int64) // 4) * 3.0]
    for x, y in zip(xs, ys):
        s, _, _ = domain_math(np.abs(x), y)
        assert np.array_equal(s.view(np.uint64), np.sqrt(np.abs(x)).view(np.uint64))
        fin = np.isfinite(x)
        _, q, qn = domain_math(x, y)
        assert np.array_equal(q[fin].view(np.uint64), (x[fin] / y[fin]).view(np.uint64))
        nz = fin & (x != 0)  # (a zero numerator gives +0 whatever its sign: callers never hold -0)
        assert np.array_equal(qn[nz].view(np.uint64), (-x[nz] / y[nz]).view(np.uint64))
        assert np.all(qn[fin & (x == 0)] == 0)


def test_lsap_known_answers_and_ties():
    from muavta_amd.batched import lsap
    g = np.load(os.path.join(GOLDEN, "lsap_cases.npz"))
    off = ro = 0
    for nr, nc in g["shape"]:
        c = g["cost"][off:off + nr * nc].reshape(nr, nc)
        m = min(nr, nc)
        for impl in ("lds", "registers"):  # both solvers against scipy's answers (tie-heavy, 1e6-masked, wide and tall)
            r, cc = lsap(c, impl=impl)
            assert np.array_equal(r, g["row"][ro:ro + m]) and np.array_equal(cc, g["col"][ro:ro + m]), (nr, nc, impl)
        off += nr * nc; ro += m
    rng = np.random.default_rng(5)
    batch = rng.integers(0, 3, (64, 64, 128)).astype(np.float64)  # maximum tile, tie-heavy
    r, c = lsap(batch)
    for k in range(0, 64, 8):
        orow, ocol = orc.lsap(batch[k])
        assert np.array_equal(r[k], orow) and np.array_equal(c[k], ocol)
    for shape in ((96, 32, 64), (96, 64, 32), (64, 24, 24), (64, 1, 64), (64, 32, 1)):  # register solver at its limits, ties everywhere
        batch = rng.integers(0, 3, shape).astype(np.float64)
        batch[rng.random(shape) < 0.3] = 1e6
        r, c = lsap(batch, impl="registers")
        r2, c2 = lsap(batch, impl="lds")
        assert np.array_equal(r, r2) and np.array_equal(c, c2)
        for k in range(0, shape[0], 6):
            orow, ocol = orc.lsap(batch[k])
            assert np.array_equal(r[k], orow) and np.array_equal(c[k], ocol), shape


def test_libm_log_bit_exact():
    """(r4) The obstacle repulsion's logarithm (sim_core.rs:44, f64::ln = the host libm's log): the device's restatement of that
    function's published algorithm against the HOST's log, bit for bit — the argument range of the path (1.05 .. 40), the
    near-1 branch, and positive normal numbers at large."""
    import math
    from muavta_amd.batched import domain_log
    rng = np.random.default_rng(17)
    x = np.concatenate([rng.uniform(1.05, 40.0, 1_200_000), rng.uniform(0.93, 1.07, 400_000), np.array([1.0, 1.05, 40.0, 0.9375, 1.0646, 1.0648]),
                        np.ldexp(rng.uniform(1.0, 2.0, 400_000), rng.integers(-1000, 1000, 400_000))])
    got = domain_log(x)
    want = np.fromiter(map(math.log, x.tolist()), dtype=np.float64, count=len(x))  # (math.log is libm's log; numpy may use its own SIMD kernels)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"{len(bad)} of {len(x)} differ, e.g. x={x[bad[0]].hex()} device {got[bad[0]].hex()} host {want[bad[0]].hex()}"


@pytest.mark.gpu
def test_libm_atan2_bit_exact():
    """(r5) The obstacle rule's heading test (sim_core.rs:46-47, f64::atan2 = the host libm's atan2, which glibc 2.35 does NOT round
    correctly): the device's restatement of that library's algorithm (csrc/muavta_atan2.h) against the HOST's atan2, bit for bit — unit
    vectors against coordinate differences (the rule's operands), every quadrant and both u = min / max forms, |x| == |y|, the
    polynomial / table boundary at u = 1/16, extreme exponents and the special operands."""
    import math
    from conftest import host_libm_note
    from muavta_amd.batched import domain_atan2
    if host_libm_note():
        pytest.skip(host_libm_note())
    rng = np.random.default_rng(23)
    n = 400_000
    a = rng.uniform(-math.pi, math.pi, n)
    d = rng.uniform(-5.0, 5.0, n)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e308, -1e308, 2.0 ** -500, 2.0 ** 500, 2.2250738585072014e-308])
    sy, sx = np.meshgrid(sp, sp)
    u16 = rng.uniform(0.0615, 0.0635, n)
    y = np.concatenate([np.sin(a), rng.uniform(-800, 800, n), rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-8, 8, n), d * rng.choice([1.0, -1.0], n), u16, -u16,
                        rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-1070, 1020, n), sy.ravel()])
    x = np.concatenate([np.cos(a), rng.uniform(-800, 800, n), rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-8, 8, n), d, np.ones(n), -np.ones(n),
                        rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-1070, 1020, n), sx.ravel()])
    got = domain_atan2(y, x)
    want = np.fromiter(map(math.atan2, y.tolist(), x.tolist()), dtype=np.float64, count=len(x))  # (math.atan2 is libm's)
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    bad = bad[~(np.isnan(got[bad]) & np.isnan(want[bad]))]
    assert len(bad) == 0, f"{len(bad)} of {len(x)} differ, e.g. y={y[bad[0]].hex()} x={x[bad[0]].hex()} device {got[bad[0]].hex()} host {want[bad[0]].hex()}"
    # the pair the device fuzz found (config 32517): an agent on the axis of an obstacle, the two headings one ulp apart
    f = float.fromhex
    px, py = f("0x1.e1186bb3bf6abp+8"), f("0x1.c51c9fc28c1a9p+7")
    pair = domain_atan2([f("-0x1.ffe26cf27704ap-1"), f("0x1.2d0c686c71993p+7") - py], [f("-0x1.5c070de57194ep-6"), f("0x1.df7adf58ae2f0p+8") - px])
    assert [v.hex() for v in pair.tolist()] == ["-0x1.978fec4a46805p+0", "-0x1.978fec4a46806p+0"]


def test_avoid_obstacles_vs_oracle():
    """K > 0 obstacles: no reference test pins this (parity unpinned: sim_core.rs cannot be compiled here); device vs the CPU
    restatement of core_sim/src/sim_core.rs:25-59.  (r4) Bit for bit: the logarithm is the host libm's on both sides
    (test_libm_log_bit_exact), sqrt / division are IEEE, and of atan2 only the sign of the wrapped angle difference is used."""
    import ctypes as C
    from muavta_amd.batched import avoid_obstacles
    rng = np.random.default_rng(3)
    obst = np.array([[300.0, 300.0, 50.0], [700.0, 200.0, 80.0], [500.0, 500.0, 30.0]])
    pos = rng.uniform(100, 900, (20000, 2)); mov = rng.uniform(-1, 1, (20000, 2))
    got = avoid_obstacles(pos, obst, mov)
    L = orc.lib()
    want = np.zeros_like(got)
    for i in range(len(pos)):
        out = np.zeros(2)
        L.orc_avoid_obstacles(obst.ctypes.data_as(C.c_void_p), 3, pos[i].ctypes.data_as(C.c_void_p),
                              mov[i].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        want[i] = out
    assert np.array_equal(got, want), f"{int((got != want).any(axis=1).sum())} of {len(pos)} differ"
    assert np.array_equal(avoid_obstacles(pos, np.zeros((0, 3)), mov), np.zeros_like(pos))  # K=0: the live configs
    # (r5) the arbitrary-precision witness of tests/sim_core_witness.py (every IEEE operation in Python floats, ln / atan2 in 300 bits,
    # rounded once): the device must equal it wherever the host libm's log is correctly rounded for the arguments met
    import math
    from sim_core_witness import avoid_cr
    near = pos[:1500].copy()
    which = rng.integers(0, 3, len(near)); ang = rng.uniform(0, 2 * math.pi, len(near)); rad = obst[which, 2] + rng.uniform(0.2, 39.9, len(near))
    near[:, 0] = obst[which, 0] + rad * np.cos(ang); near[:, 1] = obst[which, 1] + rad * np.sin(ang)
    got_near = avoid_obstacles(near, obst, mov[:1500])
    n_cr = 0
    for i in range(len(near)):
        w, info = avoid_cr(near[i].tolist(), obst.tolist(), mov[i].tolist())
        if all(r["libm_log_cr"] for r in info) and all(abs(r["angle_between"]) > 1e-9 and abs(abs(r["angle_between"]) - math.pi) > 1e-9 for r in info):
            assert got_near[i, 0] == w[0] and got_near[i, 1] == w[1], f"pair {i}: device {got_near[i].tolist()} vs arbitrary-precision witness {w}"
            n_cr += 1
    assert n_cr > 1400
    from muavta_amd.core_sim import SimCore  # the reference's call shape: lists in, [dx, dy] out (DroneEnv.py:1033)
    sc = SimCore()
    one = sc.avoid_obstacles(list(pos[5]), [list(o) for o in obst], list(mov[5]))
    assert isinstance(one, list) and len(one) == 2 and np.array_equal(one, want[5])
    assert sc.avoid_obstacles([1.0, 2.0], [], [0.5, 0.5]) == [0.0, 0.0]


def test_obstacles_random_init_and_single_task_mode():
    """Config knobs outside the WPS presets: num_obstacles>0 + random_init_pos, and multiple_tasks_per_agent=False."""
    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.params import params_from_config
    from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS
    for extra in ({"num_obstacles": 3, "random_init_pos": True}, {"multiple_tasks_per_agent": False},
                  {"capability_mask": True, "saturate_mask": True, "early_terminate": True,
                   "reward_weights": {"action": 0.5, "distance": 1.0, "quality": 1.0, "s_quality": 1.0, "time": 0.1,
                                      "alloc": 0.1, "time_penaulty": 0.25, "step": 0.3}}):
        flags = dict(WPS_ENV_FLAGS); flags.update(extra)
        spec = dict(CASE_SPECS["WPS_hard"]); spec.update({k: v for k, v in extra.items() if k in ("num_obstacles", "random_init_pos")})
        p = params_from_config(spec, flags)
        n = 6
        env = BatchedMultiUAVEnv(p, n)
        seeds = np.arange(n, dtype=np.uint64)
        env.reset(seeds)
        oracles = [orc.OracleEnv(p) for _ in range(n)]
        for i, o in enumerate(oracles):
            o.reset(i)
        for t in range(150):
            aa, ai = env.allocate(20, True)
            for i, o in enumerate(oracles):
                oa, oi = o.allocate(20, 1)
                assert np.array_equal(aa[i][: len(oa)], oa) and np.array_equal(ai[i][: len(oa)], oi), f"{extra} seed {i} t={t}"
                o.step(oa, oi)
            env.step(aa, ai)
            snap = Snapshot(env)
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"{extra} seed {i} t={t + 1}")
            if all(o.dims()["terminated"] or o.dims()["truncated"] for o in oracles):
                break


def test_invalid_and_out_of_range_actions():
    """Out-of-range index => action_reward -= 1, not an error (DroneEnv.py:835-838); dead agents are skipped."""
    case, n = "WPS_hard", 4
    from muavta_amd.params import params_from_config
    from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS
    flags = dict(WPS_ENV_FLAGS)
    flags["reward_weights"] = dict(flags["reward_weights"], action=1.0)
    p = params_from_config(CASE_SPECS[case], flags)
    from muavta_amd.batched import BatchedMultiUAVEnv
    env = BatchedMultiUAVEnv(p, n)
    env.reset(np.arange(n, dtype=np.uint64))
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    rng = np.random.default_rng(0)
    for i, o in enumerate(oracles):
        o.reset(i)
    for t in range(60):
        acts = []
        for i in range(n):
            k = int(rng.integers(0, 5))
            acts.append([(int(rng.integers(0, env.n_agents)), int(rng.integers(-2, 40))) for _ in range(k)])
        aa, ai = env.pack_actions(acts)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            o.step([a for a, _ in acts[i]], [j for _, j in acts[i]])
            compare(snap, i, o, f"random actions seed {i} t={t + 1}")


@pytest.mark.parametrize("case,tasks_per_agent", [("WPS_hard", True), ("WPS_hard", False), ("WPS_escort", True)])
def test_list_valued_actions_longer_than_the_tile(case, tasks_per_agent):
    """The reference's actions dict maps an agent to a LIST of indices and applies the items in order (DroneEnv.py:813-838):
    rows longer than the tile's action_cap go through muavta_step_lists, which applies them action_cap items at a time inside the
    one step.  Random lists of up to 3.5 x action_cap items (repeated agents, repeated tasks, invalid indices, dead agents),
    state compared with the oracle after every step."""
    from muavta_amd.params import params_from_config
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS
    from muavta_amd.batched import BatchedMultiUAVEnv
    flags = dict(WPS_ENV_FLAGS)
    flags["reward_weights"] = dict(flags["reward_weights"], action=1.0, distance=0.5, s_quality=1.0)
    flags["multiple_tasks_per_agent"] = tasks_per_agent
    ta, tt, th = TILES[case]
    p = params_from_config(CASE_SPECS[case], flags, tile_agents=ta, tile_tasks=tt, tile_threats=th)
    n = 4
    env = BatchedMultiUAVEnv(p, n)
    env.reset(np.arange(50, 50 + n, dtype=np.uint64))
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(50 + i)
    rng = np.random.default_rng(7)
    longest, compared, full = 0, 0, set()
    for t in range(40):
        acts = []
        for i in range(n):
            if i == 0 and t % 2:  # one env keeps short rows: the two entry points interleave on one handle
                k = int(rng.integers(0, 4))
                acts.append([(int(rng.integers(0, env.n_agents)), int(rng.integers(0, 12))) for _ in range(k)])
                continue
            items = []
            for a in rng.permutation(env.n_agents)[:int(rng.integers(1, env.n_agents + 1))]:
                # a list per agent, dict order; few distinct tasks (an agent's queue holds at most Q entries on a tile), some indices invalid
                items += [(int(a), int(rng.integers(-1, 5)) if rng.random() < 0.9 else 37) for _ in range(int(rng.integers(1, 10)))]
            acts.append(items[:int(3.5 * env.A_tile)])
        longest = max(longest, max(len(a) for a in acts))
        aa, ai = env.pack_actions(acts)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            if snap.ERROR[i] == 2:  # this many queued tasks per agent pass the tile's queue depth (flagged, documented): env retired from the comparison
                full.add(i)
            if i in full:
                continue
            o.step([a for a, _ in acts[i]], [j for _, j in acts[i]])
            compare(snap, i, o, f"list actions {case} seed {50 + i} t={t + 1}")
            compared += 1
    assert longest > 2 * env.n_agents and longest > env.A_tile
    assert compared >= 100 and len(full) <= 2


LIST_TRACES = sorted(glob.glob(os.path.join(GOLDEN, "lists_*.npz")))


@pytest.mark.parametrize("path", LIST_TRACES, ids=[os.path.basename(p)[6:-4] for p in LIST_TRACES])
def test_reference_list_valued_action_traces_on_the_device(path):
    """The reference's own episodes driven with list-valued actions (tests/golden/lists_*.npz, tools/gen_golden.py --lists; up to
    76 items in one step): the device applies the same rows through muavta_step_lists and must match the oracle after every step
    and the reference's positions / rewards directly."""
    from test_oracle_golden import lists_params
    from muavta_amd.batched import BatchedMultiUAVEnv
    name = os.path.basename(path)[:-4]
    # (the action-driven traces of wide-fuzz configurations — tests/fuzz_reference.py --pin-scored — run on every tile that holds the
    # fleet: WIDE1000488 needs the 40-slot tile exactly full, WIDE7178 fails on all of them without its fix)
    tile_sets = [{}] if "WIDE" not in name else [dict(tile_agents=16, tile_tasks=40, tile_threats=16), dict(tile_agents=24, tile_tasks=48, tile_threats=24),
                                                 dict(tile_agents=64, tile_tasks=128, tile_threats=48)]
    ran = 0
    for tiles in tile_sets:
        g, p = lists_params(path, **tiles)
        env = BatchedMultiUAVEnv(p, 2)  # two copies of the episode: the rows are per env
        seed = int(g["seed"])
        env.reset(np.array([seed, seed], dtype=np.uint64))
        o = orc.OracleEnv(p)
        o.reset(seed)
        acts = g["actions"]
        overflow = False
        for t in range(g["pos"].shape[0] - 1):
            ga = acts[acts[:, 0] == t]
            items = [(int(a), int(i)) for a, i in ga[:, 1:3]]
            aa, ai = env.pack_actions([items, items])
            env.step(aa, ai)
            o.step(ga[:, 1].astype(np.int32), ga[:, 2].astype(np.int32))
            snap = Snapshot(env)
            if tiles and snap.ERROR.any():  # (a smaller tile than the episode needs: only the largest must hold it)
                overflow = True
                break
            for i in range(2):
                compare(snap, i, o, f"{name} {tiles} t={t + 1}")
            assert np.array_equal(env.get("AGENT_POS")[0][:p.n_agents], g["pos"][t + 1]), f"{name} t={t + 1}: positions vs the reference"
            assert env.step_result()[0][0] == g["reward"][t + 1], f"{name} t={t + 1}: reward vs the reference"
        if overflow:
            assert tiles["tile_agents"] < 64
            continue
        ran += 1
        assert np.all(env.get("ERROR") == 0)
    assert ran >= 1


def test_agent_ids_outside_the_fleet_are_rejected():
    """An agent id >= n_agents would index the per-agent arrays of the env blob: muavta_step refuses it (MUAVTA_E_ARG)
    and leaves the state untouched (the reference's actions dict is keyed by name: an unknown name is a KeyError)."""
    from muavta_amd.native import MuavtaError
    case, n = "WPS_hard", 3
    env = _env(case, n)
    env.reset(np.arange(n, dtype=np.uint64))
    before = env.get_state()
    for bad_id in (env.n_agents, env.A_tile, 1 << 20):
        aa, ai = env.pack_actions([[(0, 1)], [(1, 0), (bad_id, 2)], []])
        with pytest.raises(MuavtaError, match="agent id"):
            env.step(aa, ai)
    assert np.array_equal(before, env.get_state())
    aa, ai = env.pack_actions([[(0, 1)], [(1, 0)], [(env.n_agents - 1, 0)]])
    aa[2, 1] = -1; aa[2, 2] = 1 << 20  # behind the terminator: ignored
    env.step(aa, ai)
    assert np.all(env.get("ERROR") == 0) and np.all(env.get("SCALARS")[:, 0] == 1)


def test_lsap_rejects_what_scipy_rejects():
    """scipy.optimize.linear_sum_assignment raises ValueError for NaN / -inf entries and for an infeasible matrix."""
    from muavta_amd.batched import lsap
    from muavta_amd.native import MuavtaError
    c = np.arange(12, dtype=np.float64).reshape(3, 4)
    for bad in (np.nan, -np.inf):
        d = c.copy(); d[1, 2] = bad
        with pytest.raises(MuavtaError, match="invalid numeric"):
            lsap(d)
    d = c.copy(); d[1, :] = np.inf
    for impl in ("registers", "lds"):
        with pytest.raises(MuavtaError, match="infeasible"):
            lsap(d, impl=impl)
    d = c.copy(); d[1, :3] = np.inf  # still feasible through column 3
    r, cc = lsap(d)
    orow, ocol = orc.lsap(d)
    assert np.array_equal(r, orow) and np.array_equal(cc, ocol)


@pytest.mark.parametrize("case,seed", [("WPS_hard", 2), ("WPS_escort", 1)])
def test_facade_on_gpu_matches_reference_observations(case, seed):
    """MultiUAVEnv facade over the HIP backend: the obs dicts / infos / rewards a PettingZoo caller sees equal
    what the reference returned (golden trace), with actions produced by the on-device allocator."""
    from muavta_amd.env import MultiUAVEnv
    from muavta_amd.params import EVENT_TAGS
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS
    g = np.load(os.path.join(GOLDEN, f"trace_{case}_s{seed}.npz"))
    ta, tt, th = TILES[case]
    env = MultiUAVEnv(CASE_SPECS[case], flags=dict(WPS_ENV_FLAGS), tile_agents=ta, tile_tasks=tt, tile_threats=th)
    obs, infos = env.reset(seed=seed)
    T = int(g["max_tasks"])
    for t in range(150):
        want = g["obs_tasks"][t]
        first = obs[env.agents_obj[0].name]
        assert len(first["tasks_info"]) == T
        for j, info in enumerate(first["tasks_info"]):
            if "id" not in info:
                assert want[j, 3] == -1
            else:
                assert info["id"] == int(want[j, 0]) and info["status"] == int(want[j, 3])
                assert np.float32(info["unmet"]) == want[j, 19] and np.float32(info["age"]) == want[j, 20]
        legal = np.unpackbits(g["obs_legal"][t], axis=-1)[:, :T].astype(bool)
        for a in env.agents_obj:
            assert obs[a.name]["legal_mask"] == list(legal[a.id])
            assert np.array_equal(np.float32(obs[a.name]["agent_position"]), g["obs_agent"][t][a.id, 0:2])
        assert np.array_equal(obs[env.agents_obj[0].name]["event_flags"], g["obs_flags"][t])
        aa, ai = env._b.allocate(int(g["interval"]), True)
        actions = {env.agents_obj[int(a)].name: int(i) for a, i in zip(aa[0], ai[0]) if a >= 0}
        obs, rew, term, trunc, infos = env.step(actions)
        assert rew[env.agents_obj[0].name] == g["reward"][t + 1]
        ev = g["events"][g["events"][:, 0] == t + 1][:, 1:]
        assert infos["events"] == [[EVENT_TAGS[int(x)], int(y)] for x, y in ev]
        vis = env.agent_visibility_map()
        NT = int(g["n_task_ids"])
        known = np.unpackbits(g["known"][t + 1], axis=-1)[:, :NT].astype(bool)
        for a in env.agents_obj:  # exact, ids of retired tasks included
            want_known = set(np.nonzero(known[a.id])[0].tolist())
            assert vis[a.name] == want_known, f"t={t + 1} {a.name}: {sorted(vis[a.name] ^ want_known)}"
    assert all(trunc.values()) and np.array_equal(np.array([float(infos["metrics"][k]) for k in METRIC_KEYS]), g["metrics"])
    assert len(env.tasks) == int(g["metrics"][13])


def test_out_of_step_mutators_hip_vs_oracle():
    """muavta_call (UAV.allocate, UAV.tasks = [...], _create_escort_for, _sync_escorts, _retire_escort, _escort_fighters_near,
    _is_task_action_valid) and the attribute writes of experiments/test_escort.py, through the facade, on the HIP backend and
    on the oracle backend side by side: every device field equal after each call, then 30 more env steps."""
    from oracle_backend import OracleBackend
    from muavta_amd.env import MultiUAVEnv
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS
    case = "WPS_escort"
    ta, tt, th = TILES[case]
    p = params_for_case(case)
    hip = MultiUAVEnv(CASE_SPECS[case], flags=dict(WPS_ENV_FLAGS), tile_agents=ta, tile_tasks=tt, tile_threats=th)
    ref = MultiUAVEnv(CASE_SPECS[case], backend=OracleBackend(p), flags=dict(WPS_ENV_FLAGS))
    envs = (hip, ref)

    def both(f):
        r = [f(e) for e in envs]
        compare(Snapshot(hip._b), 0, ref._b.o, f"after {f.__doc__}", check_obs=False)
        return r

    for e in envs:
        e.reset(seed=3)
    pick = lambda e: (next(a for a in e.agents_obj if a.type.startswith("R")), next(t for t in e.tasks if t.type == "Rec"))

    def alloc(e):
        "recon.allocate(rec)"
        recon, rec = pick(e)
        assert e._is_task_action_valid(recon, rec)
        return recon.allocate(rec, e.time_steps), recon.allocate(rec, e.time_steps)
    assert both(alloc) == [(True, False), (True, False)]

    def mk(e):
        "_create_escort_for"
        recon, rec = pick(e)
        esc = e._create_escort_for(recon, rec)
        assert esc is e._create_escort_for(recon, rec) and e._escort_by_recon[recon.name] is esc
        assert esc.kind == "Escort" and esc.eligible_agent_types == {"F1", "F2"} and not e._is_task_action_valid(recon, esc)
        assert e._is_task_action_valid(next(a for a in e.agents_obj if a.type == "F1"), esc)
        return esc.id, esc.required_agents
    r = both(mk)
    assert r[0] == r[1]

    def follow(e):
        "position writes + _sync_escorts + fighters on the escort"
        recon, _ = pick(e)
        esc = e._escort_by_recon[recon.name]
        recon.position = np.array([500.0, 400.0])
        f1 = next(a for a in e.agents_obj if a.type == "F1"); f2 = next(a for a in e.agents_obj if a.type == "F2")
        f1.position = recon.position + np.array([10.0, 0.0]); f2.position = recon.position + np.array([0.0, 10.0])
        assert f1.allocate(esc, e.time_steps) and f2.allocate(esc, e.time_steps)
        e._sync_escorts()
        assert np.array_equal(esc.position, recon.position)
        return [a.id for a in e._escort_fighters_near(recon)], [a.id for a in e._escort_fighters_near(recon, 5.0)]
    r = both(follow)
    assert r[0] == r[1] and len(r[0][0]) == 2 and r[0][1] == []

    def scaffold(e):
        "UAV.tasks = [...], UAV.state, Task.required_agents writes"
        held = {a.id for a in e._escort_fighters_near(pick(e)[0], 1e9)}
        fs = [a for a in e.get_live_agents() if a.type in ("F1", "F2") and a.id not in held][:3]
        for a in fs:
            a.tasks = [e.task_idle]   # (the reference's tests only ever assign [task_idle]: no Task bookkeeping to mirror)
            a.state = 0
        att = next(t for t in e.tasks if t.type == "Att")
        att.required_agents = 3
        return [[t.id for t in a.tasks] for a in fs], att.required_agents
    r = both(scaffold)
    assert r[0] == r[1]

    def retire(e):
        "_retire_escort"
        recon, _ = pick(e)
        esc = e._escort_by_recon[recon.name]
        e._retire_escort(esc, failed=False)
        assert esc.status == 2 and recon.name not in e._escort_by_recon
        return e.escort_completed
    assert both(retire) == [1, 1]
    for t in range(30):  # the episode goes on from the scaffolded state, allocator on each backend
        acts = []
        for e in envs:
            aa, ai = e._b.allocate(12, True)
            acts.append({e.agents_obj[int(a)].name: int(i) for a, i in zip(aa[0], ai[0]) if a >= 0})
        assert acts[0] == acts[1], f"t={t}"
        for e, ac in zip(envs, acts):
            e.step(ac)
        compare(Snapshot(hip._b), 0, ref._b.o, f"step {t} after the mutators")


# ---- next row: Urgency-Pair allocator (edge scores fused into the cost tile) ----------------------------------
@pytest.mark.parametrize("case,n", [("WPS_hard", 8), ("WPS_attn", 4), ("WPS_hard_x2", 6), ("WPS_attn_AWACS", 3)])
def test_urgency_pair_stepwise_vs_oracle(case, n):
    env = _env(case, n)
    env.set_allocator("urgency_pair")
    seeds = np.arange(n, dtype=np.uint64)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(i)
    for t in range(150):
        aa, ai = env.allocate(20, True)
        staged = env.get("STAGED_ACTIONS")
        for i, o in enumerate(oracles):
            oa, oi = o.allocate_mode(20, 1, 1)
            k = len(oa)
            assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1), f"{case} seed {i} t={t}: agents {aa[i]} vs {oa}"
            assert np.array_equal(ai[i][:k], oi) and np.array_equal(staged[i][:k, 1], o.last_actions()[:, 1]), f"{case} seed {i} t={t}"
            o.step(oa, oi)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"urgency-pair {case} seed {i} t={t + 1}")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "urgpair_metrics_*.npz"))),
                         ids=lambda p: os.path.basename(p)[16:-4])
def test_urgency_pair_fused_rollout_matches_reference(path):
    g = np.load(path)
    case = os.path.basename(path)[16:-4]
    want = g["metrics"]
    n = want.shape[0]
    env = _env(case, n)
    env.set_allocator("urgency_pair")
    env.rollout(np.arange(n, dtype=np.uint64), 150, 20, True, True)
    got = env.rollout_metrics()
    assert np.all(env.get("ERROR") == 0)
    assert np.array_equal(got, want), f"{case}: seeds {np.nonzero(~np.all(got == want, axis=1))[0][:8]} differ"
    assert np.array_equal(env.get("SCALARS")[:, 23].astype(int), g["n_replans"])
    env.set_allocator("hungarian")  # and back: the default path is untouched
    gm = np.load(os.path.join(GOLDEN, f"metrics_{case}.npz"))
    m = min(n, gm["metrics"].shape[0])
    env.rollout(np.arange(n, dtype=np.uint64), 150, int(gm["interval"]), True, False)
    assert np.array_equal(env.rollout_metrics()[:m], gm["metrics"][:m])


# ---- next row: Urgency-Coalition allocator (threat-pressure edge scores + commit locks) ---------------------------
@pytest.mark.parametrize("case,n,interval", [("WPS_escort", 6, 12), ("WPS_escort24", 3, 12), ("WPS_hard", 4, 12), ("WPS_burst64", 2, 12)])
def test_urgency_coalition_stepwise_vs_oracle(case, n, interval):
    env = _env(case, n)
    env.set_allocator("urgency_coalition")
    seeds = np.arange(n, dtype=np.uint64)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(i)
    for t in range(150):
        aa, ai = env.allocate(interval, True)
        staged = env.get("STAGED_ACTIONS")
        commit = env.get("AGENT_MISC")[:, :, 4]
        for i, o in enumerate(oracles):
            oa, oi = o.allocate_mode(interval, 1, 2)
            k = len(oa)
            assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1), f"{case} seed {i} t={t}: agents {aa[i]} vs {oa}"
            assert np.array_equal(ai[i][:k], oi) and np.array_equal(staged[i][:k, 1], o.last_actions()[:, 1]), f"{case} seed {i} t={t}"
            assert np.array_equal(commit[i][:env.n_agents], o.agent_commit_until()), f"{case} seed {i} t={t}: commit locks"
            o.step(oa, oi)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"urgency-coalition {case} seed {i} t={t + 1}")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "urgcoal_metrics_*.npz"))),
                         ids=lambda p: os.path.basename(p)[16:-4])
def test_urgency_coalition_fused_rollout_matches_reference(path):
    g = np.load(path)
    case = os.path.basename(path)[16:-4]
    want = g["metrics"]
    n = want.shape[0]
    env = _env(case, n)
    env.set_allocator("urgency_coalition")
    env.rollout(np.arange(n, dtype=np.uint64), 150, int(g["interval"]), True, True)
    got = env.rollout_metrics()
    assert np.all(env.get("ERROR") == 0)
    assert np.array_equal(got, want), f"{case}: seeds {np.nonzero(~np.all(got == want, axis=1))[0][:8]} differ"
    assert np.array_equal(env.get("SCALARS")[:, 23].astype(int), g["n_replans"])


# ---- next row: token builders (pair / raw / escort tokens) straight from the device state ------------------------
TOKEN_FILES = sorted(glob.glob(os.path.join(GOLDEN, "tokens_*.npz")))
KIND_NAME = {0: "pair", 1: "pair_raw", 2: "escort"}


@pytest.mark.parametrize("path", TOKEN_FILES, ids=[os.path.basename(p)[7:-4] for p in TOKEN_FILES])
def test_token_builders_vs_reference_and_oracle(path):
    from tokcheck import check_tokens

    g = np.load(path)
    case = os.path.basename(path)[7:-4]
    seed0, interval = int(g["seed"]), int(g["interval"])
    n = 3
    env = _env(case, n)
    mode = 2 if str(g["driver"]) == "urgcoal" else 0
    env.set_allocator("urgency_coalition" if mode == 2 else "hungarian")
    env.reset(np.arange(seed0, seed0 + n, dtype=np.uint64))
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(seed0 + i)
    steps = g["step"].tolist()
    for t in range(150):
        aa, ai = env.allocate(interval, True)
        for o in oracles:
            o.allocate_mode(interval, 1, mode)
        if t in steps:
            cache = {}

            def dev_tok(i):
                def f(kind, mt, ma):
                    key = (kind, mt, ma)
                    if key not in cache:
                        cache[key] = env.tokens(KIND_NAME[kind], mt, ma)
                    return {k: (int(v[i]) if k == "n_urgent" else v[i]) for k, v in cache[key].items()}
                return f

            check_tokens(dev_tok(0), g, steps.index(t), f"{case} t={t} (reference)")
            for i, o in enumerate(oracles):  # the other seeds: against the oracle, same checker
                for kind, mt, ma in ((0, 32, 16), (1, 32, 16), (2, int(g["e_max_tasks"]), int(g["e_max_agents"])), (0, 20, 6), (2, 12, 64)):
                    want, got = o.tokens(kind, mt, ma), dev_tok(i)(kind, mt, ma)
                    for k in want:
                        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), f"{case} seed {seed0 + i} t={t} kind {kind} {mt}x{ma}: {k}"
        env.step(aa, ai)
        for i, o in enumerate(oracles):
            k = int(np.sum(aa[i] >= 0))
            o.step(aa[i][:k], ai[i][:k])


def test_tokens_into_torch_tensors_on_device():
    import torch

    env = _env("WPS_hard_x2", 64)
    env.rollout(np.arange(64, dtype=np.uint64), 40, 20, True, False)
    want = env.tokens("pair")
    dt = {"task_feats": torch.float32, "task_mask": torch.uint8, "task_ids": torch.int32, "agent_feats": torch.float32,
          "agent_mask": torch.uint8, "agent_ids": torch.int32, "edge_valid": torch.float32, "n_urgent": torch.int32,
          "expert_mask": torch.float32, "replanned": torch.int32}
    out = {k: torch.empty(v.shape, dtype=dt[k], device="cuda") for k, v in want.items()}
    env.tokens("pair", out=out)
    env.sync()
    for k, v in want.items():
        assert np.array_equal(out[k].cpu().numpy(), v), k
    assert (want["task_mask"] == 0).any() and (want["edge_valid"] > 0).any()


# ---- next row: replay / frame export in the dashboard schema -----------------------------------------------------
@pytest.mark.parametrize("case,allocator,seed", [("WPS_escort", "urgency_coalition", 3), ("WPS_hard", "urgency_pair", 1)])
def test_replay_document_hip_backend_equals_oracle_backend(case, allocator, seed):
    """replay.generate over the HIP backend == the same over the oracle backend (which tests/test_facade_cpu.py pins
    to the reference's generate_simulation_replay.py document in the build container)."""
    from muavta_amd import replay
    from muavta_amd.env import MultiUAVEnv
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS
    from oracle_backend import OracleBackend

    ta, tt, th = TILES[case]
    hip_env = MultiUAVEnv(CASE_SPECS[case], flags=dict(WPS_ENV_FLAGS), tile_agents=ta, tile_tasks=tt, tile_threats=th)
    got = replay.generate(seed, None, case, allocator, env=hip_env)
    orc_env = MultiUAVEnv(CASE_SPECS[case], backend=OracleBackend(params_for_case(case)), flags=dict(WPS_ENV_FLAGS))
    want = replay.generate(seed, None, case, allocator, env=orc_env)
    assert got["metadata"] == want["metadata"] and len(got["frames"]) == 151
    for k, (a, b) in enumerate(zip(got["frames"], want["frames"])):
        for key in a:
            assert a[key] == b[key], f"{case} frame {k} {key}"
    assert got["events"] == want["events"] and got["final_metrics"] == want["final_metrics"]


# ---- next row: the trainers' imitation-learning data loop, batched ----------------------------------------------
IL_FILES = sorted(glob.glob(os.path.join(GOLDEN, "il_*.npz")))


@pytest.mark.parametrize("path", IL_FILES, ids=[os.path.basename(p)[3:-4] for p in IL_FILES])
def test_il_stream_vs_reference_and_oracle(path):
    from muavta_amd.il import il_stream

    g = np.load(path)
    case = os.path.basename(path)[3:-4]
    seed0, n = int(g["seed"]), 3
    env = _env(case, n)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(seed0 + i)
    steps, k = g["step"].tolist(), 0
    for t, b in il_stream(env, np.arange(seed0, seed0 + n), 150, 20, "pair", 32, 16, with_reward=True):
        # env 0: the reference's own samples
        assert int(b["replanned"][0]) == int(g["replanned"][t]), f"{case} t={t}"
        if g["replanned"][t]:
            assert t == steps[k]
            assert np.array_equal(b["expert_mask"][0], g["mask"][k]), f"{case} t={t}: expert mask"
            assert np.array_equal(b["task_feats"][0], g["tf"][k]) and np.array_equal(b["agent_feats"][0], g["af"][k])
            assert np.array_equal(b["edge_valid"][0], g["ev"][k]) and np.array_equal(b["task_ids"][0], g["tid"][k])
            k += 1
        else:
            assert not b["expert_mask"][0].any()
        assert b["step_reward"][0] == ((g["s_wps"][t] - g["s_wps"][t - 1]) / 20.0 if t else 0.0)
        # the other seeds: the oracle playing the same loop
        for i, o in enumerate(oracles):
            aa, ai = o.allocate_mode(20, 0, 3)
            tok = o.tokens(0, 32, 16)
            for key in ("expert_mask", "task_feats", "agent_feats", "edge_valid", "task_ids", "agent_ids", "task_mask", "agent_mask"):
                assert np.array_equal(b[key][i], tok[key]), f"{case} seed {seed0 + i} t={t}: {key}"
            assert int(b["replanned"][i]) == int(o.dims()["time_steps"] == o.scalars_last_plan()), f"{case} seed {seed0 + i} t={t}"
            o.step(aa, ai)
    assert k == len(steps)
    assert np.array_equal(env.metrics()[0], g["metrics"])
    for i, o in enumerate(oracles):
        assert np.array_equal(env.metrics()[i], o.metrics())


@pytest.mark.parametrize("case,kind,mt,ma", [("WPS_hard", "pair", 32, 16), ("WPS_hard_x2", "pair_raw", 32, 16), ("WPS_escort", "escort", 48, 16)])
def test_il_record_rings_equal_the_per_step_stream(case, kind, mt, ma):
    """muavta_rollout_record (one launch, per-step rings on the device) against il_stream (three launches per step, already
    pinned to the reference's samples above): every ring slot bit-equal, S_WPS series and final metrics included."""
    import torch
    from muavta_amd.il import il_record, il_stream
    n, steps = 5, 150
    seeds = np.arange(11, 11 + n)
    interval = 12 if "escort" in case else 20
    env = _env(case, n)
    rec = il_record(env, seeds, steps, interval, kind, mt, ma)
    m_rec = env.rollout_metrics()
    assert not env.get("ERROR").any()
    got = {k: v.cpu().numpy() for k, v in rec.items()}
    env2 = _env(case, n)
    prev = None
    for t, b in il_stream(env2, seeds, steps, interval, kind, mt, ma, with_reward=True):
        for key in ("task_feats", "task_mask", "task_ids", "agent_feats", "agent_mask", "agent_ids", "edge_valid", "n_urgent", "expert_mask", "replanned"):
            assert np.array_equal(got[key][t], b[key]), f"{case} t={t}: {key}"
        if t:
            assert np.array_equal(got["step_reward"][t - 1], b["step_reward"]), f"{case} t={t}: step reward"
    assert np.array_equal(got["s_wps"][steps], env2.metrics()[:, 4]) and np.array_equal(m_rec, env2.metrics())
    assert isinstance(rec["task_feats"], torch.Tensor) and rec["task_feats"].is_cuda and rec["step_reward"].shape == (steps, n)


@pytest.mark.parametrize("case,interval", [("WPS_hard", 20), ("WPS_escort24", 12), ("WPS_burst64", 20)])
def test_observation_rings_of_the_fused_rollout_vs_oracle(case, interval):
    """muavta_rollout_record with observation rings: slot t holds exactly what DroneEnv.step would have returned from step t
    (observation dict, reward, done flags) — checked against the oracle stepping the same seeds — instead of every step
    overwriting one buffer; slots after an env's last step stay MUAVTA_OBS_UNWRITTEN; the handle's own buffer gets the final
    observation; metrics equal the ring-less rollout's."""
    import torch
    n, steps = 4, 150
    seeds = np.arange(70, 70 + n, dtype=np.uint64)
    env = _env(case, n)
    dev = torch.device("cuda", env.device_index)
    rings = {k: torch.zeros(shape, dtype=getattr(torch, np.dtype(dt).name), device=dev) for k, (shape, dt) in env.obs_ring_shapes(steps).items()}
    env.rollout_record(seeds, steps, interval, True, obs_rings=rings)
    env.sync()
    m = env.rollout_metrics()
    snap = Snapshot(env)
    R = {k: v.cpu().numpy() for k, v in rings.items()}
    MT = env.max_tasks
    for i in range(n):
        o = orc.OracleEnv(params_for_case(case))
        o.reset(int(seeds[i]))
        t_end = steps
        for t in range(steps):
            a, ix = o.allocate_mode(interval, 1, 0)
            o.step(a, ix)
            ti, legal, pad, ag, fl = o.observe()
            tag = f"{case} seed {seeds[i]} ring slot {t}"
            assert np.array_equal(R["obs_tasks"][t, i].T, ti), f"{tag}: tasks_info"
            bits = ((R["obs_legal"][t, i][:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(legal.shape[0], -1)[:, :MT]
            assert np.array_equal(bits.astype(bool), legal), f"{tag}: legal_mask"
            assert np.array_equal(R["obs_pad"][t, i].astype(bool), pad), f"{tag}: pad mask"
            assert np.array_equal(R["obs_agents"][t, i], ag), f"{tag}: agent rows"
            assert np.array_equal(R["obs_flags"][t, i], fl), f"{tag}: event flags"
            d = o.dims()
            assert R["obs_reward"][t, i] == o.scalars()[1], f"{tag}: reward"
            assert R["obs_done"][t, i] == (1 if d["terminated"] else 0) | (2 if d["truncated"] else 0), f"{tag}: done flags"
            if d["terminated"] or d["truncated"]:
                t_end = t + 1
                break
        assert np.all(R["obs_done"][t_end:, i] == env.OBS_UNWRITTEN), f"{case} seed {seeds[i]}: slots after the last step"
        compare(snap, i, o, f"{case} seed {seeds[i]} after the recorded rollout")
    env2 = _env(case, n)
    env2.rollout(seeds, steps, interval, True, True)
    assert np.array_equal(m, env2.rollout_metrics())
    with pytest.raises(Exception):  # all seven rings or none
        env.rollout_record(seeds, steps, interval, True, obs_rings={k: v for k, v in rings.items() if k != "obs_done"})


@pytest.mark.parametrize("lanes", [1, 0], ids=["one-lane", "default-lanes"])
def test_queued_rollouts_overlap_seeding_and_keep_their_event_pairs(lanes):
    """Rollouts queued back to back (the seeding of launch i+1 runs on the handle's second stream under launch i, through two
    slots — and, with the default lanes, launch i+1 itself on the handle's other state lane) give the same metrics as synchronised
    ones, for alternating seed sets; kernel_ms_history returns one duration per launch, the newest equal to last_kernel_ms."""
    case, n = "WPS_hard_x2", 64
    env = _env(case, n)
    sets = [np.arange(k * 100, k * 100 + n, dtype=np.uint64) for k in range(5)]
    want = []
    for sd in sets:
        env.rollout(sd, 150, 20, True, True)
        env.sync()
        want.append(env.rollout_metrics())
    env2 = _env(case, n)
    env2.set_lanes(lanes)
    got = []
    for sd in sets:
        env2.rollout(sd, 150, 20, True, True)   # no sync in between
        got.append(None)
    ms = env2.kernel_ms_history(len(sets))
    assert ms.shape == (len(sets),) and np.all(ms > 0) and abs(float(ms[-1]) - env2.last_kernel_ms()) < 1e-6
    assert np.array_equal(env2.rollout_metrics(), want[-1])
    # interleaved: queue two launches, read the metrics of each through a fresh handle's synchronous run
    env3 = _env(case, n)
    env3.set_lanes(lanes)
    for k, sd in enumerate(sets):
        env3.rollout(sd, 150, 20, True, True)
        env3.reset(sets[(k + 1) % len(sets)])  # a reset in between takes the other seeding slot
        env3.rollout(sd, 150, 20, True, False)
        assert np.array_equal(env3.rollout_metrics(), want[k]), f"seed set {k}"
    with pytest.raises(Exception):
        env3.kernel_ms_history(65)


# ---- fuzzed configurations (knob combinations no registry case has); reference traces in tests/golden/trace_FUZZ* ----
FUZZ_TRACES = sorted(glob.glob(os.path.join(GOLDEN, "trace_FUZZ*.npz")))


@pytest.mark.parametrize("path", FUZZ_TRACES, ids=[os.path.basename(p)[6:-4] for p in FUZZ_TRACES])
def test_fuzzed_config_stepwise_vs_oracle_and_reference_metrics(path):
    from cases import params_of
    from muavta_amd.batched import BatchedMultiUAVEnv

    g = np.load(path)
    case, seed0 = os.path.basename(path)[6:-4].rsplit("_s", 1)
    seed0, interval, n = int(seed0), int(g["interval"]), 4
    small = int(g["n_task_ids"]) <= 40  # few task objects: also run on the tightest tile that fits the fleet
    for tiles in ([dict(tile_agents=16, tile_tasks=128, tile_threats=16)] + ([dict(tile_agents=16, tile_tasks=48, tile_threats=16)] if small else [])):
        p = params_of(case, **tiles)
        env = BatchedMultiUAVEnv(p, n)
        seeds = np.arange(seed0, seed0 + n, dtype=np.uint64)
        oracles = [orc.OracleEnv(p) for _ in range(n)]
        env.reset(seeds)
        for i, o in enumerate(oracles):
            o.reset(int(seeds[i]))
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"{case} seed {seeds[i]} after reset")
        for t in range(p.max_time_steps):
            aa, ai = env.allocate(interval, True)
            for i, o in enumerate(oracles):
                if o.dims()["terminated"] or o.dims()["truncated"]:
                    continue
                oa, oi = o.allocate(interval, 1)
                k = len(oa)
                assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1) and np.array_equal(ai[i][:k], oi), f"{case} seed {seeds[i]} t={t}"
                o.step(oa, oi)
            live = [i for i, o in enumerate(oracles)]
            env.step(aa, ai)
            snap = Snapshot(env)
            done_all = True
            for i, o in enumerate(oracles):
                d = o.dims()
                if not (d["terminated"] or d["truncated"]) or d["time_steps"] == t + 1:
                    compare(snap, i, o, f"{case} seed {seeds[i]} t={t + 1}")
                done_all &= bool(d["terminated"] or d["truncated"])
            if done_all or (oracles[0].dims()["terminated"] or oracles[0].dims()["truncated"]):
                break
        assert np.array_equal(env.metrics()[0], g["metrics"]), f"{case}: reference metrics"
        assert np.all(env.get("ERROR") == 0)


WIDE_TRACES = sorted(glob.glob(os.path.join(GOLDEN, "trace_WIDE*.npz")))


@pytest.mark.parametrize("path", WIDE_TRACES, ids=[os.path.basename(p)[6:-4] for p in WIDE_TRACES])
def test_wide_fuzz_regressions_stepwise_vs_oracle_and_reference_metrics(path):
    """Configurations on which the wide fuzz (tests/fuzz_device.py) found device bugs, pinned as reference traces: MORE open tasks
    than max_tasks (the observation tensor keeps the first max_tasks rows; r4's incremental row writer stored the rows beyond
    them into the next feature columns), and an escort task that expired by its hard window while its map entry lives on (it
    keeps following its UAV, DroneEnv.py:1995).  Every tile that holds the episode, stepwise against the oracle; env 0 is the
    reference's own episode (final metrics)."""
    from cases import params_of
    from muavta_amd.batched import BatchedMultiUAVEnv

    g = np.load(path)
    case, seed0 = os.path.basename(path)[6:-4].rsplit("_s", 1)
    seed0, interval, n = int(seed0), int(g["interval"]), 2
    ran = 0
    for tiles in (dict(tile_agents=16, tile_tasks=40, tile_threats=16), dict(tile_agents=24, tile_tasks=48, tile_threats=24),
                  dict(tile_agents=64, tile_tasks=128, tile_threats=48)):
        p = params_of(case, **tiles)
        env = BatchedMultiUAVEnv(p, n)
        seeds = np.array([seed0, seed0 - 1 if seed0 > 2 ** 62 else seed0 + 1], dtype=np.uint64)
        oracles = [orc.OracleEnv(p) for _ in range(n)]
        env.reset(seeds)
        for i, o in enumerate(oracles):
            o.reset(int(seeds[i]))
        overflow = False
        for t in range(p.max_time_steps):
            if any(o.dims()["terminated"] or o.dims()["truncated"] for o in oracles):
                break
            aa, ai = env.allocate(interval, True)
            for i, o in enumerate(oracles):
                oa, oi = o.allocate(interval, 1)
                k = len(oa)
                assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1) and np.array_equal(ai[i][:k], oi), f"{case} {tiles} seed {seeds[i]} t={t}"
                o.step(oa, oi)
            env.step(aa, ai)
            snap = Snapshot(env)
            if snap.ERROR.any():  # the episode needs more than this tile (only the largest one must hold it)
                overflow = True
                break
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"{case} {tiles} seed {seeds[i]} t={t + 1}")
        if overflow:
            assert tiles["tile_agents"] < 64, f"{case}: overflows the largest tile"
            continue
        ran += 1
        if oracles[0].dims()["terminated"] or oracles[0].dims()["truncated"]:
            assert np.array_equal(env.metrics()[0], g["metrics"]), f"{case} {tiles}: reference metrics"
    assert ran >= 1


@pytest.mark.parametrize("leg,k", [("mutators", 12794), ("mutators", 12782), ("mutators", 14016), ("mutators", 14068), ("mutators", 14173),
                                   ("scored", 7178), ("scored", 32517), ("mutators", 1001642), ("stepwise", 992), ("stepwise", 40), ("rl", 12041), ("lists", 12042), ("rings", 12043),
                                   ("resume", 17001), ("ilrings", 18801), ("inflight", 21001), ("rlrun", 22001), ("rlrun", 1003001), ("steprun", 22002),
                                   ("steprun", 2002001), ("lanes", 22003)])
def test_wide_fuzz_legs_on_the_configurations_that_found_bugs(leg, k):
    """One episode of a leg of tests/fuzz_device.py on the configurations that exposed device bugs (the allocator's list after an
    out-of-step _retire_escort / _create_escort_for, the recon-as-escort retire verdicts, rows beyond max_tasks, the expired escort
    that follows its UAV, the agent steered onto an obstacle's axis where the last bit of atan2 picks the side — 32517) plus one plain configuration per remaining leg, so that the fuzz driver itself stays runnable."""
    import fuzz_device as FD
    from fuzz_reference import wide_config

    msgs = []
    w = wide_config(k)
    out = getattr(FD, leg)(k, w, msgs.append)
    assert out in ("ok", "overflow") and not msgs, msgs


def test_seeds_beyond_32_bits():
    """init_by_array with a two-word key (seed >= 2^32): the batched seeding kernel against the oracle's CPython restatement."""
    case = "WPS_hard"
    seeds = np.array([2**32, 2**40 + 12345, 2**63 - 1, 2**62 + 7, 2**32 - 1, 0], dtype=np.uint64)
    env = _env(case, len(seeds))
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in seeds]
    env.reset(seeds)
    for o, sd in zip(oracles, seeds):
        o.reset(int(sd))
    snap = Snapshot(env)
    for i, o in enumerate(oracles):
        compare(snap, i, o, f"seed {seeds[i]} after reset")
    for t in range(60):
        aa, ai = env.allocate(20, True)
        for i, o in enumerate(oracles):
            oa, oi = o.allocate(20, 1)
            assert np.array_equal(aa[i][:len(oa)], oa) and np.array_equal(ai[i][:len(oa)], oi)
            o.step(oa, oi)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"seed {seeds[i]} t={t + 1}")


@pytest.mark.parametrize("mode,name", [(1, "urgency_pair"), (2, "urgency_coalition"), (3, "hungarian_gated")])
def test_allocator_modes_on_fuzzed_configs_vs_oracle(mode, name):
    """The hybrid / trainer allocator modes under knob combinations no registry case has: device vs oracle, stepwise."""
    from cases import fuzz_configs, params_of
    from muavta_amd.batched import BatchedMultiUAVEnv

    for case in sorted(fuzz_configs()):
        p = params_of(case)
        n = 2
        env = BatchedMultiUAVEnv(p, n)
        env.set_allocator(name)
        seeds = np.arange(7, 7 + n, dtype=np.uint64)
        oracles = [orc.OracleEnv(p) for _ in range(n)]
        env.reset(seeds)
        for i, o in enumerate(oracles):
            o.reset(int(seeds[i]))
        interval = 12 if p.escort_enabled else 20
        for t in range(min(p.max_time_steps, 90)):
            aa, ai = env.allocate(interval, True)
            done = [bool(o.dims()["terminated"] or o.dims()["truncated"]) for o in oracles]
            if any(done):
                break
            for i, o in enumerate(oracles):
                oa, oi = o.allocate_mode(interval, 1, mode)
                k = len(oa)
                assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1) and np.array_equal(ai[i][:k], oi), f"{case} {name} seed {seeds[i]} t={t}"
                o.step(oa, oi)
            env.step(aa, ai)
            snap = Snapshot(env)
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"{case} {name} seed {seeds[i]} t={t + 1}")
        assert np.all(env.get("ERROR") == 0)


def test_large_bursts_on_the_64_agent_tile():
    """burst_size x n_agents beyond one pass of the closest-agent scratch (64 agents x 6 threats): chunked precompute."""
    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.params import params_from_config
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS

    spec = dict(CASE_SPECS["WPS_burst64"]); spec["burst_size"] = 6
    ta, tt, th = TILES["WPS_burst64"]
    p = params_from_config(spec, dict(WPS_ENV_FLAGS), tile_agents=ta, tile_tasks=tt, tile_threats=th)
    n = 2
    env = BatchedMultiUAVEnv(p, n)
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    env.reset(np.arange(n, dtype=np.uint64))
    for i, o in enumerate(oracles):
        o.reset(i)
    for t in range(100):
        aa, ai = env.allocate(20, True)
        for i, o in enumerate(oracles):
            oa, oi = o.allocate(20, 1)
            assert np.array_equal(aa[i][:len(oa)], oa) and np.array_equal(ai[i][:len(oa)], oi), f"seed {i} t={t}"
            o.step(oa, oi)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"burst 6 seed {i} t={t + 1}")


@pytest.mark.parametrize("case,name,mode,interval,vis,n", [("WPS_hard_x2", "urgency_pair", 1, 20, 1, 2048), ("WPS_escort24", "urgency_coalition", 2, 12, 1, 512),
                                                            ("WPS_escort", "urgency_coalition", 2, 12, 1, 1024), ("WPS_hard_x2", "hungarian_gated", 3, 20, 0, 2048),
                                                            ("WPS_burst64", "urgency_coalition", 2, 12, 1, 128)])
def test_fused_rollout_allocator_modes_vs_oracle_many_seeds(case, name, mode, interval, vis, n):
    # Urgency-Pair keeps more tasks open on the 16-UAV workload: 5 of 8192 seeds need more than 32 slots / 8 queue entries
    env = _env(case, n)
    env.set_allocator(name)
    seeds = np.arange(5000, 5000 + n, dtype=np.uint64)
    env.rollout(seeds, 150, interval, bool(vis), False)
    got = env.rollout_metrics()
    assert np.all(env.get("ERROR") == 0)
    o = orc.OracleEnv(params_for_case(case))
    for i, s in enumerate(seeds):
        o.rollout_mode(int(s), 150, interval, vis, mode)
        assert np.array_equal(got[i], o.metrics()), f"{case} {name} seed {s}"


def test_16_agent_tile_holds_the_seeds_that_need_more_than_32_slots():
    """The 16-agent tile has 40 task slots and queues of 10 (the reference stops creating arrivals at max_tasks - 1 = 40
    tasks for a 16-UAV config): stepwise parity on the seeds that need more than 32 live slots / 8 queue entries."""
    case = "WPS_hard_x2"
    env = _env(case, 4)
    assert env.T == 40 and env.Q == 10
    seeds = np.array([9649, 6231, 0, 1], dtype=np.uint64)  # 9649: 34 slots under Local-Hungarian
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in seeds]
    env.reset(seeds)
    for o, sd in zip(oracles, seeds):
        o.reset(int(sd))
    for t in range(150):
        aa, ai = env.allocate(20, True)
        for i, o in enumerate(oracles):
            oa, oi = o.allocate(20, 1)
            assert np.array_equal(aa[i][:len(oa)], oa) and np.array_equal(ai[i][:len(oa)], oi), f"seed {seeds[i]} t={t}"
            o.step(oa, oi)
        env.step(aa, ai)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"16x40 tile seed {seeds[i]} t={t + 1}")
    for name, mode, sd in (("urgency_pair", 1, [6231, 8273, 9649, 11430, 11656]),):
        big = _env(case, len(sd))
        big.set_allocator(name)
        big.rollout(np.array(sd, dtype=np.uint64), 150, 20, True, False)
        assert not big.get("ERROR").any()
        o = orc.OracleEnv(params_for_case(case))
        for i, s_ in enumerate(sd):
            o.rollout_mode(int(s_), 150, 20, 1, mode)
            assert np.array_equal(big.rollout_metrics()[i], o.metrics()), f"{name} seed {s_}"


# ---- multi-GPU readiness on one GPU -------------------------------------------------------------------
def test_comm_abi_single_rank():
    """muavta_comm_uid / muavta_comm_init / muavta_allreduce_metrics / muavta_comm_destroy over RCCL with one rank."""
    from muavta_amd.dist import partial_sums, reduce_metrics
    from muavta_amd.native import MuavtaError
    env = _env("WPS_hard", 64)
    env.rollout(np.arange(64, dtype=np.uint64), 150, 20, True, False)
    m = env.rollout_metrics()
    with pytest.raises(MuavtaError, match="before muavta_comm_init"):
        env.allreduce_metrics(np.zeros(2), np.zeros(2, dtype=np.int64))
    uid = env.comm_uid()
    assert len(uid) == 128 and uid != env.comm_uid()
    env.comm_init(0, 1, uid)
    with pytest.raises(MuavtaError, match="already has a communicator"):
        env.comm_init(0, 1, uid)
    f, c = partial_sums(m)
    fo, co = env.allreduce_metrics(f, c)
    assert np.array_equal(fo, f) and np.array_equal(co, c)
    assert reduce_metrics(m, comm=env) == reduce_metrics(m)
    env.comm_destroy()
    env.comm_init(0, 1, env.comm_uid())  # a handle can join again after destroy
    env.close()


@pytest.mark.parametrize("extra", [[], ["--abi-collective"], ["--lanes", "1"]], ids=["torch-nccl", "abi-rccl", "one-state-lane"])
def test_bench_under_torchrun_world_size_1(extra):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, nccl backend = RCCL), at world size 1:
    sharding, barrier-bracketed timing, metric reduction and the single JSON line."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--envs", "512",
           "--no-cpu-baseline", "--no-extras"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["total_envs"] == 512 and out["quality"]["n_envs"] == 512
    assert out["lanes"]["allocated"] == (1 if "--lanes" in extra else 2)  # (launches queued back to back: the default handle runs them on two state lanes)
    want = orc.parallel_metrics("WPS_hard_x2", np.arange(512), 20)
    assert out["quality"]["mean_S_WPS"] == float(want[:, 4].sum()) / 512


def test_il_record_rings_of_an_episode_that_ends_early_are_fully_written():
    """muavta_rollout_record when episodes end before n_steps (early_terminate, and n_steps beyond max_time_steps): the
    reference's episode loops stop at `done` (train_pair_cost.py:108,139), so the slots behind an env's last step are
    all-pad rows with replanned = 0 and a zero step reward — never uninitialised ring memory (rings pre-filled with NaN /
    0x7f bytes here) — and the slots up to the end equal the per-step stream."""
    import torch
    from muavta_amd.batched import BatchedMultiUAVEnv
    from muavta_amd.il import il_record, il_stream
    n, steps, interval, kind, mt, ma = 48, 160, 20, "pair", 32, 16   # 160 > max_time_steps = 150: every env is truncated at 150
    p = params_for_case("WPS_easy")
    p.early_terminate = 1
    env = BatchedMultiUAVEnv(p, n)
    seeds = np.arange(300, 300 + n)
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32, np.float64: torch.float64}
    rings = {}
    for name, (shape, dtype) in env.record_shapes(kind, steps, mt, ma).items():
        t = torch.empty(shape, dtype=tdt[dtype], device="cuda")
        t.fill_(float("nan")) if t.is_floating_point() else t.fill_(0x7f)
        rings[name] = t
    rec = il_record(env, seeds, steps, interval, kind, mt, ma, rings=rings)
    assert not env.get("ERROR").any()
    got = {k: v.cpu().numpy() for k, v in rec.items()}
    ended_at = env.get("SCALARS")[:, 0].astype(int)          # time_steps when the episode ended
    assert ended_at.max() <= 150 and (ended_at < 150).any(), "no env of the batch terminated early: pick other seeds"
    for k in ("task_feats", "agent_feats", "edge_valid", "expert_mask", "s_wps", "step_reward"):
        assert np.isfinite(got[k]).all(), f"{k} holds uninitialised values"
    env2 = BatchedMultiUAVEnv(p, n)
    for t, b in il_stream(env2, seeds, 150, interval, kind, mt, ma, with_reward=True):
        live = ended_at > t                                   # sample t exists iff the episode has not ended after t steps
        for key in ("task_feats", "task_mask", "task_ids", "agent_feats", "agent_mask", "agent_ids", "edge_valid", "n_urgent", "expert_mask", "replanned"):
            assert np.array_equal(got[key][t][live], b[key][live]), f"t={t}: {key}"
        if t:
            assert np.array_equal(got["step_reward"][t - 1][ended_at >= t], b["step_reward"][ended_at >= t]), f"t={t}: step reward"
    for i in range(n):
        e = ended_at[i]
        assert (got["replanned"][e:, i] == 0).all() and (got["n_urgent"][e:, i] == 0).all()
        assert (got["task_mask"][e:, i] == 1).all() and (got["agent_mask"][e:, i] == 1).all()
        assert (got["task_ids"][e:, i] == -1).all() and (got["agent_ids"][e:, i] == -1).all()
        assert not got["task_feats"][e:, i].any() and not got["agent_feats"][e:, i].any() and not got["edge_valid"][e:, i].any() and not got["expert_mask"][e:, i].any()
        assert (got["s_wps"][e:, i] == got["s_wps"][steps, i]).all() and not got["step_reward"][e:, i].any()
    assert np.array_equal(got["s_wps"][steps], env.rollout_metrics()[:, 4])


def test_config5_global_indices_0_8191_in_eight_shards():
    """BASELINE config 5 as the 8-GPU job shards it: global env indices 0..8191 of WPS_burst64, 1024 per shard (= per GPU).
    Every env produces a result (ERROR == 0) and all 30 metrics equal the oracle's."""
    case, shard = "WPS_burst64", 1024
    env = _env(case, shard)
    want = orc.parallel_metrics(case, np.arange(8192), 20)
    for r in range(8):
        seeds = np.arange(r * shard, (r + 1) * shard, dtype=np.uint64)
        env.rollout(seeds, 150, 20, True, True)
        got, err = env.rollout_metrics(), env.get("ERROR")
        assert not err.any(), f"shard {r}: envs {seeds[np.nonzero(err)[0][:8]]} overflowed the tile (codes {np.unique(err[err != 0])})"
        bad = np.nonzero(~np.all(got == want[r * shard:(r + 1) * shard], axis=1))[0]
        assert len(bad) == 0, f"shard {r}: seeds {seeds[bad[:8]]} differ"


@pytest.mark.parametrize("name,mode,n", [("hungarian", 0, 4096), ("urgency_coalition", 2, 1024)])
def test_config4_bench_seed_range(name, mode, n):
    """BASELINE config 4 on the seeds bench.py actually runs (global env index 0..4095), and the same sweep under
    Urgency-Coalition at 1024 envs: ERROR == 0 and 30 metrics bit-equal to the oracle for every env."""
    case, interval = "WPS_escort24", 12
    env = _env(case, n)
    env.set_allocator(name)
    seeds = np.arange(n, dtype=np.uint64)
    env.rollout(seeds, 150, interval, True, True)
    got, err = env.rollout_metrics(), env.get("ERROR")
    assert not err.any(), f"{name}: envs {seeds[np.nonzero(err)[0][:8]]} overflowed the tile (codes {np.unique(err[err != 0])})"
    want = orc.parallel_metrics(case, seeds, interval, 1, mode)
    bad = np.nonzero(~np.all(got == want, axis=1))[0]
    assert len(bad) == 0, f"{name}: seeds {seeds[bad[:8]]} differ"


@pytest.mark.parametrize("case,interval,n,parts", [("WPS_hard", 20, 6, 2), ("WPS_escort", 12, 5, 3)])
def test_stepwise_bit_exact_vs_oracle_through_the_part_entry_points(case, interval, n, parts):
    """muavta_set_parts: the batch split into parts stepped on their own streams — allocate_part / step_part with host-side
    action rows (decided per part, the other parts' launches in flight), observe_part — every field of every env against the
    oracle after every step, exactly as the whole-batch stepwise test does."""
    env = _env(case, n)
    seeds = np.arange(40, 40 + n, dtype=np.uint64)
    env.reset(seeds)
    env.set_parts(parts)
    ranges = [env.part_range(p) for p in range(parts)]
    assert sum(c for _, c in ranges) == n and ranges[0][0] == 0
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    for o, s in zip(oracles, seeds):
        o.reset(int(s))
    for t in range(150):
        for p, (first, count) in enumerate(ranges):           # decide for part p while the parts before it are being stepped
            aa, ai = env.allocate_part(p, interval, True)
            for i in range(count):
                oa, oi = oracles[first + i].allocate(interval, 1)
                k = len(oa)
                assert np.array_equal(aa[i, :k], oa) and np.array_equal(ai[i, :k], oi) and (k == aa.shape[1] or aa[i, k] == -1), f"{case} t={t} env {first + i}: plan"
                oracles[first + i].step(oa, oi)
            env.step_part(p, aa, ai)
        obs_parts = [env.observe_part(p) for p in range(parts)]
        snap = Snapshot(env)                                    # whole-batch reads: ordered after every part's stream
        for p, (first, count) in enumerate(ranges):
            o_p, r_p, term_p, trunc_p = obs_parts[p]
            for key in ("tasks", "legal_mask", "mask", "agents", "event_flags"):
                assert np.array_equal(o_p[key], snap.obs[key][first:first + count]), f"{case} t={t} part {p}: observe_part {key}"
            assert np.array_equal(r_p, snap.reward[first:first + count]) and np.array_equal(trunc_p, snap.trunc[first:first + count])
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"{case} parts seed {seeds[i]} t={t + 1}")


@pytest.mark.parametrize("case,interval,n", [("WPS_hard_x2", 20, 1024), ("WPS_escort24", 12, 256)])
def test_rollout_by_parts_equals_the_whole_batch_rollout(case, interval, n):
    """150 x rollout_part(1 step) on four parts (the device-side planner, one launch per part and env step, all asynchronous) ends in
    the same state, observations and metrics as one fused 150-step rollout of the whole batch."""
    seeds = np.arange(7000, 7000 + n, dtype=np.uint64)
    ref = _env(case, n)
    ref.rollout(seeds, 150, interval, True, True)
    want_m, want_obs, want_state = ref.rollout_metrics(), ref.observe(), ref.get_state()
    env = _env(case, n)
    env.reset(seeds)
    env.set_parts(4)
    for t in range(150):
        for p in range(4):
            env.rollout_part(p, 1, interval, True, True)
    got_obs = env.observe()
    assert np.array_equal(env.metrics(), want_m) and not env.get("ERROR").any()
    for key in want_obs:
        assert np.array_equal(got_obs[key], want_obs[key]), key
    env.set_parts(0)
    assert np.array_equal(env.get_state(), want_state)


def test_compat_install_without_a_factory_resolves_to_the_hip_library():
    """`compat.install()` with no `backend_factory` (the product default) must build its envs on libmuavta.so — the injection hook
    is for this repository's CPU tests only.  A fresh interpreter: installs the aliases, imports the reference's module names,
    runs a short episode through the facade and reports which backend and which shared objects it ended up with."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import muavta_amd.compat as compat
compat.install()
from mUAV_TA.DroneEnv import MultiUAVEnv
from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions
env = MultiUAVEnv(agentEnvOptions(agents={"F1": 2, "R1": 2}, tasks={"Att": 2, "Rec": 2}, multiple_tasks_per_agent=True, max_time_steps=30))
obs, _ = env.reset(seed=3)
for _ in range(5):
    obs, rew, term, trunc, info = env.step({a: 0 for a in env.agents})
maps = open("/proc/self/maps").read()
print(json.dumps({"backend": type(env._b).__module__ + "." + type(env._b).__name__, "libmuavta": "libmuavta.so" in maps,
                  "oracle": "liboracle" in maps, "oracle_modules": [m for m in sys.modules if m in ("orc", "oracle_backend")], "t": env.time_steps}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "MUAVTA_SO"}
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["backend"].endswith("batched.BatchedMultiUAVEnv") and got["libmuavta"] and not got["oracle"] and not got["oracle_modules"] and got["t"] == 5, got


# ---- caller-supplied planner inputs: muavta_allocate_scored (a25 inputs, f3 RL half) -----------------------------------------------
SCORED_FILES = sorted(glob.glob(os.path.join(GOLDEN, "rl_*.npz")) + glob.glob(os.path.join(GOLDEN, "rah_*.npz")) + glob.glob(os.path.join(GOLDEN, "esc_*.npz")))
GATE = {"force": 0, "trainer": 1, "escort": 2, "allocator": 3}


def _scored_setup(path):
    """(case, kind name, kind id, MT, MA, interval, gate, kwargs of allocate_scored, oracle flags) of a reference-driven trace"""
    g = np.load(path)
    base = os.path.basename(path)
    if base.startswith("rl_"):
        raw = bool(int(g["raw"]))
        return g, base[3:-4], "pair_raw" if raw else "pair", 1 if raw else 0, 32, 16, 20, "trainer", dict(edge_valid_only=True), 1
    if base.startswith("rah_"):
        return g, base[4:-4], "pair", 0, 32, 16, 15, "trainer", dict(full_task_list=True), 2
    return g, base[4:-4], "escort", 2, int(g["max_tasks"]), int(g["max_agents"]), int(g["interval"]), "escort", dict(edge_valid_only=False, commit=True), 4


@pytest.mark.parametrize("path", SCORED_FILES, ids=[os.path.basename(p)[:-4] for p in SCORED_FILES])
def test_scored_allocator_reference_traces_and_oracle(path):
    """env 0 replays the reference's own episode (PairCostHybrid / AttentionRAH / AttentionEscort planner with seeded network
    outputs: tools/gen_golden.py --rl) — actions, _selected_mask, gate, S_WPS per step, final metrics; the other envs run
    other seeds with random scores / priorities / reserved sets against the oracle, every field of the state after every step."""
    g, case, kname, kind, mt, ma, interval, gate, kw, oflags = _scored_setup(path)
    n, seed0 = 4, int(g["seed"])
    # (a random-score policy churns escorts harder than any allocator of the registry: the reference episode of the 24-UAV escort
    # case holds more than the 24-agent tile's 88 pending reveals at t = 60, so it runs on the 64-agent tile)
    env = _env(case, n, **(dict(tile_agents=64, tile_tasks=128, tile_threats=48) if os.path.basename(path) == "esc_WPS_escort24.npz" else {}))
    A = env.n_agents
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    env.reset(np.arange(seed0, seed0 + n, dtype=np.uint64))
    for i, o in enumerate(oracles):
        o.reset(seed0 + i)
    rng = np.random.default_rng(5)
    steps, k = g["step"].tolist(), 0
    has_sc, has_pri = "scores" in g.files, "pri" in g.files
    for t in range(len(g["replanned"])):
        planned = bool(g["replanned"][t])
        sc = (rng.uniform(-1, 1, (n, ma, mt)) * (0.35 if kind != 2 else 1.0)).astype(np.float32) if has_sc else None
        pri = rng.uniform(0, 1, (n, mt)) if has_pri else None
        res = (rng.integers(0, 1 << A, n, dtype=np.uint64) & rng.integers(0, 1 << A, n, dtype=np.uint64)) if has_pri else None
        if planned:
            if has_sc:
                sc[0] = g["scores"][k]
            if has_pri:
                pri[0] = g["pri"][k]; res[0] = g["reserved"][k]
        elif has_pri:
            res[0] = 0
        out = env.allocate_scored(kname, mt, ma, edge_scores=sc, task_pri=pri, reserved=res, gate=gate, replan_interval=interval, **kw)
        aa, ai = out["act_agent"], out["act_index"]
        assert bool(out["replanned"][0]) == planned, f"{case} t={t}: gate"
        if planned:
            assert t == steps[k]
            if "selected" in g.files:
                assert np.array_equal(out["selected"][0], g["selected"][k]), f"{case} t={t}: selected mask vs reference"
            k += 1
        want = g["actions"][g["actions"][:, 0] == t][:, 1:]
        want = want[np.argsort(want[:, 0], kind="stable")]
        n0 = int((aa[0] >= 0).sum())
        order = np.argsort(aa[0][:n0], kind="stable")
        assert np.array_equal(np.stack([aa[0][:n0][order], ai[0][:n0][order]], axis=1).reshape(-1, 2), want.reshape(-1, 2)), f"{case} t={t}: actions vs reference"
        for i, o in enumerate(oracles):
            oa, oi, osel = o.allocate_scored(interval, 1, GATE[gate], kind, mt, ma, oflags, scores=None if sc is None else sc[i],
                                            pri=None if pri is None else pri[i], reserved=0 if res is None else int(res[i]))
            kk = len(oa)
            assert np.array_equal(aa[i][:kk], oa) and np.all(aa[i][kk:] == -1) and np.array_equal(ai[i][:kk], oi), f"{case} seed {seed0 + i} t={t}: {aa[i]} vs {oa}"
            assert np.array_equal(out["selected"][i], osel), f"{case} seed {seed0 + i} t={t}: selected mask"
            assert bool(out["replanned"][i]) == (o.scalars_last_plan() == t)
            o.step(oa, oi)
        env.step_staged()
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"scored {case} seed {seed0 + i} t={t + 1}")
        if "s_wps" in g.files:
            assert env.metrics()[0][4] == g["s_wps"][t + 1]
    assert k == len(steps) and np.array_equal(env.metrics()[0], g["metrics"])


def test_scored_allocator_device_tensors_equal_the_host_path():
    import torch

    case, n, mt, ma = "WPS_hard_x2", 64, 32, 16
    a, b = _env(case, n), _env(case, n)
    seeds = np.arange(n, dtype=np.uint64)
    a.reset(seeds); b.reset(seeds)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    sel = torch.empty((n, ma, mt), dtype=torch.float32, device=dev)
    rep = torch.empty((n,), dtype=torch.int32, device=dev)
    for t in range(60):
        sc = (rng.uniform(-1, 1, (n, ma, mt)) * 0.35).astype(np.float32)
        pri = rng.uniform(0, 1, (n, mt))
        res = rng.integers(0, 1 << 16, n, dtype=np.uint64) & rng.integers(0, 1 << 16, n, dtype=np.uint64) & rng.integers(0, 1 << 16, n, dtype=np.uint64)
        ha = a.allocate_scored("pair", mt, ma, edge_scores=sc, task_pri=pri, reserved=res, gate="trainer", replan_interval=10)
        b.allocate_scored("pair", mt, ma, edge_scores=torch.from_numpy(sc).to(dev), task_pri=torch.from_numpy(pri).to(dev),
                          reserved=torch.from_numpy(res.view(np.int64)).to(dev), gate="trainer", replan_interval=10, out={"selected": sel, "replanned": rep})
        b.sync()
        assert np.array_equal(sel.cpu().numpy(), ha["selected"]) and np.array_equal(rep.cpu().numpy(), ha["replanned"])
        assert np.array_equal(a.get("STAGED_ACTIONS"), b.get("STAGED_ACTIONS"))
        a.step_staged(); b.step_staged()
    assert np.array_equal(a.metrics(), b.metrics())


@pytest.mark.parametrize("gate,flags_kw,kname,kind,oflags", [("allocator", dict(edge_valid_only=True, full_task_list=True), "pair", 0, 3),
                                                             ("force", dict(edge_valid_only=False, commit=True), "escort", 2, 4),
                                                             ("escort", dict(edge_valid_only=True), "pair_raw", 1, 1)])
def test_scored_allocator_on_fuzzed_configs_vs_oracle(gate, flags_kw, kname, kind, oflags):
    """flag / gate / kind combinations no reference planner uses, on the fuzzed env configurations, with token pads SMALLER
    than the fleet and the open list (rows and columns beyond the pads carry no score / priority): device vs oracle"""
    import json
    from muavta_amd.params import params_from_config
    from cases import params_of

    cfgs = json.load(open(os.path.join(GOLDEN, "fuzz_configs.json")))
    rng = np.random.default_rng(23)
    for name in list(cfgs)[:8]:
        P = params_of(name)
        n, mt, ma = 3, 6, 3
        from muavta_amd.batched import BatchedMultiUAVEnv
        env = BatchedMultiUAVEnv(P, n)
        A = env.n_agents
        oracles = [orc.OracleEnv(P) for _ in range(n)]
        env.reset(np.arange(n, dtype=np.uint64))
        for i, o in enumerate(oracles):
            o.reset(i)
        for t in range(P.max_time_steps):
            sc = rng.uniform(-1, 1, (n, ma, mt)).astype(np.float32)
            pri = rng.uniform(-0.5, 1, (n, mt))
            res = rng.integers(0, 1 << A, n, dtype=np.uint64) & rng.integers(0, 1 << A, n, dtype=np.uint64)
            out = env.allocate_scored(kname, mt, ma, edge_scores=sc, task_pri=pri, reserved=res, gate=gate, replan_interval=7, use_visibility=bool(t % 2), **flags_kw)
            for i, o in enumerate(oracles):
                oa, oi, osel = o.allocate_scored(7, t % 2, GATE[gate], kind, mt, ma, oflags, scores=sc[i], pri=pri[i], reserved=int(res[i]))
                kk = len(oa)
                assert np.array_equal(out["act_agent"][i][:kk], oa) and np.all(out["act_agent"][i][kk:] == -1) and np.array_equal(out["act_index"][i][:kk], oi), f"{name} seed {i} t={t}"
                assert np.array_equal(out["selected"][i], osel), f"{name} seed {i} t={t}: selected"
                o.step(oa, oi)
            env.step_staged()
            snap = Snapshot(env)
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"scored-fuzz {name} seed {i} t={t + 1}")
            if snap.term[0] or snap.trunc[0]:
                break


RL_FILES = sorted(glob.glob(os.path.join(GOLDEN, "rl_*.npz")))


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "four-launches"])
@pytest.mark.parametrize("path", RL_FILES, ids=[os.path.basename(p)[3:-4] for p in RL_FILES])
def test_rl_stream_policy_in_the_loop_vs_reference_and_oracle(path, fused):
    """run_rl_episode batched (muavta_amd.il.rl_stream; muavta_rl_step_device when fused): env 0 replays the reference episode
    driven by PairCostHybrid.plan(scores=seeded) — tok, selected, replanned, step reward, next_tok, done — the other envs run
    other seeds against the oracle, with the policy a function of the token tensors on the GPU."""
    import torch
    from muavta_amd.il import rl_stream

    g = np.load(path)
    case = os.path.basename(path)[3:-4]
    raw = bool(int(g["raw"]))
    kname, kind = ("pair_raw", 1) if raw else ("pair", 0)
    n, seed0 = 5, int(g["seed"])
    env = _env(case, n)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(seed0 + i)
    steps = g["step"].tolist()
    state = {"k": 0, "t": 0}
    ref_scores = torch.from_numpy(g["scores"]).cuda()

    def policy(tok):
        # a deterministic function of the tokens (so the loop really closes on the device) ...
        sc = torch.tanh(tok["agent_feats"][:, :, :1] * 3.0 - tok["task_feats"][:, :, 0].unsqueeze(1) * 2.0 + tok["task_feats"][:, :, 1].unsqueeze(1)) * 0.35
        sc = sc.contiguous()
        if state["k"] < len(steps) and steps[state["k"]] == state["t"]:  # ... and for env 0 the matrix the reference was given
            sc[0] = ref_scores[state["k"]]
        return sc

    pend = None
    for t, tr in rl_stream(env, np.arange(seed0, seed0 + n), policy, n_steps=len(g["replanned"]), interval=20, kind=kname, fused=fused):
        sc = tr["scores"].cpu().numpy()
        sel, rep, rew, dn = tr["selected"].cpu().numpy(), tr["replanned"].cpu().numpy(), tr["step_reward"].cpu().numpy(), tr["done"].cpu().numpy()
        tok = {k_: v.cpu().numpy() for k_, v in tr["tok"].items()}
        nxt = {k_: v.cpu().numpy() for k_, v in tr["next_tok"].items()}
        k = state["k"]
        assert int(rep[0]) == int(g["replanned"][t]), f"{case} t={t}: gate"
        if g["replanned"][t]:
            assert t == steps[k]
            assert np.array_equal(tok["task_feats"][0], g["tf"][k]) and np.array_equal(tok["agent_feats"][0], g["af"][k]) and np.array_equal(tok["edge_valid"][0], g["ev"][k])
            assert np.array_equal(tok["task_ids"][0], g["tid"][k]) and np.array_equal(tok["agent_ids"][0], g["aid"][k])
            assert np.array_equal(sel[0], g["selected"][k]), f"{case} t={t}: selected"
            assert rew[0] == g["step_r"][k] and bool(dn[0]) == bool(g["ep_done"][k])
            assert np.array_equal(nxt["task_feats"][0], g["ntf"][k]) and np.array_equal(nxt["agent_feats"][0], g["naf"][k]) and np.array_equal(nxt["task_ids"][0], g["ntid"][k])
            state["k"] += 1
        else:
            assert not sel[0].any()
        assert rew[0] == (g["s_wps"][t + 1] - g["s_wps"][t]) / 20.0
        for i, o in enumerate(oracles):
            want_tok = o.tokens(kind, 32, 16)
            for key in ("task_feats", "agent_feats", "edge_valid", "task_ids", "agent_ids", "task_mask", "agent_mask"):
                assert np.array_equal(tok[key][i], want_tok[key]), f"{case} seed {seed0 + i} t={t}: tok {key}"
            assert int(tok["n_urgent"][i]) == want_tok["n_urgent"]
            before = o.metrics()[4]
            oa, oi, osel = o.allocate_scored(20, 1, 1, kind, 32, 16, 1, scores=sc[i])
            assert np.array_equal(sel[i], osel) and bool(rep[i]) == (o.scalars_last_plan() == t), f"{case} seed {seed0 + i} t={t}"
            d = o.step(oa, oi)
            assert rew[i] == (o.metrics()[4] - before) / 20.0 and bool(dn[i]) == bool(d)
            want_nxt = o.tokens(kind, 32, 16)
            for key in ("task_feats", "agent_feats", "edge_valid", "task_ids", "agent_ids", "task_mask", "agent_mask"):
                assert np.array_equal(nxt[key][i], want_nxt[key]), f"{case} seed {seed0 + i} t={t}: next_tok {key}"
        state["t"] = t + 1
    assert state["k"] == len(steps) and np.array_equal(env.metrics()[0], g["metrics"])
    env.refresh_observation()  # (the fused step ran without the observation write, which is what rebuilds initTime / doneTime)
    snap = Snapshot(env)
    for i, o in enumerate(oracles):
        compare(snap, i, o, f"rl_stream {case} seed {seed0 + i} final", check_obs=False)


def test_capacity_flagged_envs_escalate_to_the_next_tile():
    """rollout(escalate=True): the reference's task list is unbounded (DroneEnv.py:325-328); an env that needs more than its
    tile is re-run from its seed on the next larger tile and its metrics spliced in.  Forced here with muavta_set_slot_cap(32)
    on the 16-agent tile (at most 32 of its 40 slots): global index 9649 of WPS_hard_x2 needs 34."""
    case = "WPS_hard_x2"
    seeds = np.array([9649, 6231, 8273, 11430, 11656] + list(range(251)), dtype=np.uint64)
    n = len(seeds)
    small = _env(case, n)
    small.set_slot_cap(32)
    assert small.T == 40 and small.A_tile == 16
    small.rollout(seeds, 150, 20, True, True, escalate=True)
    flagged = np.nonzero(small.get("ERROR"))[0]
    assert 0 in flagged, "seed 9649 needs 34 live task slots: 32 were expected to overflow"
    assert sorted(small.escalated) == flagged.tolist()
    assert all(h.A_tile == 24 for h, _ in small.escalated.values())
    got = small.rollout_metrics()
    want = orc.parallel_metrics(case, seeds, 20)
    assert np.array_equal(got, want), f"rows differing: {np.nonzero(~np.all(got == want, axis=1))[0][:8]} (flagged: {flagged[:8]})"
    # without escalation the same batch reports the overflow instead of a result
    from muavta_amd.native import MuavtaError
    small.rollout(seeds, 150, 20, True, True)
    with pytest.raises(MuavtaError):
        small.metrics()
    # the allocator mode travels with the escalation
    small.set_allocator("urgency_pair")
    small.rollout(seeds, 150, 20, True, False, escalate=True)
    assert len(small.escalated) > 0
    want2 = orc.parallel_metrics(case, seeds, 20, 1, 1)
    assert np.array_equal(small.rollout_metrics(), want2)
    # two rungs: a cap the 24-agent tile cannot hold either is not available by request (its 48 slots always cover 34), so the
    # ladder's second rung is exercised with the fuzzed configs' 128-slot requests elsewhere; here: the final state of an escalated env
    h, k = small.escalated[0]
    assert int(h.get("ERROR")[k]) == 0 and h.T == 48


def _scored_soak_worker(args):
    """oracle side of test_scored_allocator_soak: the same per-(seed, step) pseudo-random planner inputs, one env at a time"""
    case, seeds, n_steps, kind, mt, ma, gate, oflags, interval = args
    import numpy as _np
    o = orc.OracleEnv(params_for_case(case))
    out = _np.zeros((len(seeds), 30))
    for i, sd in enumerate(seeds):
        o.reset(int(sd))
        for t in range(n_steps):
            sc, pri, res = _scored_soak_inputs(int(sd), t, mt, ma, o.A)
            oa, oi, _ = o.allocate_scored(interval, 1, gate, kind, mt, ma, oflags, scores=sc, pri=pri, reserved=res)
            if o.step(oa, oi):
                break
        out[i] = o.metrics()
    return out


def _scored_soak_inputs(seed, t, mt, ma, n_agents):
    rng = np.random.default_rng([seed, t, 77])
    sc = (rng.uniform(-1, 1, (ma, mt)) * 0.35).astype(np.float32)
    pri = rng.uniform(0, 1, mt)
    bits = rng.integers(0, 8, n_agents) == 0  # each agent reserved with probability 1/8
    res = int(sum(1 << int(a) for a in np.nonzero(bits)[0]))
    return sc, pri, res


@pytest.mark.parametrize("case,kname,kind,mt,ma,gate,kw,oflags,interval,n", [
    ("WPS_hard_x2", "pair", 0, 32, 16, "trainer", dict(edge_valid_only=True), 1, 20, 384),
    ("WPS_burst64", "pair", 0, 32, 16, "allocator", dict(edge_valid_only=True, full_task_list=True), 3, 20, 96),
    ("WPS_escort", "escort", 2, 32, 16, "escort", dict(edge_valid_only=False, commit=True), 4, 12, 128)])
def test_scored_allocator_soak(case, kname, kind, mt, ma, gate, kw, oflags, interval, n):
    """whole episodes of many seeds with per-step pseudo-random scores + priorities + reserved agents: final metrics vs the oracle"""
    import multiprocessing as mp

    env = _env(case, n)
    seeds = np.arange(5000, 5000 + n, dtype=np.uint64)
    env.reset(seeds)
    A = env.n_agents
    for t in range(150):
        ins = [_scored_soak_inputs(int(sd), t, mt, ma, A) for sd in seeds]
        sc = np.stack([x[0] for x in ins]); pri = np.stack([x[1] for x in ins]); res = np.array([x[2] for x in ins], dtype=np.uint64)
        env.allocate_scored(kname, mt, ma, edge_scores=sc, task_pri=pri, reserved=res, gate=gate, replan_interval=interval, want_selected=False, fetch=False, **kw)
        env.step_staged()
    assert not env.get("ERROR").any()
    got = env.metrics()
    orc.lib()
    procs = 8
    chunk = (n + procs - 1) // procs
    jobs = [(case, seeds[i:i + chunk].tolist(), 150, kind, mt, ma, GATE[gate], oflags, interval) for i in range(0, n, chunk)]
    with mp.get_context("spawn").Pool(procs) as pool:
        want = np.concatenate(pool.map(_scored_soak_worker, jobs), axis=0)
    assert np.array_equal(got, want), f"{case}: rows differing {np.nonzero(~np.all(got == want, axis=1))[0][:8]}"


@pytest.mark.parametrize("case,interval,n,mode", [("WPS_hard_x2", 20, 256, "hungarian"), ("WPS_escort24", 12, 128, "hungarian"), ("WPS_burst64", 20, 64, "hungarian"),
                                                  ("WPS_escort", 12, 128, "urgency_coalition"), ("FUZZ03", 20, 64, "hungarian"), ("FUZZ07", 20, 64, "hungarian"),
                                                  ("FUZZ11", 12, 64, "hungarian")])
def test_incremental_observation_rows_equal_a_full_rewrite(case, interval, n, mode):
    """(r4) The observation writer only rewrites the per-step columns of rows whose static columns the handle's buffer already
    holds (OBS_STATIC).  Two identical batches stepped in lockstep: A reads the incrementally maintained buffer, B has it rewritten
    in full before every read (muavta_refresh_observation) — every tensor must be equal after every step."""
    from cases import params_of
    from muavta_amd.batched import BatchedMultiUAVEnv

    P = params_of(case)
    a, b = BatchedMultiUAVEnv(P, n), BatchedMultiUAVEnv(P, n)
    a.set_allocator(mode); b.set_allocator(mode)
    seeds = np.arange(300, 300 + n, dtype=np.uint64)
    a.reset(seeds); b.reset(seeds)
    n_light = 0
    for t in range(P.max_time_steps):
        for e in (a, b):
            e.allocate(interval, True, fetch=False)
            e.step_staged()
        n_light += int(np.count_nonzero(a.get("SCALARS")[:, 0] >= 0))  # (keeps the host mirror honest: a get between steps must not disturb the flag)
        b.refresh_observation()
        oa, ob = a.observe(), b.observe()
        for name in oa:
            x, y = np.ascontiguousarray(oa[name]), np.ascontiguousarray(ob[name])
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"{case} t={t + 1}: {name} differs in envs {np.nonzero((x != y).reshape(n, -1).any(axis=1))[0][:6]}"
        _, term, trunc = a.step_result()
        if (term | trunc).all():
            break
    assert np.array_equal(a.metrics(), b.metrics())


@pytest.mark.parametrize("handles", [1, 2])
def test_in_flight_rollouts_equal_one_handle_at_a_time(handles):
    """muavta_amd.pipeline.InFlightRollouts: batches in flight on the two state lanes of one handle (and of two handles) give the
    batches a single one-lane handle gives, launch by launch"""
    from muavta_amd.pipeline import InFlightRollouts

    case, n = "WPS_escort24", 256
    batches = [np.arange(b * n, (b + 1) * n, dtype=np.uint64) for b in range(7)]
    pipe = InFlightRollouts(params_for_case(case), n, handles=handles)
    got = pipe.run(batches, 150, 12)
    assert all(e.lanes() == (2, 2) for e in pipe.envs)
    pipe.close()
    one = _env(case, n)
    one.set_lanes(1)
    for b, seeds in enumerate(batches):
        one.rollout(seeds, 150, 12, True, True)
        assert np.array_equal(got[b], one.rollout_metrics()), f"batch {b}"
    assert one.lanes() == (1, 1)
    assert np.array_equal(got[0], orc.parallel_metrics(case, batches[0], 12))


@pytest.mark.parametrize("case,interval,n", [("WPS_hard_x2", 20, 2048), ("WPS_escort24", 12, 1024), ("WPS_burst64", 20, 256)])
def test_state_lanes_of_one_handle_every_batch_bit_equal_and_every_entry_point_follows_the_latest_batch(case, interval, n):
    """muavta_set_lanes (default: the second lane appears when a seeded rollout is queued while the previous one still runs): six
    batches queued back to back on ONE handle, each read through rollout_metrics(back=1) while the next one runs — all 30 metrics
    of every env of every batch equal the oracle's; afterwards the handle's state, observation, per-step entry points and the
    launch-time history all refer to the LATEST batch, exactly as on one lane; a handle that synchronises between rollouts never
    allocates the second lane."""
    env = _env(case, n)
    assert env.lanes() == (0, 1)
    batches = [np.arange(1000 * b, 1000 * b + n, dtype=np.uint64) for b in range(6)]
    for seeds in batches[:3]:  # default mode, nothing read in between: the second launch finds the first still running
        env.rollout(seeds, 150, interval, True, True)
    assert env.lanes() == (0, 2), "launches queued back to back were expected to find the previous one still running"
    assert np.array_equal(env.rollout_metrics(), orc.parallel_metrics(case, batches[2], interval))
    env.set_lanes(2)  # always alternate: the pipeline `launch batch b; read batch b - 1` may rely on reach-back
    got = []
    for b, seeds in enumerate(batches):
        env.rollout(seeds, 150, interval, True, True)
        if b:
            got.append(env.rollout_metrics(back=1))  # batch b - 1, on the other lane, while batch b runs
            assert not np.count_nonzero(env.error_flags(back=1))
    got.append(env.rollout_metrics())
    for b, seeds in enumerate(batches):
        assert np.array_equal(got[b], orc.parallel_metrics(case, seeds, interval)), f"{case} batch {b}"
    ms = env.kernel_ms_history(6)
    assert ms.shape == (6,) and np.all(ms > 0)
    # everything follows the latest batch: continue it stepwise against a one-lane handle that ran only that batch
    ref = _env(case, n)
    ref.set_lanes(1)
    ref.rollout(batches[-1], 150, interval, True, True)
    assert np.array_equal(env.metrics(), ref.metrics())
    for name in ("AGENT_POS", "TASK_ID", "TASK_STATUS", "SCALARS", "OPEN_IDS", "KNOWN"):
        assert np.array_equal(env.get(name), ref.get(name)), name
    oa, ob = env.observe(), ref.observe()
    assert all(np.array_equal(oa[k], ob[k]) for k in oa)
    # a fresh episode, 40 fused steps, then per-step calls: on whichever lane the seeded rollout landed
    env.rollout(batches[0], 40, interval, True, True); ref.rollout(batches[0], 40, interval, True, True)
    for _ in range(5):
        aa, ai = env.allocate(interval, True); ba, bi = ref.allocate(interval, True)
        assert np.array_equal(aa, ba) and np.array_equal(ai, bi)
        env.step(aa, ai); ref.step(ba, bi)
    env.rollout(None, 105, interval, True, True); ref.rollout(None, 105, interval, True, True)
    assert np.array_equal(env.rollout_metrics(), ref.rollout_metrics()) and np.array_equal(env.rollout_metrics(), got[0])
    from muavta_amd.native import MuavtaError
    with pytest.raises(MuavtaError):  # the launch before the latest one (the 40-step one) ran on this same lane: gone, and the call says so
        env.rollout_metrics(back=1)
    # a caller that synchronises between its rollouts stays on one lane
    lone = _env(case, 64)
    for b in range(3):
        lone.rollout(batches[b][:64], 150, interval, True, True)
        lone.sync()
    assert lone.lanes() == (0, 1)
    # back to one lane: the second one is released, results unchanged
    env.set_lanes(1)
    assert env.lanes() == (1, 1)
    env.rollout(batches[1], 150, interval, True, True)
    assert np.array_equal(env.rollout_metrics(), got[1])


def test_rl_step_by_sub_batches_equals_the_whole_batch_launch():
    """MuavtaRlStep.part: the fused RL step launched per sub-batch on the parts' own streams (whole-batch tensors, the part's rows)
    leaves the same tensors and the same env state as one launch over the whole batch."""
    import torch

    case, n, mt, ma = "WPS_hard_x2", 96, 32, 16
    a, b = _env(case, n), _env(case, n)
    seeds = np.arange(40, 40 + n, dtype=np.uint64)
    a.reset(seeds); b.reset(seeds)
    b.set_parts(3)
    dev = torch.device("cuda", 0)
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}

    def outs(env):
        o = {"next_tok": {k: torch.zeros(sh, dtype=tdt[dt], device=dev) for k, (sh, dt) in env.token_shapes("pair", mt, ma).items()},
             "selected": torch.zeros((n, ma, mt), dtype=torch.float32, device=dev), "replanned": torch.zeros((n,), dtype=torch.int32, device=dev),
             "s_wps": torch.zeros((2, n), dtype=torch.float64, device=dev), "done": torch.zeros((n,), dtype=torch.uint8, device=dev)}
        return o

    oa, ob = outs(a), outs(b)
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    for t in range(60):
        sc = ((torch.rand((n, ma, mt), generator=gen, device=dev) * 2 - 1) * 0.35).contiguous()
        torch.cuda.synchronize()
        a.rl_step("pair", mt, ma, edge_scores=sc, gate="trainer", replan_interval=10, **oa)
        for p in range(3):
            b.rl_step("pair", mt, ma, edge_scores=sc, gate="trainer", replan_interval=10, part=p, **ob)
        a.sync(); b.wait_part(-1)
        for k in ("selected", "replanned", "s_wps", "done"):
            assert torch.equal(oa[k], ob[k]), f"t={t}: {k}"
        for k in oa["next_tok"]:
            assert torch.equal(oa["next_tok"][k], ob["next_tok"][k]), f"t={t}: next_tok {k}"
    b.set_parts(0)
    assert np.array_equal(a.metrics(), b.metrics())
    assert np.array_equal(a.get("AGENT_POS"), b.get("AGENT_POS"))


# ---- run to the next replan gate: muavta_rl_run_device / muavta_step_run ----------------------------------------------------------------
@pytest.mark.parametrize("max_steps", [0, 4], ids=["to-the-gate", "at-most-4-steps"])
@pytest.mark.parametrize("path", RL_FILES, ids=[os.path.basename(p)[3:-4] for p in RL_FILES])
def test_rl_run_ahead_policy_consulted_at_gates_vs_reference_and_oracle(path, max_steps):
    """run_rl_episode with the policy consulted only when _should_replan fires (experiments/train_pair_cost.py:139-145): muavta_amd.il.rl_run_stream /
    muavta_rl_run_device.  Env 0 replays the reference episode launch by launch — one launch per gate of the reference (tests/golden/rl_*.npz hold
    every gate step): tok, selected, step reward, next_tok, ep_done, and the quiet stretch ends at the reference's next gate.  The other envs run
    other seeds (their clocks drift apart: every env advances to ITS next gate) against the oracle's run-to-the-gate, the policy a function of the
    token tensors on the GPU; final state of every env vs the oracle, field by field."""
    import torch
    from muavta_amd.il import rl_run_stream

    g = np.load(path)
    case = os.path.basename(path)[3:-4]
    raw = bool(int(g["raw"]))
    kname, kind = ("pair_raw", 1) if raw else ("pair", 0)
    n, seed0 = 6, int(g["seed"])
    env = _env(case, n)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(seed0 + i)
    steps, T = g["step"].tolist(), len(g["replanned"])
    state = {"k": 0, "t0": 0}
    ref_scores = torch.from_numpy(g["scores"]).cuda()

    def policy(tok):
        sc = torch.tanh(tok["agent_feats"][:, :, :1] * 3.0 - tok["task_feats"][:, :, 0].unsqueeze(1) * 2.0 + tok["task_feats"][:, :, 1].unsqueeze(1)) * 0.35
        sc = sc.contiguous()
        if state["k"] < len(steps) and steps[state["k"]] == state["t0"]:  # env 0 is parked at the reference's gate k: the matrix the reference was given
            sc[0] = ref_scores[state["k"]]
        return sc

    launches = 0
    tot_steps = np.zeros(n, dtype=np.int64)
    for kk, tr in rl_run_stream(env, np.arange(seed0, seed0 + n), policy, interval=20, kind=kname, max_steps=max_steps):
        launches += 1
        sc = tr["scores"].cpu().numpy()
        sel, rep, rew, dn = tr["selected"].cpu().numpy(), tr["replanned"].cpu().numpy(), tr["step_reward"].cpu().numpy(), tr["done"].cpu().numpy()
        nst, prk, rsum = tr["n_stepped"].cpu().numpy(), tr["park"].cpu().numpy(), tr["reward_sum"].cpu().numpy()
        tok = {k_: v.cpu().numpy() for k_, v in tr["tok"].items()}
        nxt = {k_: v.cpu().numpy() for k_, v in tr["next_tok"].items()}
        ptk = {k_: v.cpu().numpy() for k_, v in tr["park_tok"].items()}
        # env 0 vs the reference episode
        k, t0 = state["k"], state["t0"]
        if t0 < T:
            at = k < len(steps) and steps[k] == t0
            assert bool(rep[0]) == at, f"{case} t={t0}: gate"
            if at:
                assert np.array_equal(tok["task_feats"][0], g["tf"][k]) and np.array_equal(tok["agent_feats"][0], g["af"][k]) and np.array_equal(tok["edge_valid"][0], g["ev"][k])
                assert np.array_equal(tok["task_ids"][0], g["tid"][k]) and np.array_equal(tok["agent_ids"][0], g["aid"][k])
                assert np.array_equal(sel[0], g["selected"][k]), f"{case} t={t0}: selected"
                assert rew[0] == g["step_r"][k] and bool(dn[0]) == bool(g["ep_done"][k])
                assert np.array_equal(nxt["task_feats"][0], g["ntf"][k]) and np.array_equal(nxt["agent_feats"][0], g["naf"][k]) and np.array_equal(nxt["task_ids"][0], g["ntid"][k])
                state["k"] = k = k + 1
            else:
                assert not sel[0].any()
            assert rew[0] == (g["s_wps"][t0 + 1] - g["s_wps"][t0]) / 20.0
            nxt_gate = steps[k] if k < len(steps) else T
            assert int(nst[0]) == (nxt_gate - t0 if max_steps == 0 else min(nxt_gate - t0, max_steps)), f"{case} t={t0}: steps taken, reference's next gate at {nxt_gate}"
            state["t0"] = t0 + int(nst[0])
        else:
            assert int(nst[0]) == 0 and not rep[0]
        # every env vs the oracle's run-to-the-gate with the same scores
        for i, o in enumerate(oracles):
            r = o.rl_run(20, 1, 1, kind, 32, 16, 1, scores=sc[i], max_steps=max_steps)
            tag = f"{case} seed {seed0 + i} launch {kk}"
            assert int(nst[i]) == r["n_stepped"] and int(prk[i]) == r["park"] and bool(rep[i]) == r["replanned"], f"{tag}: {nst[i]} {prk[i]} {rep[i]} vs {r['n_stepped']} {r['park']} {r['replanned']}"
            assert np.array_equal(sel[i], r["selected"]), f"{tag}: selected"
            assert rew[i] == (r["s_after"] - r["s_before"]) / 20.0 and int(dn[i]) == r["done"], f"{tag}: step reward / done"
            assert rsum[i] == r["reward_sum"], f"{tag}: reward sum"
            if r["replanned"]:
                for key in ("task_feats", "agent_feats", "edge_valid", "task_ids", "agent_ids", "task_mask", "agent_mask"):
                    assert np.array_equal(nxt[key][i], r["next_tok"][key]), f"{tag}: next_tok {key}"
                assert int(nxt["n_urgent"][i]) == r["next_tok"]["n_urgent"]
            for key in ("task_feats", "agent_feats", "edge_valid", "task_ids", "agent_ids", "task_mask", "agent_mask"):
                assert np.array_equal(ptk[key][i], r["park_tok"][key]), f"{tag}: park_tok {key}"
            assert int(ptk["n_urgent"][i]) == r["park_tok"]["n_urgent"]
        tot_steps += nst
    assert state["k"] == len(steps) and np.all(tot_steps == T) and np.array_equal(env.metrics()[0], g["metrics"])
    if max_steps == 0:
        assert launches >= len(steps)  # (the batch needs as many launches as its env with the most gates)
    env.refresh_observation()
    snap = Snapshot(env)
    for i, o in enumerate(oracles):
        compare(snap, i, o, f"rl_run_stream {case} seed {seed0 + i} final", check_obs=False)


@pytest.mark.parametrize("case,interval,gate,n", [("WPS_hard", 20, "trainer", 6), ("WPS_escort", 12, "escort", 5), ("WPS_hard_x2", 20, "allocator", 6), ("WPS_burst64", 15, "trainer", 2)])
def test_step_run_host_planner_runs_ahead_to_its_gate_vs_oracle(case, interval, gate, n):
    """muavta_step_run: env.step(actions of a host-side plan) + env.step({}) up to the env's next gate in ONE launch (the loop of
    experiments/wps_eval.py:248-254,273).  The plan is the device allocator's, fetched to the host and handed back as action rows (odd
    launches) or left staged (even launches); every field of every env + the observation of the state it stopped in, after every launch."""
    G = {"force": 0, "trainer": 1, "escort": 2, "allocator": 3}[gate]
    env = _env(case, n)
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    env.reset(np.arange(n, dtype=np.uint64))
    for i, o in enumerate(oracles):
        o.reset(i)
    launches, total = 0, np.zeros(n, dtype=np.int64)
    while True:
        cap = 0 if launches % 3 else 5
        aa, ai = env.allocate(interval, True)
        for i, o in enumerate(oracles):
            oa, oi = o.allocate(interval, 1)
            k = len(oa)
            assert np.array_equal(aa[i][:k], oa) and np.all(aa[i][k:] == -1) and np.array_equal(ai[i][:k], oi), f"{case} seed {i} launch {launches}: plan"
        if launches % 2:
            nst, prk, rs = env.step_run(aa, ai, gate=gate, replan_interval=interval, max_steps=cap)
        else:
            nst, prk, rs = env.step_run(None, None, gate=gate, replan_interval=interval, max_steps=cap)
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            d = o.dims()
            if d["terminated"] or d["truncated"]:
                assert nst[i] == 0 and (prk[i] & 3)
                continue
            # (the device's plan equals the oracle's, checked above: replay it through step, then the quiet stretch)
            pa, pi = aa[i][aa[i] >= 0], ai[i][:int((aa[i] >= 0).sum())]
            o.step(pa, pi)
            r0 = float(o.scalars()[1])
            q, ag, rq = o.run_quiet(G, interval, cap, 1, r0)
            dd = o.dims()
            want_park = int(dd["terminated"]) | (int(dd["truncated"]) << 1) | (4 if ag else 0)
            assert int(nst[i]) == 1 + q and int(prk[i]) == want_park and rs[i] == rq, f"{case} seed {i} launch {launches}: {nst[i]} {prk[i]} {rs[i]} vs {1 + q} {want_park} {rq}"
            compare(snap, i, o, f"step_run {case} seed {i} launch {launches}")
        total += nst
        launches += 1
        if np.all(prk & 3):
            break
        assert launches < 400
    m = env.metrics()
    for i, o in enumerate(oracles):
        assert np.array_equal(m[i], o.metrics()), f"{case} seed {i}: final metrics"
    assert launches < int(total.max())  # fewer launches than env steps: that is the point


def test_device_fuzz_fresh_slice():
    """A slice of the wide device fuzz (tests/fuzz_device.py: the fused rollout on every tile that holds the fleet + all twelve legs) on
    configurations NO earlier run has seen: the first id of each family (small / large-fleet / EDGE of tests/fuzz_reference.py::wide_config)
    comes from the committed counter file tests/fuzz_counter.json, which the builder bumps every round past everything its own batches
    covered.  Round 4's seven device bugs were all invisible to the case-based tests and found by this driver, and round 5's one (the last bit of atan2) too; ~240 s of it — ≈300 configurations — now run wherever
    `pytest -m gpu` runs (MUAVTA_FUZZ_SLICE_SECONDS overrides the budget).  A mismatch fails the test with the configuration id and leg."""
    import json
    import time

    import fuzz_device as FD
    from fuzz_reference import wide_config

    budget = float(os.environ.get("MUAVTA_FUZZ_SLICE_SECONDS", "240"))
    counter = json.load(open(os.path.join(os.path.dirname(__file__), "fuzz_counter.json")))
    fams = [(f, int(counter[f])) for f in ("small", "large", "edge")]
    msgs, notes, tot, covered = [], [], {}, {f: 0 for f, _ in fams}
    t0, i = time.time(), 0
    while time.time() - t0 < budget:
        fam, base = fams[i % len(fams)]
        k = base + i // len(fams)
        w = wide_config(k)
        log = lambda m, k=k, fam=fam: (notes if " note: " in m else msgs).append(f"[{fam} k={k}] {m}")  # noqa: E731  ("note": an env beyond the LARGEST tile, flagged and skipped)
        try:
            b, f, c = FD.fused(k, w, 3, log)
            tot["fused_bad"] = tot.get("fused_bad", 0) + b; tot["fused_flagged"] = tot.get("fused_flagged", 0) + f; tot["fused_checked"] = tot.get("fused_checked", 0) + c
            for leg in FD.LEGS:
                key = f"{leg}_{getattr(FD, leg)(k, w, log)}"
                tot[key] = tot.get(key, 0) + 1
        except Exception as exc:  # (a capacity flag met by a call that refuses flagged batches; a configuration muavta_create rejects)
            if "overflowed a tile" in str(exc):
                tot["capacity_exceptions"] = tot.get("capacity_exceptions", 0) + 1
            else:
                msgs.append(f"[{fam} k={k}] ERROR {type(exc).__name__}: {str(exc)[:300]}")
        covered[fam] += 1
        i += 1
    summary = {"round": counter.get("round"), "seconds": round(time.time() - t0, 1), "configs": i,
               "ranges": {f: [b, b + covered[f] - 1] for f, b in fams if covered[f]}, "totals": tot, "mismatches": msgs[:20], "capacity_notes": len(notes)}
    print("device fuzz, fresh slice:", json.dumps(summary))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(summary, open(os.path.join(out_dir, "fuzz_fresh_slice.json"), "w"), indent=1)
    assert i >= 3, "the slice ran fewer than one configuration per family"
    assert not msgs and not tot.get("fused_bad"), msgs[:5]


def test_n_ranks_with_unequal_shards_on_the_one_gpu_random_configurations():
    """The N > 1 path with the PRODUCT on hardware: three ranks (spawned processes, one handle each on this box's one GPU, the collective over
    gloo: RCCL wants a GPU per rank), shard sizes drawn per configuration (unequal, some empty), random wide_config draws on the tile that
    holds their fleet — dist.reduce_metrics on every rank = the rank-ordered reduction of the ORACLE's metrics of the same global indices."""
    from test_dist_cpu import _run_ranks, check_rank_results, fuzz_jobs

    rng = np.random.default_rng(78)
    world = 3
    jobs = fuzz_jobs(51000, 6, world, rng, max_agents=24) + fuzz_jobs(1011000, 2, world, rng, max_agents=64)
    check_rank_results(_run_ranks(world, "hip", jobs, timeout=900), world, jobs, truth_backend="oracle")


@pytest.mark.parametrize("case,n,interval", [("WPS_hard", 5, 20), ("WPS_escort", 4, 12)])
def test_vectorised_facade_views_equal_single_env_facades(case, n, interval):
    """MultiUAVEnv.batch(n): n reference-shaped env objects over ONE handle (one launch, one state mirror and one observation copy per step for
    all of them) against n single-env facades stepped side by side with the same actions: observation dicts, rewards, done flags, infos
    (events, selected, final metrics), last_tasks_info, the Task / UAV views a planner reads and agent_visibility_map(), after every step."""
    from muavta_amd.env import MultiUAVEnv
    from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS

    ta, tt, th = TILES[case]
    kw = dict(flags=dict(WPS_ENV_FLAGS), tile_agents=ta, tile_tasks=tt, tile_threats=th)
    batch = MultiUAVEnv.batch(CASE_SPECS[case], n, **kw)
    singles = [MultiUAVEnv(CASE_SPECS[case], **kw) for _ in range(n)]
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
    for t in range(150):
        acts = []
        for i, (v, s) in enumerate(zip(batch.envs, singles)):
            aa, ai = v._b.allocate(interval, True)          # the batch: ONE k_allocate launch for all views, fetched once
            sa, si = s._b.allocate(interval, True)
            assert np.array_equal(aa, sa) and np.array_equal(ai, si)
            acts.append({v.agents_obj[int(a)].name: int(j) for a, j in zip(aa[0], ai[0]) if a >= 0})
        outs = batch.step(acts)
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
                    assert x.status == y.status and np.array_equal(x.position, y.position) and np.array_equal(x.currentReqs, y.currentReqs) and len(x.allocationDetails) == len(y.allocationDetails)
            if all(tr.values()):
                assert binfo["metrics"] == info["metrics"]
    assert all(all(o[3].values()) for o in outs)
    batch.close()


CONTEXT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "context_*.npz")))


@pytest.mark.parametrize("path", CONTEXT_FILES, ids=[os.path.basename(p)[8:-4] for p in CONTEXT_FILES])
def test_context_vector_vs_reference_and_oracle(path):
    """muavta_context: build_context_summary of the ContextPair hybrids for every env (ContextPairHybrid.py:33-78) — env 0 against the
    reference's own values along its episode (tests/golden/context_*.npz), the other seeds against the oracle; host and device-tensor variants."""
    import torch

    g = np.load(path)
    case = os.path.basename(path)[8:-4]
    seed0, interval, n = int(g["seed"]), int(g["interval"]), 4
    env = _env(case, n)
    env.reset(np.arange(seed0, seed0 + n, dtype=np.uint64))
    oracles = [orc.OracleEnv(params_for_case(case)) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(seed0 + i)
    steps = g["step"].tolist()
    dev_out = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    for t in range(150):
        aa, ai = env.allocate(interval, True)
        for o in oracles:
            o.allocate(interval, 1)
        if t in steps:
            k = steps.index(t)
            c32, c12, craw = env.context("pair", 32), env.context("pair", 12), env.context("pair_raw", 32)
            assert np.array_equal(c32[0], g["ctx"][k]) and np.array_equal(c12[0], g["ctx12"][k]) and np.array_equal(craw[0], g["ctx_raw"][k]), f"{case} t={t} (reference)"
            for i, o in enumerate(oracles):
                assert np.array_equal(c32[i], o.context(0, 32)) and np.array_equal(c12[i], o.context(0, 12)) and np.array_equal(craw[i], o.context(1, 32)), f"{case} seed {seed0 + i} t={t}"
            env.context("pair", 32, out=dev_out); env.sync()
            assert np.array_equal(dev_out.cpu().numpy(), c32)
        env.step(aa, ai)
        for i, o in enumerate(oracles):
            kk = int(np.sum(aa[i] >= 0))
            o.step(aa[i][:kk], ai[i][:kk])


@pytest.mark.skipif(os.environ.get("MUAVTA_UNVERIFIED_GPU_TESTS") != "1",
                    reason="written after round 5's GPU minutes were spent and never run on a device: MUAVTA_UNVERIFIED_GPU_TESTS=1 enables it; verify, then drop the guard")
def test_facade_switch_write_after_reset_recreates_the_handle_on_the_gpu():
    """main.py:130-141 assigns `multiple_tasks_per_agent = True` on the env object right after reset().  The facade re-creates the device handle with the new
    parameter and the episode's seed and swaps it in after comparing the resident state of the two handles (env.py: the property's setter).  Over the HIP
    backend the episode that follows must equal, call by call, the one the same facade gives over the oracle backend (which the CPU suite compares with the
    reference env: tests/test_facade_cpu.py), and a write after a step must raise."""
    from oracle_backend import OracleBackend
    from muavta_amd.env import MultiUAVEnv
    from muavta_amd.scenarios import CASE_SPECS

    flags = {"multiple_tasks_per_agent": False}
    hip, cpu = MultiUAVEnv(CASE_SPECS["static_strike"], flags=dict(flags)), MultiUAVEnv(CASE_SPECS["static_strike"], flags=dict(flags), backend_factory=OracleBackend)
    assert type(hip.backend).__name__ == "BatchedMultiUAVEnv" and hip.multiple_tasks_per_agent is False
    first = hip.backend
    for seed in (5, 6):
        o_hip, _ = hip.reset(seed=seed)
        o_cpu, _ = cpu.reset(seed=seed)
        hip.multiple_tasks_per_agent = True
        cpu.multiple_tasks_per_agent = True
        assert hip.multiple_tasks_per_agent is True and hip.backend is not first
        rng = np.random.default_rng(seed)
        for t in range(60):
            n_open = len(cpu.last_tasks_info)
            assert [x.id for x in hip.last_tasks_info] == [x.id for x in cpu.last_tasks_info], (seed, t)
            acts = {a.name: [int(x) for x in rng.integers(1, n_open, size=int(rng.integers(1, 4)))] for a in cpu.get_live_agents()[: 1 + t % 3]} if n_open > 1 and t % 4 == 0 else {}
            r_hip, r_cpu = hip.step({k: list(v) for k, v in acts.items()}), cpu.step({k: list(v) for k, v in acts.items()})
            assert r_hip[1] == r_cpu[1] and r_hip[2] == r_cpu[2] and r_hip[3] == r_cpu[3], (seed, t)
            for name in r_cpu[0]:
                for key in ("agent_position", "agent_caps", "alloc_task", "mask", "legal_mask", "event_flags"):
                    assert np.array_equal(np.asarray(r_hip[0][name][key]), np.asarray(r_cpu[0][name][key])), (seed, t, name, key)
            assert [[x.id for x in a.tasks] for a in hip.agents_obj] == [[x.id for x in a.tasks] for a in cpu.agents_obj], (seed, t)
            assert np.array_equal(np.array([a.position for a in hip.agents_obj]), np.array([a.position for a in cpu.agents_obj])), (seed, t)
    with pytest.raises(ValueError, match="right after reset"):
        hip.multiple_tasks_per_agent = False
    assert int(np.count_nonzero(hip.backend.get("ERROR"))) == 0
    hip.close(); cpu.close()
