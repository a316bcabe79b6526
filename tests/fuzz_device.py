#!/usr/bin/env python3
"""Wide configuration fuzz on the GPU (not part of the test suite): the random configurations of tests/fuzz_reference.py::wide_config
— the ones the CPU oracle is checked against the REFERENCE on in the container — run through the HIP library on all three tiles
and are compared with the oracle:

  fused     muavta_rollout over N seeds per (config, tile), the allocator mode rotating with the config number: all 30 metrics
            of every env that did not overflow its tile, bit for bit;
  stepwise  reset + max_time_steps x (allocate -> step) on one tile per config, every field of every env and the observation
            after every step (test_gpu_parity.compare), the token builders every fourth step;
  scored    the allocator with caller-supplied edge scores / task priorities / reserved agents (muavta_allocate_scored), a gate /
            flag / token-kind / pad combination per config, pseudo-random inputs per step, visibility toggling: plan, _selected_mask
            and the full state after every step;
  lists     list-valued actions (env.step({agent: [index, ...]}), DroneEnv.py:813-838): random lists — repeated tasks, indices beyond
            the open list, dead agents, more items than the tile's action capacity — through muavta_step_lists, state after every step;
  rings     the RECORDING rollout (muavta_rollout_record with per-step observation rings): slot t = what DroneEnv.step returned at step
            t (observation dict, reward, done), the slots behind an env's last step untouched, final state and metrics;
  mutators  the out-of-step calls of the reference's own tests and planners (muavta_call behind the facade: UAV.allocate,
            UAV.tasks = [task_idle], state / position / required_agents writes, _create_escort_for, _sync_escorts, _retire_escort,
            _escort_fighters_near, _is_task_action_valid) at random points of an episode, the facade on the HIP backend next to the
            facade on the oracle backend: every field after every call and every step;
  ilrings   the trainers' data rings (muavta_rollout_record through il.il_record: per-step token tensors, edge_valid, the expert's plan
            as a mask, the gate bit, the S_WPS series) against the oracle stepping the same seeds under the trainers' gate;
  inflight  two handles whose launches are queued alternately WITHOUT a synchronisation in between, several seed sets each (the seeding
            of launch i + 1 runs under launch i through the two seeding slots, batches of the two handles overlap on the device): the
            metrics of every handle's last batch;
  resume    checkpoint / resume and the sub-batch entry points: a fused rollout up to a random step, muavta_get_state + muavta_get_rng,
            a FRESH handle restored from them (muavta_set_state + muavta_set_rng), the rest of the episode there as sub-batch launches
            (muavta_set_parts + muavta_rollout_part): metrics and final state of the uninterrupted oracle episode;
  rl        the fused policy-in-the-loop step (muavta_rl_step_device through il.rl_stream: scored Hungarian -> step -> S_WPS ->
            next tokens in one launch) with a seeded score tensor per step: selected mask, gate, step reward, next tokens, done flags
            and the final metrics.

  rlrun     the policy consulted once per GATE (muavta_rl_run_device through il.rl_run_stream: the planned step, then empty-action steps up
            to the env's next gate, at most max_steps of them — drawn per config — under the trainer / escort / allocator gate): per launch
            the selected mask, gate bit, step reward, done, steps taken, park flags, reward sum, next and park tokens, against the oracle's
            run-to-the-gate; final metrics and state;
  steprun   the same run-ahead for a host-side plan (muavta_step_run: the device allocator's plan handed back as action rows or left
            staged): plan, steps taken, park flags, reward sum and every field + the observation of the state each env stopped in;
  lanes     ONE handle with two state lanes (muavta_set_lanes): seeded rollouts queued back to back, every batch read through
            muavta_rollout_metrics_back while the next one runs; all 30 metrics of every batch.

    python tests/fuzz_device.py [first_k [n_configs [seeds_per_config]]] [--more]

Lives under tests/ because it uses the oracle as its checker."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import orc  # noqa: E402
from fuzz_reference import wide_config  # noqa: E402
from muavta_amd.batched import BatchedMultiUAVEnv  # noqa: E402
from muavta_amd.params import params_from_config  # noqa: E402
from test_gpu_parity import Snapshot, compare  # noqa: E402

from test_gpu_parity import GATE  # noqa: E402

TILES = ((16, 40, 16), (24, 48, 24), (64, 128, 48))
SCORED = (("allocator", dict(edge_valid_only=True, full_task_list=True), "pair", 0, 3), ("force", dict(edge_valid_only=False, commit=True), "escort", 2, 4),
          ("escort", dict(edge_valid_only=True), "pair_raw", 1, 1), ("trainer", dict(edge_valid_only=True), "pair", 0, 1),
          ("trainer", dict(full_task_list=True, edge_valid_only=False), "pair", 0, 2), ("escort", dict(edge_valid_only=False, commit=True), "escort", 2, 4))
PADS = ((32, 16), (6, 3), (48, 16), (12, 8))
MODES = ((0, "hungarian"), (1, "urgency_pair"), (2, "urgency_coalition"), (3, "hungarian_gated"))


def tiles_for(cfg, k=None):
    """the tiles that hold the configuration's fleet and threat list (the large family of wide_config needs the bigger ones); with k
    also a CAPPED 16-agent tile (tile_tasks below 40 caps the live slots): the episodes that fit it run with the tile full or nearly
    full most of the time — on-demand slot recycling, the capacity edges — and the ones that do not are flagged and skipped"""
    na, nh = sum(cfg["agents"].values()), sum(n for _, n in cfg["threats_list"])
    out = tuple(t for t in TILES if na <= t[0] and nh <= t[2])
    if k is not None and na <= 16 and nh <= 16:
        n_static = sum(cfg["tasks"].values()) + len(cfg["threats_list"])
        out += ((16, min(39, n_static + 2 + (k * 7) % 12), 16),)
    return out


def reserved_bits(rng, n, A):
    """n random reserved-agent masks over A agents, each agent reserved with probability 1/4"""
    if A < 63:
        return rng.integers(0, 1 << A, n, dtype=np.uint64) & rng.integers(0, 1 << A, n, dtype=np.uint64)
    bits = rng.integers(0, 4, (n, A)) == 0
    return np.array([sum(1 << int(a) for a in np.nonzero(row)[0]) for row in bits], dtype=np.uint64)


def params(cfg, tile):
    from fuzz_device_params import params_of_wide
    return params_of_wide(cfg, tile)


def make_env(p, n, tile):
    """a handle on `tile`; a slot count below the smallest tile's 40 is the CAPPED variant (muavta_set_slot_cap: field widths stay the tile's)"""
    env = BatchedMultiUAVEnv(p, n)
    if tile[1] < 40:
        env.set_slot_cap(tile[1])
    return env


ESCALATED = [0]  # envs whose metric row came from rollout(escalate=True)


def fused(k, w, n_seeds, log):
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    mode, name = MODES[k % 4]
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n_seeds)], dtype=np.uint64)
    p0 = params(cfg, TILES[2])
    steps = p0.max_time_steps
    o = orc.OracleEnv(p0)
    want = []
    use_vis = k % 5 != 0    # every fifth config plans without the visibility map (Global-Hungarian)
    write_obs = k % 3 != 0  # every third skips the per-step observation write (a different instantiation of the step loop)
    split = (k % 7 == 0) and steps > 20  # ... and some run the episode as two launches, the second continuing without a reset
    for s in seeds:
        o.rollout_mode(int(s), steps, interval, int(use_vis), mode)
        want.append(o.metrics().copy())
    bad = flagged = checked = 0
    n_esc = [0]
    for tile in tiles_for(cfg, k):
        env = make_env(params(cfg, tile), n_seeds, tile)
        env.set_allocator(name)
        escalate = (not split) and k % 2 == 0 and tile[0] < 64  # every other config: flagged envs re-run on the next larger tile and spliced in
        if split:
            env.rollout(seeds, steps // 3, interval, use_vis, write_obs)
            env.rollout(None, steps - steps // 3, interval, use_vis, write_obs)
        else:
            try:
                env.rollout(seeds, steps, interval, use_vis, write_obs, escalate=escalate)
            except Exception as exc:  # (an env that overflows the LARGEST tile: rollout(escalate=True) says so)
                if "no tile left" not in str(exc):
                    raise
                escalate = False
        got, err = env.rollout_metrics(), env.get("ERROR")
        if escalate:  # rows of escalated envs come from the larger tile: every env is comparable
            esc = set(env.escalated)
            err = np.array([0 if i in esc else e for i, e in enumerate(err)])
            n_esc[0] += len(esc)
        for i in range(n_seeds):
            if err[i]:
                flagged += 1
                continue
            checked += 1
            if not np.array_equal(got[i], want[i]):
                bad += 1
                d = np.nonzero(got[i] != want[i])[0]
                log(f"k={k} FUSED MISMATCH tile {tile} mode {name} seed {int(seeds[i])} interval {interval} vis {use_vis} obs {write_obs} split {split}: metric columns {d.tolist()} got {got[i][d].tolist()} want {want[i][d].tolist()}")
        if tile == TILES[2] and err.any():
            log(f"k={k} note: {int((err != 0).sum())} envs overflow the 64 x 128 tile, codes {np.unique(err[err != 0]).tolist()}")
        env.close()
    ESCALATED[0] += n_esc[0]
    return bad, flagged, checked


def stepwise(k, w, log):
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    mode, name = MODES[(k // 3) % 4]
    tile = tiles_for(cfg, k)[k % len(tiles_for(cfg, k))]
    p = params(cfg, tile)
    n = 2
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = make_env(p, n, tile)
    env.set_allocator(name)
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    tag = f"k={k} tile {tile} mode {name} interval {interval}"
    try:
        snap = Snapshot(env)
        if snap.ERROR.any():
            return "overflow"
        for i, o in enumerate(oracles):
            compare(snap, i, o, f"{tag} seed {seeds[i]} after reset")
        for t in range(p.max_time_steps):
            done = [bool(o.dims()["terminated"] or o.dims()["truncated"]) for o in oracles]
            if any(done):
                break
            aa, ai = env.allocate(interval, True)
            plans = []
            for i, o in enumerate(oracles):
                oa, oi = o.allocate_mode(interval, 1, mode)
                kk = len(oa)
                assert np.array_equal(aa[i][:kk], oa) and np.all(aa[i][kk:] == -1) and np.array_equal(ai[i][:kk], oi), \
                    f"{tag} seed {seeds[i]} t={t}: plan {aa[i][:kk + 2]} / {ai[i][:kk + 2]} vs {oa} / {oi}"
                plans.append((oa, oi))
            if t % 4 == 3:  # the token builders (muavta_tokens) between the plan and the step (the expert mask is the staged plan's): kind and pads rotate
                kind, (mt, ma) = (t // 4 + k) % 3, PADS[(t // 4 + k // 3) % len(PADS)]
                got = env.tokens(("pair", "pair_raw", "escort")[kind], mt, ma)
                for i, o in enumerate(oracles):
                    want = o.tokens(kind, mt, ma)
                    for key in want:
                        gv = int(got[key][i]) if key == "n_urgent" else got[key][i]
                        assert np.array_equal(np.asarray(gv), np.asarray(want[key])), f"{tag} seed {seeds[i]} t={t}: tokens kind {kind} pads {mt}x{ma}: {key}"
                if kind != 2:  # ... and the ContextPair hybrids' context vector over the same token pad (muavta_context)
                    ctx = env.context(("pair", "pair_raw")[kind], mt)
                    for i, o in enumerate(oracles):
                        assert np.array_equal(ctx[i], o.context(kind, mt)), f"{tag} seed {seeds[i]} t={t}: context vector kind {kind} pad {mt}: {ctx[i]} vs {o.context(kind, mt)}"
            for i, o in enumerate(oracles):
                o.step(*plans[i])
            env.step(aa, ai)
            snap = Snapshot(env)
            if snap.ERROR.any():
                return "overflow"
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"{tag} seed {seeds[i]} t={t + 1}")
        m = env.metrics()
        for i, o in enumerate(oracles):
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: final metrics"
    except AssertionError as exc:
        log(f"STEPWISE MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def scored(k, w, log):
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    gate, kw, kname, kind, oflags = SCORED[k % len(SCORED)]
    mt, ma = PADS[(k // len(SCORED)) % len(PADS)]
    tile = tiles_for(cfg, k)[(k // 2) % len(tiles_for(cfg, k))]
    p = params(cfg, tile)
    n = 2
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = make_env(p, n, tile)
    A = env.n_agents
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    rng = np.random.default_rng(1000 + k)
    tag = f"k={k} scored tile {tile} gate {gate} kind {kname} pads {mt}x{ma} flags {kw} interval {interval}"
    try:
        if env.get("ERROR").any():
            return "overflow"
        for t in range(p.max_time_steps):
            if any(bool(o.dims()["terminated"] or o.dims()["truncated"]) for o in oracles):
                break
            sc = (rng.uniform(-1, 1, (n, ma, mt)) * (0.35 if kind != 2 else 1.0)).astype(np.float32)
            pri = rng.uniform(-0.5, 1, (n, mt))
            res = reserved_bits(rng, n, A)
            vis = bool((t // 3) % 2)
            out = env.allocate_scored(kname, mt, ma, edge_scores=sc, task_pri=pri, reserved=res, gate=gate, replan_interval=interval, use_visibility=vis, **kw)
            for i, o in enumerate(oracles):
                oa, oi, osel = o.allocate_scored(interval, int(vis), GATE[gate], kind, mt, ma, oflags, scores=sc[i], pri=pri[i], reserved=int(res[i]))
                kk = len(oa)
                assert np.array_equal(out["act_agent"][i][:kk], oa) and np.all(out["act_agent"][i][kk:] == -1) and np.array_equal(out["act_index"][i][:kk], oi), \
                    f"{tag} seed {seeds[i]} t={t}: plan {out['act_agent'][i][:kk + 2]} / {out['act_index'][i][:kk + 2]} vs {oa} / {oi}"
                assert np.array_equal(out["selected"][i], osel), f"{tag} seed {seeds[i]} t={t}: selected mask"
                assert bool(out["replanned"][i]) == (o.scalars_last_plan() == t), f"{tag} seed {seeds[i]} t={t}: gate"
                o.step(oa, oi)
            env.step_staged()
            snap = Snapshot(env)
            if snap.ERROR.any():
                return "overflow"
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"{tag} seed {seeds[i]} t={t + 1}")
    except AssertionError as exc:
        log(f"SCORED MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def lists(k, w, log):
    cfg, seed = w["cfg"], w["seed"]
    tile = tiles_for(cfg, k)[(k // 5) % len(tiles_for(cfg, k))]
    p = params(cfg, tile)
    n = 2
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = make_env(p, n, tile)
    A = env.n_agents
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    rng = np.random.default_rng(2000 + k)
    tag = f"k={k} lists tile {tile}"
    try:
        if env.get("ERROR").any():
            return "overflow"
        for t in range(min(p.max_time_steps, 80)):
            if any(bool(o.dims()["terminated"] or o.dims()["truncated"]) for o in oracles):
                break
            rows = []
            for i in range(n):
                items = []
                for a in rng.permutation(A)[:int(rng.integers(0, A + 1))]:
                    for _ in range(int(rng.integers(1, 7))):
                        items.append((int(a), int(rng.integers(0, 5)) if rng.random() < 0.9 else int(rng.integers(20, 140))))
                rows.append(items)
            aa, ai = env.pack_actions(rows)
            env.step(aa, ai)
            for i, o in enumerate(oracles):
                o.step(np.array([x[0] for x in rows[i]], dtype=np.int32), np.array([x[1] for x in rows[i]], dtype=np.int32))
            snap = Snapshot(env)
            if snap.ERROR.any():
                return "overflow"
            for i, o in enumerate(oracles):
                compare(snap, i, o, f"{tag} seed {seeds[i]} t={t + 1}")
    except AssertionError as exc:
        log(f"LISTS MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def rings(k, w, log):
    import torch

    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    tile = tiles_for(cfg)[(k // 11) % len(tiles_for(cfg))]
    p = params(cfg, tile)
    n, steps = 3, p.max_time_steps
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = BatchedMultiUAVEnv(p, n)
    tag = f"k={k} rings tile {tile} interval {interval}"
    try:
        dev = torch.device("cuda", env.device_index)
        R = {key: torch.zeros(shape, dtype=getattr(torch, np.dtype(dt).name), device=dev) for key, (shape, dt) in env.obs_ring_shapes(steps).items()}
        env.rollout_record(seeds, steps, interval, True, obs_rings=R)
        env.sync()
        snap = Snapshot(env)
        R = {key: v.cpu().numpy() for key, v in R.items()}
        MT = env.max_tasks
        m = env.rollout_metrics()
        for i in range(n):
            if snap.ERROR[i]:
                continue
            o = orc.OracleEnv(p)
            o.reset(int(seeds[i]))
            t_end = steps
            for t in range(steps):
                a, ix = o.allocate_mode(interval, 1, 0)
                o.step(a, ix)
                ti, legal, pad, ag, fl = o.observe()
                st = f"{tag} seed {seeds[i]} slot {t}"
                assert np.array_equal(R["obs_tasks"][t, i].T, ti), f"{st}: tasks_info"
                bits = ((R["obs_legal"][t, i][:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(legal.shape[0], -1)[:, :MT]
                assert np.array_equal(bits.astype(bool), legal), f"{st}: legal_mask"
                assert np.array_equal(R["obs_pad"][t, i].astype(bool), pad), f"{st}: pad mask"
                assert np.array_equal(R["obs_agents"][t, i], ag), f"{st}: agent rows"
                assert np.array_equal(R["obs_flags"][t, i], fl), f"{st}: event flags"
                d = o.dims()
                assert R["obs_reward"][t, i] == o.scalars()[1], f"{st}: reward"
                assert R["obs_done"][t, i] == (1 if d["terminated"] else 0) | (2 if d["truncated"] else 0), f"{st}: done flags"
                if d["terminated"] or d["truncated"]:
                    t_end = t + 1
                    break
            assert np.all(R["obs_done"][t_end:, i] == env.OBS_UNWRITTEN), f"{tag} seed {seeds[i]}: slots after the last step"
            compare(snap, i, o, f"{tag} seed {seeds[i]} after the recorded rollout")
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: metrics"
    except AssertionError as exc:
        log(f"RINGS MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def mutators(k, w, log, verbose=False):
    from oracle_backend import OracleBackend
    from muavta_amd.env import MultiUAVEnv

    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    tile = tiles_for(cfg)[(k // 13) % len(tiles_for(cfg))]
    c = dict(cfg)
    c["threats_list"] = [tuple(x) for x in c["threats_list"]]
    c["escort_agent_types"] = tuple(c["escort_agent_types"])
    p = params(cfg, tile)
    tiles = dict(tile_agents=tile[0], tile_tasks=tile[1], tile_threats=tile[2])
    hip = MultiUAVEnv(dict(c), **tiles)
    ref = MultiUAVEnv(dict(c), backend=OracleBackend(p), **tiles)
    envs = (hip, ref)
    rng = np.random.default_rng(4000 + k)
    tag = f"k={k} mutators tile {tile}"
    try:
        for e in envs:
            e.reset(seed=seed)
        for t in range(min(p.max_time_steps, 60)):
            if hip._b.get("ERROR").any():
                return "overflow"
            for _ in range(int(rng.integers(0, 3))):
                op = int(rng.integers(0, 8))
                ai_, ti_ = int(rng.integers(0, 64)), int(rng.integers(0, 64))
                vec = rng.uniform(50.0, 650.0, 2)
                val = int(rng.integers(0, 4))
                outs = []
                for e in envs:
                    live = e.get_live_agents()
                    if not live:
                        outs.append(None)
                        continue
                    a = live[ai_ % len(live)]
                    open_ = list(e.last_tasks_info)
                    task = open_[ti_ % len(open_)] if open_ else None
                    r = None
                    if op == 0 and task is not None:
                        r = (bool(e._is_task_action_valid(a, task)),)
                        if r[0]:
                            r += (bool(a.allocate(task, e.time_steps)),)
                    elif op == 1:
                        # (`agent.tasks = [...]` is a plain list assignment in the reference — the tasks' allocationDetails keep the agent.  Its
                        # own tests only scaffold IDLE agents that way (test_escort.py:95), and that is the supported contract: the device
                        # keeps allocationDetails in the queues)
                        if len(a.tasks) == 1 and a.tasks[0].id == 0:
                            a.tasks = [e.task_idle]
                            a.state = 0
                    elif op == 2:
                        a.position = vec
                    elif op == 3 and task is not None:
                        task.required_agents = val
                    elif op == 4 and e.escort_enabled and task is not None and a.type in ("R1", "R2") and task.type == "Rec":
                        esc = e._create_escort_for(a, task)
                        r = None if esc is None else (esc.id, esc.required_agents)
                    elif op == 5 and e.escort_enabled:
                        e._sync_escorts()
                    elif op == 6 and e.escort_enabled and e._escort_by_recon:
                        names = sorted(e._escort_by_recon)
                        e._retire_escort(e._escort_by_recon[names[ti_ % len(names)]], failed=bool(val & 1))
                        r = (e.escort_completed, e.escort_failed)
                    elif op == 7 and e.escort_enabled:
                        r = [x.id for x in e._escort_fighters_near(a, float(vec[0]))]
                    outs.append(r)
                    if verbose and e is hip:
                        log(f"   t={t} op {op} agent {a.name} (queue {[x.id for x in a.tasks]}, state {a.state}) task {None if task is None else (task.id, task.type)} vec {vec.tolist()} val {val} -> {r}")
                if hip._b.get("ERROR").any():  # (e.g. UAV.allocate beyond the tile's queue depth returns False AND raises the capacity flag)
                    return "overflow"
                assert outs[0] == outs[1], f"{tag} t={t} op {op}: returned {outs[0]} vs {outs[1]}"
                compare(Snapshot(hip._b), 0, ref._b.o, f"{tag} t={t} after op {op}", check_obs=False)
            acts = []
            for e in envs:
                aa, ai = e._b.allocate(interval, True)
                acts.append({e.agents_obj[int(a)].name: int(i) for a, i in zip(aa[0], ai[0]) if a >= 0})
            if verbose and acts[0] != acts[1]:
                o = ref._b.o
                shapes, costs, rws, cls = o.lsap_calls()
                log(f"   oracle LSAP shapes {shapes.tolist()} first costs {costs[:40].tolist()} rows {rws.tolist()} cols {cls.tolist()}")
                trow, reqs = o.tasks()
                log(f"   open {o.open_ids().tolist()}")
                for kk in o.open_ids().tolist():
                    log(f"   task {kk}: type {int(trow[kk, 6])} required {int(trow[kk, 9])} ndet {int(trow[kk, 5])} cur {reqs[kk, 0].tolist()} alloc {reqs[kk, 1].tolist()} pos {trow[kk, 1:3].tolist()}")
                log(f"   device staged {hip._b.get('STAGED_ACTIONS')[0][:8].tolist()}  oracle last_actions {o.last_actions().tolist()}")
                sn = Snapshot(hip._b)
                log(f"   device task ids {sn.TASK_ID[0][:16].tolist()} meta required {[int(m[3]) for m in sn.TASK_META[0][:16]]} ndet {[int(m[5]) for m in sn.TASK_META[0][:16]]}")
            assert acts[0] == acts[1], f"{tag} t={t}: plans {acts[0]} vs {acts[1]}"
            done = False
            for e, ac in zip(envs, acts):
                _, _, term, trunc, _ = e.step(ac)
                done = all(term.values()) or all(trunc.values())
            if hip._b.get("ERROR").any():
                return "overflow"
            compare(Snapshot(hip._b), 0, ref._b.o, f"{tag} step {t + 1}")
            if done:
                break
    except AssertionError as exc:
        log(f"MUTATORS MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        hip._b.close()
    return "ok"


def ilrings(k, w, log):
    from muavta_amd.il import il_record

    cfg, seed = w["cfg"], w["seed"]
    tile = tiles_for(cfg)[(k // 17) % len(tiles_for(cfg))]
    p = params(cfg, tile)
    kind = k % 3
    kname = ("pair", "pair_raw", "escort")[kind]
    mt, ma = ((32, 16), (12, 8), (48, 16))[(k // 3) % 3]
    n, steps, interval = 3, p.max_time_steps, 20
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = BatchedMultiUAVEnv(p, n)
    tag = f"k={k} ilrings tile {tile} kind {kname} pads {mt}x{ma}"
    try:
        rec = {key: v.cpu().numpy() for key, v in il_record(env, seeds, steps, interval, kname, mt, ma).items()}
        err = env.get("ERROR")
        m = env.rollout_metrics()
        for i in range(n):
            if err[i]:
                continue
            o = orc.OracleEnv(p)
            o.reset(int(seeds[i]))
            assert rec["s_wps"][0, i] == o.metrics()[4], f"{tag} seed {seeds[i]}: S_WPS[0]"
            for t in range(steps):
                oa, oi = o.allocate_mode(interval, 0, 3)  # the trainers' expert: Global-Hungarian (no visibility map, train_pair_cost.py:111) under _should_replan(env, events, interval)
                want = o.tokens(kind, mt, ma)
                for key in want:
                    gv = int(rec[key][t, i]) if key == "n_urgent" else rec[key][t, i]
                    assert np.array_equal(np.asarray(gv), np.asarray(want[key])), f"{tag} seed {seeds[i]} t={t}: {key}"
                assert bool(rec["replanned"][t, i]) == (o.scalars_last_plan() == t), f"{tag} seed {seeds[i]} t={t}: gate"
                done = o.step(oa, oi)
                assert rec["s_wps"][t + 1, i] == o.metrics()[4], f"{tag} seed {seeds[i]} t={t}: S_WPS series"
                if done:
                    break
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: metrics"
    except AssertionError as exc:
        log(f"ILRINGS MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def inflight(k, w, log):
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    tile = tiles_for(cfg)[(k // 19) % len(tiles_for(cfg))]
    p = params(cfg, tile)
    n, steps, rounds = 8, p.max_time_steps, 4
    mode, name = MODES[(k // 2) % 4]
    base = min(seed, 2 ** 62)
    envs = [BatchedMultiUAVEnv(p, n) for _ in range(2)]
    tag = f"k={k} inflight tile {tile} mode {name}"
    try:
        last = [None, None]
        for e in envs:
            e.set_allocator(name)
        for r in range(rounds):
            for j, e in enumerate(envs):
                sd = np.arange(base + 100 * (2 * r + j), base + 100 * (2 * r + j) + n, dtype=np.uint64)
                e.rollout(sd, steps, interval, True, bool((k + r) & 1))  # queued: no sync between launches or handles
                last[j] = sd
        for j, e in enumerate(envs):
            e.sync()
            got, err = e.rollout_metrics(), e.get("ERROR")
            o = orc.OracleEnv(p)
            for i in range(n):
                if err[i]:
                    continue
                o.rollout_mode(int(last[j][i]), steps, interval, 1, mode)
                assert np.array_equal(got[i], o.metrics()), f"{tag} handle {j} seed {int(last[j][i])}: metric columns {np.nonzero(got[i] != o.metrics())[0].tolist()}"
    except AssertionError as exc:
        log(f"INFLIGHT MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        for e in envs:
            e.close()
    return "ok"


def resume(k, w, log):
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    tile = tiles_for(cfg)[(k // 3) % len(tiles_for(cfg))]
    p = params(cfg, tile)
    n, steps = 6, p.max_time_steps
    cut = 1 + (k * 13) % max(steps - 1, 1)
    mode, name = MODES[(k // 5) % 4]
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    a = BatchedMultiUAVEnv(p, n)
    b = BatchedMultiUAVEnv(p, n)
    tag = f"k={k} resume tile {tile} mode {name} cut {cut}/{steps}"
    try:
        a.set_allocator(name)
        a.rollout(seeds, cut, interval, True, True)
        a.sync()
        st, rg = a.get_state(), a.get_rng()
        b.set_allocator(name)
        b.set_state(st)
        b.set_rng(rg)
        parts = 2 + k % 2
        b.set_parts(parts)
        for part in range(parts):
            b.rollout_part(part, steps - cut, interval, True, True)
        b.wait_part(-1)
        b.sync()
        b.refresh_observation()  # (an env whose episode ended before the cut takes no step in b: its observation is b's to rebuild from the restored state)
        snap = Snapshot(b)
        m = b.rollout_metrics()  # (the rows of the last launches; muavta_metrics refuses a batch with a capacity-flagged env)
        for i in range(n):
            if snap.ERROR[i]:
                continue
            o = orc.OracleEnv(p)
            o.rollout_mode(int(seeds[i]), steps, interval, 1, mode)
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: metrics columns {np.nonzero(m[i] != o.metrics())[0].tolist()}"
            compare(snap, i, o, f"{tag} seed {seeds[i]} final state")
    except AssertionError as exc:
        log(f"RESUME MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        a.close(); b.close()
    return "ok"


def rl(k, w, log):
    import torch
    from muavta_amd import il

    cfg, seed = w["cfg"], w["seed"]
    kind = (0, 1)[k % 2]
    kname = ("pair", "pair_raw")[kind]
    mt, ma = ((32, 16), (12, 8), (48, 16))[(k // 2) % 3]
    tile = tiles_for(cfg)[(k // 7) % len(tiles_for(cfg))]
    p = params(cfg, tile)
    n = 3
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = BatchedMultiUAVEnv(p, n)
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    rng = np.random.default_rng(3000 + k)
    dev = torch.device("cuda", 0)
    cur = {}

    def policy(tok):
        cur["sc"] = (rng.uniform(-1, 1, (n, ma, mt)) * 0.35).astype(np.float32)
        return torch.from_numpy(cur["sc"]).to(dev)

    tag = f"k={k} rl tile {tile} kind {kname} pads {mt}x{ma}"
    done_o = [False] * n
    try:
        for t, tr in il.rl_stream(env, seeds, policy, n_steps=p.max_time_steps, interval=20, kind=kname, max_tasks=mt, max_agents=ma, gate="trainer", fused=True):
            if env.get("ERROR").any():
                return "overflow"
            sel, rep = tr["selected"].cpu().numpy(), tr["replanned"].cpu().numpy()
            rew, dn = tr["step_reward"].cpu().numpy(), tr["done"].cpu().numpy()
            nxt = {key: v.cpu().numpy() for key, v in tr["next_tok"].items()}
            for i, o in enumerate(oracles):
                if done_o[i]:
                    continue
                s0 = o.metrics()[4]
                oa, oi, osel = o.allocate_scored(20, 1, GATE["trainer"], kind, mt, ma, 1, scores=cur["sc"][i])
                assert np.array_equal(sel[i], osel), f"{tag} seed {seeds[i]} t={t}: selected mask"
                assert bool(rep[i]) == (o.scalars_last_plan() == t), f"{tag} seed {seeds[i]} t={t}: gate"
                d = o.step(oa, oi)
                assert rew[i] == (o.metrics()[4] - s0) / 20.0, f"{tag} seed {seeds[i]} t={t}: step reward {rew[i]} vs {(o.metrics()[4] - s0) / 20.0}"
                dd = o.dims()
                assert int(dn[i]) == (int(bool(dd["terminated"])) | (int(bool(dd["truncated"])) << 1)), f"{tag} seed {seeds[i]} t={t}: done"
                want = o.tokens(kind, mt, ma)
                for key in want:
                    if key in nxt:
                        gv = int(nxt[key][i]) if key == "n_urgent" else nxt[key][i]
                        assert np.array_equal(np.asarray(gv), np.asarray(want[key])), f"{tag} seed {seeds[i]} t={t}: next tokens: {key}"
                done_o[i] = bool(d)
            if all(done_o):
                break
        m = env.metrics()
        for i, o in enumerate(oracles):
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: final metrics"
    except AssertionError as exc:
        log(f"RL MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def rlrun(k, w, log):
    import torch
    from muavta_amd import il

    cfg, seed = w["cfg"], w["seed"]
    kind = (0, 1)[k % 2]
    kname = ("pair", "pair_raw")[kind]
    mt, ma = ((32, 16), (12, 8), (48, 16))[(k // 2) % 3]
    tile = tiles_for(cfg)[(k // 7) % len(tiles_for(cfg))]
    gname, interval = (("trainer", 20), ("escort", 12), ("allocator", 7), ("trainer", 5))[(k // 3) % 4]
    max_steps = (0, 0, 3, 1, 7)[(k // 5) % 5]
    p = params(cfg, tile)
    n = 3
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = BatchedMultiUAVEnv(p, n)
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    rng = np.random.default_rng(5000 + k)
    dev = torch.device("cuda", 0)
    cur = {}

    def policy(tok):
        cur["sc"] = (rng.uniform(-1, 1, (n, ma, mt)) * 0.35).astype(np.float32)
        return torch.from_numpy(cur["sc"]).to(dev)

    tag = f"k={k} rlrun tile {tile} kind {kname} pads {mt}x{ma} gate {gname}/{interval} max_steps {max_steps}"
    keys = ("task_feats", "agent_feats", "edge_valid", "task_ids", "agent_ids", "task_mask", "agent_mask")
    try:
        launches = 0
        for kk, tr in il.rl_run_stream(env, seeds, policy, interval=interval, kind=kname, max_tasks=mt, max_agents=ma, gate=gname, max_steps=max_steps,
                                       max_launches=4 * p.max_time_steps + 8):
            launches += 1
            if env.get("ERROR").any():
                return "overflow"
            sel, rep = tr["selected"].cpu().numpy(), tr["replanned"].cpu().numpy()
            rew, dn = tr["step_reward"].cpu().numpy(), tr["done"].cpu().numpy()
            nst, prk, rsum = tr["n_stepped"].cpu().numpy(), tr["park"].cpu().numpy(), tr["reward_sum"].cpu().numpy()
            nxt = {key: v.cpu().numpy() for key, v in tr["next_tok"].items()}
            ptk = {key: v.cpu().numpy() for key, v in tr["park_tok"].items()}
            for i, o in enumerate(oracles):
                r = o.rl_run(interval, 1, GATE[gname], kind, mt, ma, 1, scores=cur["sc"][i], max_steps=max_steps)
                t = f"{tag} seed {seeds[i]} launch {kk}"
                assert int(nst[i]) == r["n_stepped"] and int(prk[i]) == r["park"] and bool(rep[i]) == r["replanned"], f"{t}: steps / park / gate {nst[i]} {prk[i]} {rep[i]} vs {r['n_stepped']} {r['park']} {r['replanned']}"
                assert np.array_equal(sel[i], r["selected"]), f"{t}: selected mask"
                assert rew[i] == (r["s_after"] - r["s_before"]) / 20.0 and int(dn[i]) == r["done"], f"{t}: step reward / done"
                assert rsum[i] == r["reward_sum"], f"{t}: reward sum {rsum[i]} vs {r['reward_sum']}"
                for which, got, want in (("next", nxt, r.get("next_tok")), ("park", ptk, r["park_tok"])):
                    if want is None:
                        continue
                    for key in keys:
                        assert np.array_equal(got[key][i], want[key]), f"{t}: {which} tokens: {key}"
                    assert int(got["n_urgent"][i]) == want["n_urgent"], f"{t}: {which} tokens: n_urgent"
        assert all(o.dims()["terminated"] or o.dims()["truncated"] for o in oracles), f"{tag}: the stream ended after {launches} launches with an episode still running"
        m = env.metrics()
        env.refresh_observation()
        snap = Snapshot(env)
        for i, o in enumerate(oracles):
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: final metrics"
            compare(snap, i, o, f"{tag} seed {seeds[i]} final state", check_obs=False)
    except AssertionError as exc:
        log(f"RLRUN MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def steprun(k, w, log):
    cfg, seed, interval = w["cfg"], w["seed"], w["interval"]
    tile = tiles_for(cfg)[(k // 11) % len(tiles_for(cfg))]
    gname = ("trainer", "escort", "allocator")[k % 3]
    p = params(cfg, tile)
    n = 3
    seeds = np.array([seed - i if seed > 2 ** 62 else seed + i for i in range(n)], dtype=np.uint64)
    env = BatchedMultiUAVEnv(p, n)
    oracles = [orc.OracleEnv(p) for _ in range(n)]
    env.reset(seeds)
    for i, o in enumerate(oracles):
        o.reset(int(seeds[i]))
    tag = f"k={k} steprun tile {tile} gate {gname}/{interval}"
    try:
        launches = 0
        while True:
            cap = (0, 4, 0, 1)[(k + launches) % 4]
            aa, ai = env.allocate(interval, True)
            if env.get("ERROR").any():
                return "overflow"
            for i, o in enumerate(oracles):
                oa, oi = o.allocate(interval, 1)
                kk = len(oa)
                assert np.array_equal(aa[i][:kk], oa) and np.all(aa[i][kk:] == -1) and np.array_equal(ai[i][:kk], oi), f"{tag} seed {seeds[i]} launch {launches}: plan"
            nst, prk, rs = env.step_run(aa, ai, gate=gname, replan_interval=interval, max_steps=cap) if launches % 2 else env.step_run(None, None, gate=gname, replan_interval=interval, max_steps=cap)
            snap = Snapshot(env)
            if snap.ERROR.any():
                return "overflow"
            for i, o in enumerate(oracles):
                d = o.dims()
                if d["terminated"] or d["truncated"]:
                    assert nst[i] == 0 and (prk[i] & 3), f"{tag} seed {seeds[i]} launch {launches}: an ended episode stepped"
                    continue
                na = int((aa[i] >= 0).sum())
                o.step(aa[i][:na], ai[i][:na])
                q, ag, rq = o.run_quiet(GATE[gname], interval, cap, 1, float(o.scalars()[1]))
                dd = o.dims()
                want_park = int(dd["terminated"]) | (int(dd["truncated"]) << 1) | (4 if ag else 0)
                assert int(nst[i]) == 1 + q and int(prk[i]) == want_park and rs[i] == rq, f"{tag} seed {seeds[i]} launch {launches}: {nst[i]} {prk[i]} {rs[i]} vs {1 + q} {want_park} {rq}"
                compare(snap, i, o, f"{tag} seed {seeds[i]} launch {launches}")
            launches += 1
            if np.all(prk & 3):
                break
            assert launches <= 4 * p.max_time_steps + 8, f"{tag}: no end"
        m = env.metrics()
        for i, o in enumerate(oracles):
            assert np.array_equal(m[i], o.metrics()), f"{tag} seed {seeds[i]}: final metrics"
    except AssertionError as exc:
        log(f"STEPRUN MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


def lanes(k, w, log):
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    tile = tiles_for(cfg)[(k // 13) % len(tiles_for(cfg))]
    p = params(cfg, tile)
    n, steps, rounds = 8, p.max_time_steps, 5
    mode, name = MODES[(k // 2) % 4]
    base = min(seed, 2 ** 62)
    env = BatchedMultiUAVEnv(p, n)
    tag = f"k={k} lanes tile {tile} mode {name}"
    try:
        env.set_allocator(name)
        env.set_lanes(2)
        batches = [np.arange(base + 100 * r, base + 100 * r + n, dtype=np.uint64) for r in range(rounds)]
        got, errs = [], []
        for r, sd in enumerate(batches):
            env.rollout(sd, steps, interval, True, bool((k + r) & 1))  # queued: batch r - 1 is read while batch r runs
            if r:
                got.append(env.rollout_metrics(back=1)); errs.append(env.error_flags(back=1))
        got.append(env.rollout_metrics()); errs.append(env.error_flags())
        o = orc.OracleEnv(p)
        for r, sd in enumerate(batches):
            for i in range(n):
                if errs[r][i]:
                    continue
                o.rollout_mode(int(sd[i]), steps, interval, 1, mode)
                assert np.array_equal(got[r][i], o.metrics()), f"{tag} batch {r} seed {int(sd[i])}: metric columns {np.nonzero(got[r][i] != o.metrics())[0].tolist()}"
    except AssertionError as exc:
        log(f"LANES MISMATCH {str(exc)[:600]}")
        return "bad"
    finally:
        env.close()
    return "ok"


LEGS = ("stepwise", "scored", "lists", "rl", "rings", "mutators", "resume", "ilrings", "inflight", "rlrun", "steprun", "lanes")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if len(args) > 0 else 0
    n_cfg = int(args[1]) if len(args) > 1 else 50
    n_seeds = int(args[2]) if len(args) > 2 else 16

    def log(msg):
        print(msg, flush=True)

    t0 = time.time()
    tot = {"fused_bad": 0, "fused_flagged": 0, "fused_checked": 0, "step_ok": 0, "step_bad": 0, "step_overflow": 0, "scored_ok": 0, "scored_bad": 0,
           "scored_overflow": 0, "lists_ok": 0, "lists_bad": 0, "lists_overflow": 0, "rl_ok": 0, "rl_bad": 0, "rl_overflow": 0, "rings_ok": 0, "rings_bad": 0, "mutators_ok": 0, "mutators_bad": 0, "mutators_overflow": 0, "resume_ok": 0, "resume_bad": 0, "resume_overflow": 0, "ilrings_ok": 0, "ilrings_bad": 0, "inflight_ok": 0, "inflight_bad": 0,
           "rlrun_ok": 0, "rlrun_bad": 0, "rlrun_overflow": 0, "steprun_ok": 0, "steprun_bad": 0, "steprun_overflow": 0, "lanes_ok": 0, "lanes_bad": 0, "errors": 0}
    for k in range(first, first + n_cfg):
        w = wide_config(k)
        try:
            b, f, c = fused(k, w, n_seeds, log)
            tot["fused_bad"] += b; tot["fused_flagged"] += f; tot["fused_checked"] += c
            tot["step_" + stepwise(k, w, log)] += 1
            tot["scored_" + scored(k, w, log)] += 1
            if "--more" in sys.argv:
                tot["lists_" + lists(k, w, log)] += 1
                tot["rl_" + rl(k, w, log)] += 1
                tot["rings_" + rings(k, w, log)] += 1
                tot["mutators_" + mutators(k, w, log)] += 1
                tot["resume_" + resume(k, w, log)] += 1
                tot["ilrings_" + ilrings(k, w, log)] += 1
                tot["inflight_" + inflight(k, w, log)] += 1
                tot["rlrun_" + rlrun(k, w, log)] += 1
                tot["steprun_" + steprun(k, w, log)] += 1
                tot["lanes_" + lanes(k, w, log)] += 1
        except Exception as exc:  # a configuration the library rejects (muavta_create's argument checks): reported, not fatal
            if "overflowed a tile" in str(exc):  # (a capacity flag met by a call that refuses flagged batches, e.g. the facade's metrics)
                tot["capacity_exceptions"] = tot.get("capacity_exceptions", 0) + 1
                continue
            tot["errors"] += 1
            log(f"k={k} ERROR {type(exc).__name__}: {str(exc)[:300]}")
        if (k - first) % 10 == 9:
            log(f"... {k - first + 1} configs, {time.time() - t0:.0f} s: {tot}")
    log(f"configs {first}..{first + n_cfg - 1}, {n_seeds} seeds each on 3 tiles: {tot}  escalated envs compared: {ESCALATED[0]}  ({time.time() - t0:.0f} s)")
