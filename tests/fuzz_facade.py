#!/usr/bin/env python3
"""Wide differential fuzz of the FACADE, this container only (not part of the test suite): for random configurations
(fuzz_reference.wide_config) the reference's own HungarianAllocator + harness glue (experiments/paper_eval.py::_open_tasks / _events,
experiments/wps_eval.py::_apply_assign, imported unmodified) drive, in lockstep,
  (a) the reference's MultiUAVEnv and
  (b) muavta_amd.env.MultiUAVEnv over the test-only oracle backend,
and after every step everything a caller can see is compared: the observation dicts (f32 precision: the batched API's tensors are
f32), rewards, done flags, infos['events' / 'selected' / 'metrics'], last_tasks_info, the object views the allocators read
(positions, states, queues, next_free_*, caps, task reqs / status / deadlines / kinds, threats) and the visibility map.

    python tests/fuzz_facade.py [first_k [n_configs [procs]]] [--mutators] [--lists]

Lives under tests/ because it uses the oracle (through tests/oracle_backend.py) as the facade's backend."""
import os
import sys
import time
import traceback
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from fuzz_reference import wide_config  # noqa: E402


class QueueTooDeep(Exception):
    pass


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def same_obs(ro, fo, T, tag):
    assert set(ro) == set(fo), f"{tag}: observation keys"
    for name in ro:
        r, f = ro[name], fo[name]
        assert np.array_equal(f32(r["agent_position"]), f32(f["agent_position"])), f"{tag} {name}: agent_position"
        assert np.array_equal(f32(r["agent_caps"]), f32(f["agent_caps"])), f"{tag} {name}: agent_caps"
        assert int(r["alloc_task"]) == int(f["alloc_task"]), f"{tag} {name}: alloc_task"
        # (more than max_tasks open tasks: the reference's lists grow, DroneEnv.py:410-413; the fixed-width tensors keep max_tasks rows)
        rt, ft = r["tasks_info"][:T], f["tasks_info"]
        assert len(ft) == T and len(rt) == T, f"{tag} {name}: tasks_info length {len(ft)} / {len(rt)}"
        for j, (a, b) in enumerate(zip(rt, ft)):
            assert set(a) == set(b), f"{tag} {name}: tasks_info[{j}] keys {sorted(a)} vs {sorted(b)}"
            for k in a:
                if isinstance(a[k], np.ndarray) or isinstance(b[k], np.ndarray):
                    assert np.array_equal(f32(a[k]), f32(b[k])), f"{tag} {name}: tasks_info[{j}][{k}]"
                elif isinstance(a[k], float) or isinstance(b[k], float):
                    assert np.float32(a[k]) == np.float32(b[k]), f"{tag} {name}: tasks_info[{j}][{k}] {a[k]} vs {b[k]}"
                else:
                    assert a[k] == b[k], f"{tag} {name}: tasks_info[{j}][{k}] {a[k]} vs {b[k]}"
        assert list(r["mask"][:T]) == list(f["mask"]), f"{tag} {name}: mask"
        assert [bool(x) for x in r["legal_mask"][:T]] == list(f["legal_mask"]), f"{tag} {name}: legal_mask"
        assert np.array_equal(f32(r["event_flags"]), f32(f["event_flags"])), f"{tag} {name}: event_flags"


def same_world(R, F, tag):
    assert R.time_steps == F.time_steps, f"{tag}: time_steps"
    for k in ("F_Reward", "total_distance", "n_on_time", "n_missed_windows", "n_windowed_tasks", "n_task_switches", "n_reallocations",
              "n_arrivals", "conclusion_time", "escort_requests", "escort_completed", "escort_failed", "protection_breaches",
              "threats_intercepted", "recon_losses", "escort_losses"):
        assert getattr(R, k) == getattr(F, k), f"{tag}: env.{k} {getattr(R, k)} vs {getattr(F, k)}"
    assert [t.id for t in R.last_tasks_info] == [t.id for t in F.last_tasks_info], f"{tag}: last_tasks_info"
    assert [t.id for t in R.tasks] == [t.id for t in F.tasks], f"{tag}: env.tasks ids"
    assert [a.name for a in R.get_live_agents()] == [a.name for a in F.get_live_agents()], f"{tag}: live agents"
    for ra, fa in zip(R.agents_obj, F.agents_obj):
        assert ra.name == fa.name and ra.type == fa.type and ra.typeIdx == fa.typeIdx and ra.id == fa.id, f"{tag}: agent identity"
        assert np.array_equal(np.asarray(ra.position, dtype=np.float64), fa.position), f"{tag} {ra.name}: position"
        assert ra.state == fa.state, f"{tag} {ra.name}: state"
        if len(ra.tasks) > 16:
            raise QueueTooDeep(f"{ra.name} queues {len(ra.tasks)} tasks")  # (the test backend exports 16 queue columns; the device tiles hold 10-12)
        assert [t.id for t in ra.tasks] == [t.id for t in fa.tasks], f"{tag} {ra.name}: queue"
        assert np.array_equal(np.asarray(ra.next_free_position, dtype=np.float64), fa.next_free_position), f"{tag} {ra.name}: next_free_position"
        assert float(ra.next_free_time) == fa.next_free_time, f"{tag} {ra.name}: next_free_time"
        assert np.array_equal(np.asarray(ra.currentCap2Task, dtype=np.float64), fa.currentCap2Task), f"{tag} {ra.name}: caps"
        assert ra.max_speed == fa.max_speed and ra.engage_range == fa.engage_range, f"{tag} {ra.name}: speed / range"
        assert int(getattr(ra, "commit_until", 0) or 0) == int(fa.commit_until or 0), f"{tag} {ra.name}: commit_until"
    for rt, ft in zip(R.tasks, F.tasks):
        t = f"{tag} task {rt.id}"
        assert rt.status == ft.status, f"{t}: status {rt.status} vs {ft.status}"
        assert rt.type == ft.type and rt.typeIdx == ft.typeIdx, f"{t}: type"
        assert np.array_equal(np.asarray(rt.position, dtype=np.float64), ft.position), f"{t}: position {rt.position} vs {ft.position}"
        assert getattr(rt, "hard_deadline", None) == getattr(ft, "hard_deadline", None), f"{t}: hard_deadline"
        assert rt.kind == ft.kind, f"{t}: kind"
        assert int(rt.created_at or 0) == int(ft.created_at or 0), f"{t}: created_at"
        if rt.status != 2:  # (the views of a concluded task keep its last resident record)
            assert np.array_equal(rt.currentReqs, ft.currentReqs), f"{t}: currentReqs"
            assert np.array_equal(rt.allocatedReqs, ft.allocatedReqs), f"{t}: allocatedReqs"
            assert np.array_equal(rt.orgReqs[rt.typeIdx], ft.orgReqs[ft.typeIdx]), f"{t}: orgReqs"
            assert len(rt.allocationDetails) == len(ft.allocationDetails), f"{t}: len(allocationDetails)"
            assert int(rt.required_agents or 0) == int(ft.required_agents or 0), f"{t}: required_agents"
            pa, pb = getattr(rt, "protected_agent", None), getattr(ft, "protected_agent", None)
            assert (pa is None) == (pb is None) and (pa is None or pa.id == pb.id), f"{t}: protected_agent"
    assert [(h.id, int(h.status)) for h in R.threats] == [(h.id, int(h.status)) for h in F.threats], f"{tag}: threats"
    for rh, fh in zip(R.threats, F.threats):
        assert np.array_equal(np.asarray(rh.position, dtype=np.float64), fh.position), f"{tag} threat {rh.id}: position"
    rv, fv = R.agent_visibility_map(), F.agent_visibility_map()
    assert (rv is None) == (fv is None), f"{tag}: visibility map None-ness"
    if rv is not None:
        for n in rv:
            assert set(rv[n]) == set(fv[n]), f"{tag}: visibility of {n}: {sorted(set(rv[n]) ^ set(fv[n]))}"


def run_one(k: int):
    import refshim
    refshim.install()
    from experiments.paper_eval import _events, _open_tasks
    from experiments.wps_eval import _apply_assign
    from mUAV_TA.DroneEnv import MultiUAVEnv as RefEnv
    from mUAV_TA.MultiDroneEnvUtils import agentEnvOptions
    from TaskAllocation.OptimizationBased.HungarianAllocator import HungarianAllocator
    from muavta_amd.env import MultiUAVEnv as Facade
    from muavta_amd.params import METRIC_KEYS, params_from_config
    from oracle_backend import OracleBackend

    w = wide_config(k)
    cfg, interval, seed = w["cfg"], w["interval"], w["seed"]
    opts = agentEnvOptions(render_speed=-1, action_mode="TaskAssign", multiple_agents_per_task=True, fixed_seed=-1, **cfg)
    tag = f"WIDE{k}"
    try:
        R = RefEnv(opts)
        robs, rinfo = R.reset(seed=seed)
    except Exception as exc:
        return k, "skip", f"{type(exc).__name__}: {exc}"
    try:
        p = params_from_config(opts, None, tile_agents=64, tile_tasks=128, tile_threats=48)
        F = Facade(opts, backend=OracleBackend(p), tile_agents=64, tile_tasks=128, tile_threats=48)
        fobs, finfo = F.reset(seed=seed)
        T = R.max_tasks
        assert F.max_tasks == T and F.possible_agents == R.possible_agents and F.max_agents == R.max_agents
        same_obs(robs, fobs, T, f"{tag} reset")
        same_world(R, F, f"{tag} reset")
        mut, lists = "--mutators" in sys.argv, "--lists" in sys.argv
        rng = np.random.default_rng(4000 + k)
        hr = HungarianAllocator(replan_interval=interval, max_coord=R.max_coord)
        hf = HungarianAllocator(replan_interval=interval, max_coord=F.max_coord)
        rdone = {a: False for a in R.agents}
        rtrunc = dict(rdone)
        t = 0
        while not all(rdone.values()) and not all(rtrunc.values()):
            for _ in range(int(rng.integers(0, 3)) if mut else 0):
                # the out-of-step calls of the reference's own tests and planners (the op set of tests/fuzz_device.py::mutators), the same
                # call on the reference env and on the facade
                op = int(rng.integers(0, 8))
                ai_, ti_ = int(rng.integers(0, 64)), int(rng.integers(0, 64))
                vec = rng.uniform(50.0, 650.0, 2)
                val = int(rng.integers(0, 4))
                outs = []
                for e in (R, F):
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
                        if len(a.tasks) == 1 and a.tasks[0].id == 0:
                            a.tasks = [e.task_idle]
                            a.state = 0
                    elif op == 2:
                        a.position = np.array(vec)
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
                assert outs[0] == outs[1], f"{tag} t={t} op {op}: returned {outs[0]} vs {outs[1]}"
                same_world(R, F, f"{tag} t={t} after op {op}")
            if lists:  # no allocator: random scalar / LIST-valued actions keyed by agent name (repeated tasks, indices beyond the open list, dead agents) —
                # the action dict of tests/fuzz_reference.py --lists, here through the facade's own action packing
                names = [a.name for a in R.agents_obj]
                ra = {}
                for j in rng.permutation(len(names))[:int(rng.integers(0, len(names) + 1))]:
                    idxs = [int(rng.integers(0, 6)) if rng.random() < 0.9 else int(rng.integers(20, 140)) for _ in range(int(rng.integers(1, 7)))]
                    ra[names[j]] = idxs if (len(idxs) > 1 or rng.random() < 0.5) else idxs[0]
                fa = {n: (list(v) if isinstance(v, list) else v) for n, v in ra.items()}
                if t >= 100:
                    break
            else:
                try:
                    ra = _apply_assign(R, hr.allocate_tasks(R.get_live_agents(), _open_tasks(R), time_step=R.time_steps, events=_events(rinfo),
                                                             agent_known_ids=R.agent_visibility_map()))
                except Exception as exc:
                    return k, "skip", f"reference raised at t={t}: {type(exc).__name__}: {exc}"
                fa = _apply_assign(F, hf.allocate_tasks(F.get_live_agents(), _open_tasks(F), time_step=F.time_steps, events=_events(finfo),
                                                         agent_known_ids=F.agent_visibility_map()))
                assert ra == fa, f"{tag} t={t}: actions {ra} vs {fa}"
            try:
                robs, rrew, rdone, rtrunc, rinfo = R.step(ra)
            except Exception as exc:
                return k, "skip", f"reference raised in step {t}: {type(exc).__name__}: {exc}"
            fobs, frew, fdone, ftrunc, finfo = F.step(fa)
            t += 1
            st = f"{tag} t={t}"
            assert {n: float(v) for n, v in rrew.items()} == frew, f"{st}: rewards"
            assert dict(rdone) == fdone and dict(rtrunc) == ftrunc, f"{st}: done flags"
            assert [list(e) for e in rinfo["events"]] == finfo["events"], f"{st}: events {rinfo['events']} vs {finfo['events']}"
            assert rinfo.get("selected") == finfo.get("selected"), f"{st}: selected"
            same_obs(robs, fobs, T, st)
            same_world(R, F, st)
            assert ("metrics" in rinfo) == ("metrics" in finfo), f"{st}: metrics presence"
        if "metrics" in rinfo:  # (a --lists episode is cut at 100 steps: no final info then)
            assert list(rinfo["metrics"].keys()) == METRIC_KEYS == list(finfo["metrics"].keys())
            for key in METRIC_KEYS:
                assert float(rinfo["metrics"][key]) == float(finfo["metrics"][key]), f"{tag}: metric {key}"
        assert R.compute_s_wps() == F.compute_s_wps() and R.compute_s_esc() == F.compute_s_esc()
    except QueueTooDeep as exc:
        return k, "skip", str(exc)
    except AssertionError as exc:
        return k, "MISMATCH", str(exc)[:500]
    except Exception as exc:
        return k, "ERROR", "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-900:]
    return k, "ok", f"steps {t}"


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if len(args) > 0 else 0
    n = int(args[1]) if len(args) > 1 else 50
    procs = int(args[2]) if len(args) > 2 else 4
    counts = {}
    t0 = time.time()
    with Pool(procs) as pool:
        for k, status, msg in pool.imap_unordered(run_one, range(first, first + n)):
            counts[status] = counts.get(status, 0) + 1
            if status != "ok" or "--verbose" in sys.argv:
                print(f"k={k} {status}: {msg}", flush=True)
    print(f"configs {first}..{first + n - 1}: {counts}  ({time.time() - t0:.0f} s)", flush=True)
