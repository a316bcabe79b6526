"""Replay / frame export in the reference's dashboard schema.

`frame()` restates `_frame` of experiments/generate_simulation_replay.py:120-222 on top of the facade's object views
(so it reads device state through `muavta_get`), `infer_events()` the reviewer events of :60-117, and `generate()` the
episode loop + JSON document of :225-306 — with the planner running ON THE DEVICE (`muavta_set_allocator` +
`muavta_allocate`): 'urgency_coalition' is the reference's "Urgency-Coalition + Coalition-Hungarian" replay for
WPS_escort.  The file `generate()` writes is what server/api.py:64-93 serves to the dashboard.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Optional

from .env import MultiUAVEnv
from .scenarios import CASE_SPECS, WPS_ENV_FLAGS

REPLAN_TAGS = ("Reset_Allocation", "New_Threat", "Agent_Fail", "Escort_Created", "Escort_Retired")
ALGORITHM_LABEL = {"urgency_coalition": "Urgency-Coalition + Coalition-Hungarian", "urgency_pair": "Urgency-Pair + Local-Hungarian",
                   "hungarian": "Local-Hungarian"}


def should_replan(env, events, interval: int = 15) -> bool:  # generate_simulation_replay.py:21-37
    if env.time_steps == 0 or env.time_steps % interval == 0:
        return True
    return any((ev[0] if isinstance(ev, (list, tuple)) and ev else ev) in REPLAN_TAGS for ev in events)


def event_record(event, time_step: int) -> dict:  # :48-54
    if isinstance(event, (list, tuple)):
        return {"time": time_step, "type": str(event[0]) if event else "Unknown", "detail": [str(v) for v in event[1:]]}
    return {"time": time_step, "type": str(event), "detail": []}


def frame(env, events: list, replanned: bool, committed: List[str]) -> dict:
    """One dashboard frame of `env` (the facade or any object with the reference env's attributes)."""
    visibility = env.agent_visibility_map()
    if not visibility:
        visibility = {a.name: {t.id for t in env.tasks if t.id != 0} for a in env.get_live_agents()}
    known_count: Dict[int, int] = {}
    for known in visibility.values():
        for tid in known:
            known_count[tid] = known_count.get(tid, 0) + 1
    agents = []
    for a in env.agents_obj:
        head = a.tasks[0] if a.tasks else env.task_idle
        agents.append({"id": int(a.id), "name": a.name, "type": a.type, "position": [float(a.position[0]), float(a.position[1])],
                       "state": int(a.state), "task_id": int(head.id), "commit_until": int(getattr(a, "commit_until", 0) or 0),
                       "known_tasks": len(visibility.get(a.name, set()))})
    tasks = []
    for t in env.tasks:
        if t.id == 0:
            continue
        deadline = getattr(t, "hard_deadline", None)
        kind = getattr(t, "kind", None)
        prot = getattr(t, "protected_agent", None)
        tasks.append({"id": int(t.id), "type": t.type, "kind": kind, "position": [float(t.position[0]), float(t.position[1])],
                      "status": int(t.status), "created_at": int(getattr(t, "created_at", 0) or 0),
                      "deadline": None if deadline is None else int(deadline),
                      "required": float(t.currentReqs[t.typeIdx]), "allocated": float(t.allocatedReqs[t.typeIdx]),
                      "known_by": int(known_count.get(t.id, 0)), "is_dynamic": deadline is not None, "is_escort": kind == "Escort",
                      "required_agents": int(getattr(t, "required_agents", 0) or 0),
                      "assigned_agents": len(getattr(t, "allocationDetails", {}) or {}),
                      "protected_agent": None if prot is None else str(prot.name),
                      "protected_position": None if prot is None else [float(prot.position[0]), float(prot.position[1])]})
    threats = []
    for h in env.threats:
        mt, it = getattr(h, "mission_target_agent", None), getattr(h, "intercepting_agent", None)
        threats.append({"id": int(h.id), "position": [float(h.position[0]), float(h.position[1])], "status": int(h.status),
                        "group": int(h.threat_group), "threat_type": getattr(h, "threat_type", None),
                        "mission_target": None if mt is None else str(mt.name), "intercepting": None if it is None else str(it.name)})
    s_wps = float(env.compute_s_wps())
    return {
        "time": int(env.time_steps), "agents": agents, "tasks": tasks, "threats": threats,
        "events": [event_record(ev, env.time_steps) for ev in events],
        "decision": {"replanned": replanned, "new_commits": committed},
        "metrics": {
            "s_wps": s_wps, "s_esc": float(env.compute_s_esc()) if hasattr(env, "compute_s_esc") else s_wps,
            "on_time": int(env.n_on_time), "missed": int(env.n_missed_windows), "switches": int(env.n_task_switches),
            "distance": float(env.total_distance), "active_agents": sum(1 for a in env.agents_obj if a.state != -1),
            "open_tasks": sum(1 for t in env.tasks if t.id != 0 and t.status != 2),
            "escort_coverage": float(getattr(env, "escort_covered_steps", 0) / max(getattr(env, "escort_required_steps", 0), 1)),
            "recon_losses": int(getattr(env, "recon_losses", 0)), "protected_rec": int(getattr(env, "protected_rec_completed", 0)),
            "mutual_support": int(getattr(env, "mutual_support_engagements", 0)),
        },
    }


def infer_events(previous: dict, current: dict) -> List[dict]:
    """Reviewer events for state changes the env does not emit (:60-117)."""
    t = current["time"]
    out: List[dict] = []
    prev_agents = {a["name"]: a for a in previous["agents"]}
    prev_tasks = {(k["type"], int(k["id"])): k for k in previous["tasks"]}
    prev_threats = {h["id"] for h in previous["threats"]}
    for a in current["agents"]:
        old = prev_agents.get(a["name"])
        if old and old["state"] != -1 and a["state"] == -1:
            out.append({"time": t, "type": "Agent_Fail", "detail": [a["name"]]})
    for k in current["tasks"]:
        old = prev_tasks.get((k["type"], int(k["id"])))
        label = f"{k['type']}{k['id']}"
        if old is None:
            out.append({"time": t, "type": "Task_Arrival", "detail": [label, "left" if k["position"][0] < 600 else "right"]})
        elif old["status"] != 2 and k["status"] == 2:
            missed = k["deadline"] is not None and t > k["deadline"]
            out.append({"time": t, "type": "Window_Missed" if missed else "Task_Completed", "detail": [label]})
        if old and old["known_by"] == 0 and k["known_by"] > 0:
            out.append({"time": t, "type": "Task_Discovered", "detail": [label, f"by {k['known_by']} UAV(s)"]})
    for h in current["threats"]:
        if h["id"] not in prev_threats:
            out.append({"time": t, "type": "Threat_Spawn", "detail": [str(h["id"])]})
    for name in current["decision"]["new_commits"]:
        out.append({"time": t, "type": "Agent_Commit", "detail": [name]})
    if current["decision"]["replanned"]:
        out.append({"time": t, "type": "Replan", "detail": []})
    return out


def generate(seed: int, output: Optional[Path] = None, scenario: str = "WPS_escort", allocator: str = "urgency_coalition",
             interval: int = 15, env: Optional[MultiUAVEnv] = None, title: Optional[str] = None) -> Dict[str, Any]:
    """Run one episode with the on-device planner and return (and optionally write) the replay document (:225-306)."""
    spec = CASE_SPECS[scenario]
    if env is None:
        env = MultiUAVEnv(spec, flags=dict(WPS_ENV_FLAGS))
    env._b.set_allocator(allocator)
    _, info = env.reset(seed=seed)
    done = {a: False for a in env.agents}
    truncated = {a: False for a in env.agents}
    frames = [frame(env, [], False, [])]
    event_log: List[dict] = []
    p = env._params
    while not all(done.values()) and not all(truncated.values()):
        previous_events = list(info.get("events", [])) if isinstance(info, dict) else []
        actions, replanned = {}, False
        if allocator == "hungarian" or should_replan(env, previous_events, interval):
            aa, ai = env._b.allocate(interval if allocator != "hungarian" else interval, True)  # plan(): on the device
            actions = {env.agents_obj[int(a)].name: int(i) for a, i in zip(aa[0], ai[0]) if a >= 0}
            replanned = allocator != "hungarian" or bool(actions)
        _, _, done, truncated, info = env.step(actions)
        events = list(info.get("events", []))
        current = frame(env, events, replanned, [])
        inferred = infer_events(frames[-1], current)
        current["events"].extend(inferred)
        event_log.extend([event_record(ev, env.time_steps) for ev in events] + inferred)
        frames.append(current)
    replay = {
        "metadata": {
            "title": title or f"{scenario}: {ALGORITHM_LABEL[allocator]}", "scenario": scenario, "algorithm": ALGORITHM_LABEL[allocator],
            "seed": seed, "max_time_steps": int(p.max_time_steps), "area": [float(env.area_width), float(env.area_height)],
            "dynamics": {
                "arrival_rate": float(p.arrival_rate), "fail_rate": float(p.fail_rate), "sense_radius": float(p.sense_radius),
                "threat_delay": int(p.threat_delay), "hard_windows": bool(p.hard_windows), "window_length": int(p.window_length),
                "burst_mode": bool(p.burst_mode), "burst_size": int(p.burst_size), "dual_region_bursts": bool(p.dual_region_bursts),
                "share_knowledge": bool(p.share_knowledge), "commit_horizon": int(p.commit_horizon),
                "reassign_penalty": float(p.reassign_penalty), "escort_enabled": bool(p.escort_enabled),
                "escort_radius": float(p.escort_radius or 0.0),
            },
        },
        "events": event_log, "frames": frames, "final_metrics": frames[-1]["metrics"],
    }
    if output is not None:
        output = Path(output)
        output.parent.mkdir(parents=True, exist_ok=True)
        output.write_text(json.dumps(replay, indent=2), encoding="utf-8")
    return replay
