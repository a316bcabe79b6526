"""Drop-in proof, this container only: the reference's OWN harness files — experiments/wps_eval.py::
run_wps_episode, experiments/escort_eval.py::run_escort_episode and experiments/test_escort.py — run unchanged over
`muavta_amd.compat` (the facade + the oracle backend; no GPU here) and reproduce the metrics the reference env produced
itself (tests/golden/*metrics*).  A scratch copy of the checkout is used because test_escort.py writes a checkpoint next
to itself and /root/reference is read-only; nothing of it is kept."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from muavta_amd.params import METRIC_KEYS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mUAV_TA")), reason="reference checkout not present (GPU box)")


@pytest.fixture(scope="module")
def ref_copy(tmp_path_factory):
    dst = tmp_path_factory.mktemp("ref")
    for d in ("experiments", "TaskAllocation"):
        shutil.copytree(os.path.join(REF, d), os.path.join(dst, d), ignore=shutil.ignore_patterns("results", "__pycache__", "*.pth", "*.csv"))
    yield str(dst)
    shutil.rmtree(dst, ignore_errors=True)


def _env():
    return dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="2", PYTHONHASHSEED="0")  # (hash seed: the reference's CBBA iterates over sets of names -
    # its results move from run to run on the reference env itself unless string hashing is pinned)


def _drive(mode, ref_copy, *more):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "compat_driver.py"), mode, ref_copy, *more], capture_output=True, text=True, timeout=1500, env=_env(), cwd=ref_copy)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stdout


def _drive_pair(mode, ref_copy):
    """`mode`_native (the reference's own env) and `mode` (the aliases installed) side by side in two fresh interpreters -> their JSON documents"""
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "compat_driver.py"), m, ref_copy], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env(), cwd=ref_copy)
             for m in (mode + "_native", mode)]
    docs = []
    for pr in procs:
        so, se = pr.communicate(timeout=1500)
        assert pr.returncode == 0, so[-3000:] + "\n" + se[-3000:]
        docs.append(json.loads(so.strip().splitlines()[-1]))
    return docs


def test_reference_episode_runners_run_unchanged_and_reproduce_the_reference_metrics(ref_copy):
    out, _ = _drive("episodes", ref_copy)
    K = {k: i for i, k in enumerate(METRIC_KEYS)}
    files = {("Local-Hungarian", "WPS_hard"): "metrics_WPS_hard.npz", ("Local-Hungarian", "WPS_easy"): "metrics_WPS_easy.npz",
             ("Urgency-Pair", "WPS_hard"): "urgpair_metrics_WPS_hard.npz", ("Coalition-Hungarian", "WPS_escort"): "metrics_WPS_escort.npz",
             ("Urgency-Coalition", "WPS_escort"): "urgcoal_metrics_WPS_escort.npz"}
    assert len(out["wps"]) == 6 and len(out["escort"]) == 3
    for r in out["wps"]:
        g = np.load(os.path.join(GOLDEN, files[(r["algorithm"], r["case"])]))
        want = g["metrics"][int(r["seed"])]
        for key in ("F_Reward", "S_WPS", "on_time_rate", "n_missed_windows", "n_on_time", "n_windowed_tasks", "reserve_idle_fraction",
                    "makespan", "total_distance", "n_task_switches"):
            assert r[key] == want[K[key]], (r["algorithm"], r["case"], r["seed"], key, r[key], want[K[key]])
        assert r["max_coord"] == 1200.0
        if r["algorithm"] == "Local-Hungarian":  # (the Urgency-* runners count their own plan() calls)
            assert r["algo_replans"] == float(g["n_replans"][int(r["seed"])])
    for r in out["escort"]:
        g = np.load(os.path.join(GOLDEN, files[(r["algorithm"], r["case"])]))
        want = g["metrics"][int(r["seed"])]
        for key in ("S_ESC", "S_WPS", "escort_coverage_rate", "protected_rec_completed", "recon_losses", "escort_losses",
                    "threats_intercepted", "mutual_support_engagements", "n_missed_windows", "on_time_rate", "n_task_switches"):
            assert r[key] == want[K[key]], (r["algorithm"], r["seed"], key, r[key], want[K[key]])
        if r["algorithm"] == "Coalition-Hungarian":
            assert r["n_replans"] == float(g["n_replans"][int(r["seed"])])


def test_reference_test_escort_passes_7_of_7(ref_copy):
    out, stdout = _drive("test_escort", ref_copy)
    assert out == {"ok": True}
    for name in ("unique_task_ids", "escort_lifecycle", "coalition_hungarian", "threat_diversion_inputs", "cbba_pi_coalition",
                 "pi_schedule_impact", "att_coalition_v2"):
        assert f"OK {name}" in stdout, stdout[-2000:]
    assert "ALL PASSED" in stdout


def test_reference_trainers_run_unchanged_and_train_the_same_checkpoints(ref_copy):
    """SURVEY §8 f3 at the reference's own boundary: every trainer file of the reference that drives the env directly — experiments/
    train_pair_cost.py (IL, RL from the IL checkpoint, MLP and attention + context scorers), train_escort.py, train_att_rah.py,
    train_att_commit.py, train_rah.py, train_hybrid.py (RG-DQN, RA-DQN on D3_combined: agent failures, arrivals, the TBTA flags) — has its
    main() called UNCHANGED twice in fresh interpreters: over the reference's env, and over `muavta_amd.compat` (facade + oracle backend).
    Same global seeds in front of both; episodes are reset UNSEEDED, as the trainers do (the env draws its seed from the global `random`
    module, DroneEnv.py:525-526).  Every episode's return / loss tuple, every evaluation score, main()'s return value and the SHA-256 over
    every tensor of the checkpoint it saved must be equal: the planners' state builders and token builders read the facade's object views
    and echoed configuration (`burst_mode`, ... — DroneEnv.py:101-201) exactly as they read the reference env's."""
    native, facade = _drive_pair("trainers", ref_copy)
    assert set(native) == set(facade) and len(native) == 9
    for name in native:
        for key in native[name]:
            assert native[name][key] == facade[name][key], (name, key, native[name][key], facade[name][key])
        assert len(native[name]["checkpoint_sha256"]) == 64 and native[name]["printed"], name
    assert len({r["checkpoint_sha256"] for r in native.values()}) == 9  # (nine different trainings, not one constant)


def test_every_algorithm_of_the_reference_eval_harnesses_gives_the_reference_result(ref_copy):
    """experiments/wps_eval.py::run_wps_episode for each of the 19 algorithms it knows (Global / Local Hungarian, CBBA-replan, PI, capability greedy,
    the RAH / Att-RAH ablations, Att / MLP / Urgency commit, Att / MLP / Urgency pair, Att / MLP / GNN context-pair — the learned ones with
    randomly initialised networks of the same torch seed) over WPS_hard / WPS_attn / WPS_burst / WPS_commit / WPS_attn_L, and escort_eval.py::
    run_escort_episode for each of its 7 (Global-Coalition, Coalition-Hungarian, CBBA / PI coalition, Urgency / MLP / Att coalition) at two
    replan intervals, and paper_eval.py::run_episode — the paper's static / dynamic table — for 9 of its 11 (Random, Greedy, Cap-Greedy, Swarm-GAP,
    CBBA, CBBA-Replan, Hungarian, RG-DQN, RA-DQN; not tianshou's TBTA, not the ILP oracle: `pulp` is absent) on two of static_strike / recon_strike_mix /
    agent_scaling_mid / D1_attrition / D2_popup_threats / D3_combined each (agent failures, pop-up threats, arrivals; S_Reward sums every step's reward
    dict): every result key but the wall-clock ones must equal what the same call returns over the reference's own env.  The
    allocators other than the Hungarian (not accelerated) read the facade's Task / UAV / threat views and mutate nothing the
    device does not know about — this is the test that they keep working when a user switches the env import."""
    native, facade = _drive_pair("sweep", ref_copy)
    assert len(native["wps_algorithms"]) == 19 and len(native["escort_algorithms"]) == 7
    assert len(native["wps"]) == 19 and len(native["escort"]) == 7 and len(native["paper_algorithms"]) == 9 and len(native["paper"]) == 18
    for grp in ("wps", "escort", "paper"):
        assert set(native[grp]) == set(facade[grp])
        for k, want in native[grp].items():
            assert len(want) >= 11 and facade[grp][k] == want, (k, {kk: (want[kk], facade[grp][k].get(kk)) for kk in want if want[kk] != facade[grp][k].get(kk)})
    assert len({r["S_WPS"] for r in native["wps"].values()}) >= 15  # (different planners, different episodes)


def test_reference_command_lines_write_the_same_csv_files(ref_copy):
    """Whole command lines of the reference, main() to CSV, unchanged: train_escort.py (Att-Coalition, then --mlp) -> escort_eval.py with the two
    checkpoints just trained (7 algorithms); wps_eval.py on its default suite (WPS_easy, WPS_hard) with six algorithms incl. the summary's bootstrap
    confidence intervals and the per-episode CSV; run_scaling.py (8 generated fleet / task scaling cases x Cap-Greedy, CBBA, CBBA-Replan, Hungarian
    through paper_eval.evaluate_case); paper_eval.py itself on its dynamic suite (D1_attrition, D2_popup_threats, D3_combined: agent failures, pop-up threats,
    arrivals) under its `d3` flag preset x Random, Greedy, Cap-Greedy, CBBA-Replan, Hungarian; generate_simulation_replay.py — the reference's dashboard exporter
    itself, not this repository's restatement of it — on WPS_escort and WPS_commit (the whole JSON document: 151 frames of agents / tasks / threats / escort links /
    metrics and the event list, compared through its SHA-256 and its last frame); benchmark.py at the root of the checkout (Random / Greedy / CBBA x 5 episodes, list-valued actions,
    `fixed_seed`, `get_initial_state`, `current_agent`: its per-episode reward / completion printout) and `run_case_algorithm` of main.py (the legacy entry
    point: Random / Greedy / Swarm-GAP / CBBA on one of its fleet-scaling cases; it assigns `multiple_tasks_per_agent = True` on the env after every reset and
    calls `close()`): every per-episode reward list it returns.  Every cell of every CSV but the wall-clock
    columns must equal the file the same command writes over the reference's own env."""
    native, facade = _drive_pair("scripts", ref_copy)
    assert {k: len(v) for k, v in native.items()} == {"escort_eval_csv": 7, "wps_eval_csv": 12, "wps_eval_episodes_csv": 24, "run_scaling_csv": 32, "paper_eval_csv": 15, "replay_json": 2, "benchmark_py": 15, "main_py": 4}
    rep_native, rep_facade = native.pop("replay_json"), facade.pop("replay_json")
    assert rep_facade == rep_native and all(r["frames"] == 151 and r["events"] > 100 and len(r["sha256"]) == 64 for r in rep_native.values())
    assert "Escort_Created" in rep_native["WPS_escort"]["event_types"] and rep_native["WPS_escort"]["sha256"] != rep_native["WPS_commit"]["sha256"]
    main_native, main_facade = native.pop("main_py"), facade.pop("main_py")
    assert set(main_native) == {"Random", "Greedy", "Swarm-GAP", "CBBA"} and main_facade == main_native
    assert len({repr(r["mean_S_reward"]) for r in main_native.values()}) == 4 and all(len(r["mean_S_reward"]) == 2 for r in main_native.values())
    for name, rows in native.items():
        assert len(facade[name]) == len(rows)
        for i, (want, got) in enumerate(zip(rows, facade[name])):
            assert got == want, (name, i, {k: (want[k], got.get(k)) for k in want if want[k] != got.get(k)})
