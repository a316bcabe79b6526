#!/usr/bin/env python3
"""Runs in a FRESH interpreter (spawned by tests/test_compat_cpu.py, this container only): installs the import aliases of
muavta_amd.compat over the CPU oracle backend, then imports the reference's own harness files UNCHANGED from a scratch copy of
the read-only checkout and runs them.  Prints one JSON document on the last line.

    compat_driver.py episodes <ref_copy>          run_wps_episode / run_escort_episode for a few (algorithm, case, seed)
    compat_driver.py test_escort <ref_copy>       experiments/test_escort.py, all seven tests, as `python test_escort.py` would
    compat_driver.py trainers <ref_copy>          experiments/train_pair_cost.py::main() (IL, then RL from the IL checkpoint; MLP and attention+context
                                                  scorers) — per-episode losses / returns, the evaluation scores and a hash of every trained tensor
    compat_driver.py trainers_native <ref_copy>   the same WITHOUT the aliases: the reference's own env (what the line above has to reproduce)
    compat_driver.py sweep[_native] <ref_copy>    experiments/wps_eval.py::run_wps_episode for EVERY algorithm it knows (19: Hungarian variants, CBBA, PI, capability
                                                  greedy, the RAH / commit / pair / context-pair hybrids with randomly initialised networks) and
                                                  escort_eval.py::run_escort_episode for every algorithm it knows — all result keys but the timings
    compat_driver.py scripts[_native] <ref_copy>  whole command lines, main() to CSV: train_escort.py (Att + MLP) -> escort_eval.py with those checkpoints;
                                                  wps_eval.py (default suite, six algorithms, per-episode CSV); run_scaling.py (8 generated cases x 4 algorithms);
                                                  benchmark.py of the checkout's root (Random / Greedy / CBBA); main.py's run_case_algorithm (Random / Greedy / Swarm-GAP / CBBA); a third
                                                  argument picks steps (escort,wps,scaling,paper,replay,benchmark,main); paper_eval.py --suite dynamic --env-flags d3
"""
import json
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
mode, ref = sys.argv[1], sys.argv[2]
os.environ["MUAVTA_REFERENCE"] = ref
import refshim  # noqa: E402  (stand-ins for gymnasium / pettingzoo / seaborn / tianshou, which this image lacks)

refshim.REF = ref
refshim.install()
from oracle_backend import OracleBackend  # noqa: E402
import muavta_amd.compat as compat  # noqa: E402

if mode.endswith("_native"):
    sys.path.append("/root/reference")  # mUAV_TA itself: the read-only checkout (nothing is written there: PYTHONDONTWRITEBYTECODE)
else:
    compat.install(backend_factory=OracleBackend)  # mUAV_TA.* and core_sim now resolve to this repository
    assert sys.modules["mUAV_TA.DroneEnv"].MultiUAVEnv is compat.MultiUAVEnv

if mode == "episodes":
    from experiments.wps_eval import run_wps_episode
    from experiments.escort_eval import run_escort_episode
    from TaskAllocation.Hybrid.AttentionEscort import UrgencyCoalition
    import experiments.wps_eval as W

    assert W.MultiUAVEnv is compat.MultiUAVEnv
    out = {"wps": [], "escort": []}
    for algo, case, seeds in (("Local-Hungarian", "WPS_hard", (0, 1, 2)), ("Local-Hungarian", "WPS_easy", (0,)), ("Urgency-Pair", "WPS_hard", (0, 1))):
        for s in seeds:
            r = run_wps_episode(algo, case, s)
            out["wps"].append({"algorithm": algo, "case": case, "seed": s, **{k: float(v) for k, v in r.items()}})
    for algo, case, seeds in (("Coalition-Hungarian", "WPS_escort", (0, 1)), ("Urgency-Coalition", "WPS_escort", (0,))):
        for s in seeds:
            r = run_escort_episode(algo, case, s, urg=UrgencyCoalition() if algo == "Urgency-Coalition" else None)
            out["escort"].append({k: (v if isinstance(v, str) else float(v)) for k, v in r.items()})
    print(json.dumps(out))
elif mode == "test_escort":
    runpy.run_path(os.path.join(ref, "experiments", "test_escort.py"), run_name="__main__")
    print(json.dumps({"ok": True}))
elif mode in ("trainers", "trainers_native"):
    import contextlib
    import hashlib
    import importlib
    import io
    import random
    import types

    import torch

    torch.set_num_threads(1)
    rec = {}

    def recording(fn, key):
        def wrapped(*a, **k):
            r = fn(*a, **k)
            rec.setdefault(key, []).append(repr(r))  # (floats: repr round-trips every bit)
            return r
        return wrapped

    def tensor_hash(obj, h):
        if isinstance(obj, torch.Tensor):
            h.update(obj.detach().cpu().contiguous().numpy().tobytes())
        elif isinstance(obj, dict):
            for k in sorted(obj, key=str):
                h.update(str(k).encode()); tensor_hash(obj[k], h)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                tensor_hash(v, h)
        elif isinstance(obj, (int, float, str, bool)) or obj is None:
            h.update(repr(obj).encode())

    pair = ["--case", "WPS_hard", "--d-model", "16", "--nhead", "2", "--n-layers", "1", "--il-warmup", "2", "--il-batch", "4", "--eval-eps", "2"]
    runs = (  # (name, the reference's trainer file, its command line, checkpoint to start from)
        ("pair_il_mlp", "train_pair_cost", ["--phase", "il", "--mlp", "--episodes", "3", "--eval-every", "3"] + pair, None),
        ("pair_rl_mlp", "train_pair_cost", ["--phase", "rl", "--mlp", "--episodes", "2", "--eval-every", "2"] + pair, "pair_il_mlp"),
        ("pair_il_att_context", "train_pair_cost", ["--phase", "il", "--context", "--episodes", "2", "--eval-every", "2"] + pair, None),
        ("escort_att", "train_escort", ["--case", "WPS_escort", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1", "--d-model", "16", "--nhead", "2", "--n-layers", "1"], None),
        ("att_rah", "train_att_rah", ["--case", "WPS_hard", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1"], None),
        ("att_commit", "train_att_commit", ["--case", "WPS_commit", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1"], None),
        ("rah", "train_rah", ["--case", "WPS_hard", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1"], None),
        ("hybrid_rg_dqn", "train_hybrid", ["--algo", "RG-DQN", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1"], None),
        ("hybrid_ra_dqn", "train_hybrid", ["--algo", "RA-DQN", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1"], None),
    )
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    out, ckpts = {}, {}
    for n_run, (name, modname, argv, init) in enumerate(runs):
        if only and name not in only:
            continue
        T = importlib.import_module(f"experiments.{modname}")
        assert (T.MultiUAVEnv is compat.MultiUAVEnv) == (mode == "trainers")
        for attr, fn in list(vars(T).items()):  # the episode runners and evaluation loops of the file, recorded (not changed)
            if isinstance(fn, types.FunctionType) and fn.__module__ == T.__name__ and (attr.startswith("run_") or attr.startswith("eval")) and not hasattr(fn, "_rec"):
                w = recording(fn, attr); w._rec = True
                setattr(T, attr, w)
        rec.clear()
        random.seed(20240 + n_run)  # an unseeded env.reset() draws its seed from the GLOBAL random module (DroneEnv.py:525-526); main() seeds numpy and torch only
        torch.manual_seed(777 + n_run)  # (train_hybrid.py seeds numpy only: its networks' initial weights come from torch's global generator)
        ckpts[name] = os.path.join(ref, f"ckpt_{mode}_{name}.pth")
        sys.argv = [modname + ".py"] + argv + ["--seed", "3", "--out", ckpts[name]] + (["--init", ckpts[init]] if init else [])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ret = T.main()
        h = hashlib.sha256()
        tensor_hash(torch.load(ckpts[name], map_location="cpu", weights_only=True), h)  # (a file this process wrote a moment ago)
        out[name] = dict(rec, main_returned=repr(ret).replace(ckpts[name], "<ckpt>"), checkpoint_sha256=h.hexdigest(),
                         printed=[ln.replace(ckpts[name], "<ckpt>") for ln in buf.getvalue().splitlines() if "EVAL" in ln.upper() or "Done" in ln][:6])
    print(json.dumps(out))
elif mode in ("sweep", "sweep_native"):
    import inspect
    import random

    import numpy as np
    import torch

    import experiments.escort_eval as E
    import experiments.wps_eval as W

    assert (W.MultiUAVEnv is compat.MultiUAVEnv) == (mode == "sweep") and (E.MultiUAVEnv is compat.MultiUAVEnv) == (mode == "sweep")
    torch.set_num_threads(1)
    torch.manual_seed(11); np.random.seed(11); random.seed(11)
    pol = dict(rah=W.ReserveAwareHybrid(), att_rah=W.AttentionRAH(), att_commit=W.AttentionCommit(use_attention=True), mlp_commit=W.AttentionCommit(use_attention=False),
               urg_commit=W.UrgencyCommit(), att_pair=W.PairCostHybrid(use_attention=True), mlp_pair=W.PairCostHybrid(use_attention=False), urg_pair=W.UrgencyPair(max_tasks=32, max_agents=16),
               att_ctx=W.ContextPairHybrid(use_attention=True), mlp_ctx=W.ContextPairHybrid(use_attention=False), gnn_ctx=W.GNNContextPairHybrid())
    for v in pol.values():  # (as wps_eval.main() sets them after loading a checkpoint)
        if hasattr(v, "eps"):
            v.eps = 0.0
    src = inspect.getsource(W.run_wps_episode)
    import re
    algos = sorted(set(re.findall(r'"([A-Za-z]+(?:-[A-Za-z]+)+)"', src)) - {"Reset-Allocation"})
    timing = lambda k: "ms" in k.lower() or "time_s" in k.lower() or k.lower().endswith("_sec")  # noqa: E731
    out = {"wps": {}, "escort": {}}
    cases = (("WPS_hard", 5), ("WPS_attn", 2), ("WPS_burst", 1), ("WPS_commit", 4), ("WPS_attn_L", 0))
    for i, algo in enumerate(algos):
        case, seed = cases[i % len(cases)]
        torch.manual_seed(100 + i); np.random.seed(100 + i); random.seed(100 + i)
        r = W.run_wps_episode(algo, case, seed, **pol)
        out["wps"][f"{algo}|{case}|{seed}"] = {k: (v if isinstance(v, str) else repr(float(v))) for k, v in r.items() if not timing(k)}
    esrc = inspect.getsource(E.run_escort_episode)
    ealgos = sorted(set(re.findall(r'"([A-Za-z]+(?:-[A-Za-z0-9]+)+)"', esrc)))
    torch.manual_seed(12)
    epol = dict(att=E.AttentionEscort(use_attention=True), mlp=E.AttentionEscort(use_attention=False), urg=E.UrgencyCoalition())
    for v in epol.values():
        if hasattr(v, "eps"):
            v.eps = 0.0
    ecases = (("WPS_escort", 1, 12), ("WPS_escort", 2, 12), ("WPS_escort", 3, 8))
    for i, algo in enumerate(ealgos):
        case, seed, interval = ecases[i % len(ecases)]
        torch.manual_seed(200 + i); np.random.seed(200 + i); random.seed(200 + i)
        r = E.run_escort_episode(algo, case, seed, replan_interval=interval, **epol)
        out["escort"][f"{algo}|{case}|{seed}|{interval}"] = {k: (v if isinstance(v, str) else repr(float(v))) for k, v in r.items() if not timing(k)}
    # the paper's static / dynamic table: experiments/paper_eval.py::run_episode, every algorithm but TBTA (a tianshou policy: out of scope) and the ILP oracle
    import experiments.paper_eval as PE
    from TaskAllocation.Hybrid.ReplanGate import ReplanGateAgent, ResidualAssignmentAgent

    assert (PE.MultiUAVEnv is compat.MultiUAVEnv) == (mode == "sweep")
    psrc = inspect.getsource(PE.run_episode)
    palgos = sorted(set(re.findall(r'algorithm == "([A-Za-z-]+)"', psrc)) - {"TBTA", "ILP-Oracle"})  # (ILP-Oracle needs `pulp`, which this image lacks)
    torch.manual_seed(13)
    hyb = {"RG-DQN": ReplanGateAgent(), "RA-DQN": ResidualAssignmentAgent()}
    for v in hyb.values():
        v.eps = 0.0
    pcases = (("static_strike", 2, PE.DEFAULT_ENV_FLAGS), ("D3_combined", 1, PE.TBTA_E3_FLAGS), ("recon_strike_mix", 0, PE.TBTA_E3_FLAGS), ("D1_attrition", 3, PE.DEFAULT_ENV_FLAGS),
              ("D2_popup_threats", 4, PE.TBTA_E3_FLAGS), ("agent_scaling_mid", 5, PE.DEFAULT_ENV_FLAGS))
    out["paper"] = {}
    for i, algo in enumerate(palgos):
        for j in range(2):
            case, seed, fl = pcases[(2 * i + j) % len(pcases)]
            torch.manual_seed(300 + i); np.random.seed(300 + i); random.seed(300 + i)
            r = PE.run_episode(algo, case, seed, dict(fl), hybrid_agent=hyb.get(algo))
            out["paper"][f"{algo}|{case}|{seed}"] = {k: (v if isinstance(v, str) else repr(float(v))) for k, v in r.items() if not timing(k)}
    out["paper_algorithms"] = palgos
    out["escort_algorithms"], out["wps_algorithms"] = ealgos, algos
    print(json.dumps(out))
elif mode in ("scripts", "scripts_native"):
    import contextlib
    import csv
    import importlib
    import io
    import random

    import numpy as np
    import torch

    torch.set_num_threads(1)
    work = os.path.join(ref, f"out_{mode}")
    os.makedirs(work, exist_ok=True)
    timing = lambda k: "ms" in k.lower() or k.lower() in ("seconds", "time_s") or "wall" in k.lower()  # noqa: E731  (wall-clock columns)

    def run_main(modname, argv, seed):
        M = importlib.import_module(f"experiments.{modname}")
        assert (M.MultiUAVEnv is compat.MultiUAVEnv) == (mode == "scripts") if hasattr(M, "MultiUAVEnv") else True
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        sys.argv = [modname + ".py"] + argv
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            M.main()
        return M, buf.getvalue()

    def rows_of(path):
        with open(path, newline="", encoding="utf-8") as f:
            return [{k: v for k, v in r.items() if not timing(k)} for r in csv.DictReader(f)]

    out = {}
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    want = lambda step: only is None or step in only  # noqa: E731
    if want("escort"):
        att, mlp = os.path.join(work, "att.pth"), os.path.join(work, "mlp.pth")
        run_main("train_escort", ["--episodes", "2", "--eval-every", "2", "--eval-eps", "1", "--out", att], 41)
        run_main("train_escort", ["--mlp", "--episodes", "2", "--eval-every", "2", "--eval-eps", "1", "--out", mlp], 42)
        E, printed = run_main("escort_eval", ["--episodes", "1", "--seed0", "5", "--att-ckpt", att, "--mlp-ckpt", mlp, "--tag", mode], 43)
        out["escort_eval_csv"] = rows_of(os.path.join(E.RESULTS, f"WPS_escort_escort_eval_{mode}.csv"))
    if want("wps"):
        wcsv, wep = os.path.join(work, "wps.csv"), os.path.join(work, "wps_episodes.csv")
        run_main("wps_eval", ["--episodes", "2", "--algorithms", "Local-Cap-Greedy,Local-Hungarian,Local-PI,Global-Hungarian,Urgency-Pair,Urgency-Commit", "--out", wcsv, "--episodes-out", wep], 44)
        out["wps_eval_csv"], out["wps_eval_episodes_csv"] = rows_of(wcsv), rows_of(wep)
    if want("scaling"):
        scsv = os.path.join(work, "scaling.csv")
        run_main("run_scaling", ["--episodes", "1", "--out", scsv], 45)
        out["run_scaling_csv"] = rows_of(scsv)
    if want("paper"):
        # the paper's table generator as a command line: the dynamic suite (D1_attrition, D2_popup_threats, D3_combined) under its `d3` flag preset
        # (time windows in the observation, the dynamic idle penalty, its own reward weights) x five allocators
        pcsv = os.path.join(work, "paper.csv")
        run_main("paper_eval", ["--suite", "dynamic", "--episodes", "2", "--env-flags", "d3", "--algorithms", "Random,Greedy,Cap-Greedy,CBBA-Replan,Hungarian", "--out", pcsv], 48)
        out["paper_eval_csv"] = rows_of(pcsv)
    if want("replay"):
        # the reference's dashboard exporter ITSELF (not muavta_amd/replay.py, its restatement) over the env: every frame's agents / tasks / threats / escort
        # links / metrics and the event list of two scenarios, as the JSON document it writes
        import hashlib

        out["replay_json"] = {}
        for scen, seed in (("WPS_escort", 0), ("WPS_commit", 1)):
            rp = os.path.join(work, f"{scen}_replay.json")
            run_main("generate_simulation_replay", ["--seed", str(seed), "--scenario", scen, "--out", rp], 49)
            doc = json.load(open(rp, encoding="utf-8"))
            out["replay_json"][scen] = {"frames": len(doc["frames"]), "events": len(doc["events"]), "event_types": sorted({e["type"] for e in doc["events"]}),
                                        "sha256": hashlib.sha256(json.dumps(doc, sort_keys=True).encode()).hexdigest(), "last_frame": doc["frames"][-1]}
    if want("benchmark"):
        # the reference's own throughput script (benchmark.py at the root of the checkout: Random / Greedy / CBBA, list-valued actions, fixed_seed = 42,
        # get_initial_state, current_agent): run by path, its per-episode reward / completion printout kept, its SPS and timing dropped
        import re
        import runpy

        os.environ["MPLBACKEND"] = "Agg"
        os.chdir(work)  # (it saves benchmark_results.png into the working directory)
        random.seed(46); np.random.seed(46)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            runpy.run_path("/root/reference/benchmark.py", run_name="__main__")
        algo, rows = None, []
        for ln in buf.getvalue().splitlines():
            m = re.match(r"Running (\S+) Benchmark", ln)
            if m:
                algo = m.group(1)
            m = re.match(r"Ep (\d+)/(\d+) \| Time: .* \| SPS: .* \| Reward: (\S+) \| Completed: (\S+)", ln)
            if m:
                rows.append({"algorithm": algo, "episode": m.group(1), "reward": m.group(3), "completed": m.group(4)})
        out["benchmark_py"] = rows
    if want("main"):
        # main.py at the root of the checkout (the legacy entry point): run_case_algorithm for the four allocators that need no tianshou policy — Random,
        # Greedy, Swarm-GAP, CBBA — on one of its own fleet-scaling cases.  It builds the env with multiple_tasks_per_agent=False and assigns True on the env
        # object after every reset (main.py:130-141), calls get_initial_state() and close(); imported as a module, so its process pool does not start
        import importlib.util

        os.environ["MPLBACKEND"] = "Agg"
        spec = importlib.util.spec_from_file_location("reference_main", "/root/reference/main.py")
        M = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(M)
        assert (M.MultiUAVEnv is compat.MultiUAVEnv) == (mode == "scripts")
        case = {"case": 3, "F1": 1, "R1": 3, "F2": 0, "R2": 0, "Att": 6, "Rec": 24}  # (main.py:328-332, i = 3)
        keep = ("n_agents", "n_tasks", "mean_S_reward", "mean_R_reward", "time_reward", "distance_reward", "quality_reward", "losses", "kills", "process_runs")
        out["main_py"] = {}
        for algo in ("Random", "Greedy", "Swarm-GAP", "CBBA"):
            random.seed(47); np.random.seed(47)
            with contextlib.redirect_stdout(io.StringIO()):
                r = M.run_case_algorithm(case, algo, 2, 1, 0.0, -1, False, "Agents")
            out["main_py"][algo] = {k: r[k] for k in keep}
    print(json.dumps(out))
