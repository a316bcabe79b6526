#!/usr/bin/env python3
"""Runs in a FRESH interpreter (spawned by tests/test_compat_cpu.py, this container only): installs the import aliases of
muavta_amd.compat over the CPU oracle backend, then imports the reference's own harness files UNCHANGED from a scratch copy of
the read-only checkout and runs them.  Prints one JSON document on the last line.

    compat_driver.py episodes <ref_copy>          run_wps_episode / run_escort_episode for a few (algorithm, case, seed)
    compat_driver.py test_escort <ref_copy>       experiments/test_escort.py, all seven tests, as `python test_escort.py` would
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
