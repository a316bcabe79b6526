"""Case name -> MuavtaParams for the parity tests: registry cases (muavta_amd.scenarios) and the fuzzed configurations
whose reference traces are committed as tests/golden/trace_FUZZ*.npz (configs: tests/golden/fuzz_configs.json) and
tests/golden/trace_WIDE*.npz (configs the wide fuzz found device bugs on: tests/golden/wide_configs.json, tests/fuzz_reference.py --pin)."""
import json
import os

from muavta_amd.params import params_for_case, params_from_config

_FUZZ = None


def fuzz_configs():
    global _FUZZ
    if _FUZZ is None:
        _FUZZ = {}
        for name in ("fuzz_configs.json", "wide_configs.json"):
            path = os.path.join(os.path.dirname(__file__), "golden", name)
            if os.path.exists(path):
                with open(path) as f:
                    _FUZZ.update(json.load(f))
    return _FUZZ


def params_of(case: str, **tiles):
    if case.startswith("FUZZ") or case.startswith("WIDE"):
        cfg = dict(fuzz_configs()[case])
        cfg["threats_list"] = [tuple(x) for x in cfg["threats_list"]]
        cfg["escort_agent_types"] = tuple(cfg["escort_agent_types"])
        return params_from_config(cfg, None, tile_agents=tiles.get("tile_agents", 16), tile_tasks=tiles.get("tile_tasks", 128),
                                  tile_threats=tiles.get("tile_threats", 16))
    return params_for_case(case, **tiles)
