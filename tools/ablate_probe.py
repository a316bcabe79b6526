#!/usr/bin/env python3
"""Builder's probe: true cost of the fixed phases of a step.  Runs the QUIET workload (config 2 with nothing happening: the ablated
phases have nothing to do, so leaving them out does not change what the other phases see) on libraries built with
-DMUAVTA_DIAGNOSTIC_BUILD -DMUAVTA_ABLATE=<1 << bit> (tools/_build/libmuavta_abl<bit>.so; csrc/muavta_diag.h) and on the shipped one; prints kernel ms per variant.
    usage: MUAVTA_SO=... python tools/ablate_probe.py   (one process per library: tools/ablate_probe.sh)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_from_config
from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS
spec = dict(CASE_SPECS["WPS_hard_x2"])
spec.update(threats_list=[], arrival_rate=0.0, fail_rate=0.0, sense_radius=0.0, threat_delay=0, tasks={"Att": 0, "Rec": 1, "Hold": 0})
p = params_from_config(spec, dict(WPS_ENV_FLAGS), tile_agents=16, tile_tasks=40, tile_threats=16)
env = BatchedMultiUAVEnv(p, 4096)
seeds = np.arange(4096, dtype=np.uint64)
for _ in range(3):
    env.rollout(seeds, 150, 1000, True, False); env.sync()
ms = []
for _ in range(8):
    env.rollout(seeds, 150, 1000, True, False); ms.append(env.last_kernel_ms())
print(f"{os.path.basename(os.environ.get('MUAVTA_SO', 'shipped')):24s} quiet, no obs: {np.mean(ms):.4f} ms", flush=True)
