#!/usr/bin/env python3
"""cProfile of the drop-in facade's step loop on the HIP backend (bench.py's facade_figures loop: the harness reads, the device plan read back as
[(name, Task)], env.step(dict)): where a facade step's time goes on the host.  usage: python tools/facade_profile.py [case]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muavta_amd.env import MultiUAVEnv  # noqa: E402
from muavta_amd.scenarios import CASE_SPECS, TILES, WPS_ENV_FLAGS  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "WPS_hard"
ta, tt, th = TILES[case]
env = MultiUAVEnv(CASE_SPECS[case], flags=dict(WPS_ENV_FLAGS), tile_agents=ta, tile_tasks=tt, tile_threats=th)


def run(seeds):
    n = 0
    for seed in seeds:
        env.reset(seed=seed)
        done = False
        while not done:
            env.get_live_agents()
            [t for t in env.tasks if t.id != 0 and t.status != 2]
            env.agent_visibility_map()
            aa, ai = env._b.allocate(20, True)
            pairs = [(env.agents_obj[int(a)].name, env.last_tasks_info[int(i)]) for a, i in zip(aa[0], ai[0]) if a >= 0]
            actions = {}
            for name, task in pairs:
                if env.last_tasks_info and task in env.last_tasks_info:
                    actions[name] = env.last_tasks_info.index(task)
            _, _, term, trunc, _ = env.step(actions)
            n += 1
            done = all(term.values()) or all(trunc.values())
    return n


run([0])
t0 = time.perf_counter()
n = run(range(1, 9))
print(f"{n / (time.perf_counter() - t0):.0f} steps/s")
pr = cProfile.Profile()
pr.enable()
n = run(range(9, 13))
pr.disable()
print(n, "steps profiled")
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
