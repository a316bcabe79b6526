import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from muavta_amd.batched import BatchedMultiUAVEnv
from muavta_amd.params import params_from_config
from muavta_amd.scenarios import CASE_SPECS, WPS_ENV_FLAGS
base = dict(CASE_SPECS["WPS_hard_x2"])
def run(name, spec, obs=True, interval=20):
    p = params_from_config(spec, dict(WPS_ENV_FLAGS), tile_agents=16, tile_tasks=40, tile_threats=16)
    env = BatchedMultiUAVEnv(p, 4096)
    seeds = np.arange(4096, dtype=np.uint64)
    for _ in range(2): env.rollout(seeds, 150, interval, True, obs); env.sync()
    ms = []
    for _ in range(5):
        env.rollout(seeds, 150, interval, True, obs); ms.append(env.last_kernel_ms())
    print(f"{name:50s} {np.mean(ms):6.2f} ms  {4096*150/np.mean(ms)/1e3:7.1f} M env-steps/s", flush=True)
run("headline", base)
run("headline, no obs", base, obs=False)
s = dict(base); s["threats_list"] = []; run("no threats", s)
s = dict(base); s["arrival_rate"] = 0.0; run("no arrivals", s)
s = dict(base); s["fail_rate"] = 0.0; run("no failures", s)
s = dict(base); s["threats_list"] = []; s["arrival_rate"] = 0.0; s["fail_rate"] = 0.0; run("quiet (no threats/arrivals/failures)", s)
run("quiet, no obs", s, obs=False)
run("quiet, no obs, never replan (interval 1000)", s, obs=False, interval=1000)
s2 = dict(s); s2["sense_radius"] = 0.0; s2["threat_delay"] = 0; run("quiet, no sensing", s2, obs=False, interval=1000)
s3 = dict(s2); s3["tasks"] = {"Att": 0, "Rec": 1, "Hold": 0}; run("quiet, 1 task", s3, obs=False, interval=1000)
