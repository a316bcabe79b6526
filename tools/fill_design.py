#!/usr/bin/env python3
"""Builder's tool: fill the @@PLACEHOLDER@@ figures of DESIGN.md's round-5 sections from a bench.py JSON line.  usage: fill_design.py bench.json < template > DESIGN.md"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = sys.stdin.read()
ot = d["other_tiles"]
M = lambda x: f"{x / 1e6:.1f}"
rep = {
    "H_VALUE": M(d["value"]), "H_ONE": M(d["value_one_lane"]), "H_ISO": f"{d['roofline']['isolated']['kernel_ms']:.2f}", "H_OVL": f"{d['roofline']['kernel_ms']:.2f}",
    "E_VALUE": M(ot["WPS_escort24"]["env_steps_per_s"]), "E_ONE": M(ot["WPS_escort24"]["one_lane"]["env_steps_per_s"]),
    "E_ISO": f"{ot['WPS_escort24']['roofline']['isolated']['kernel_ms']:.2f}", "E_OVL": f"{ot['WPS_escort24']['roofline']['kernel_ms']:.2f}",
    "B_VALUE": M(ot["WPS_burst64"]["env_steps_per_s"]), "B_ONE": M(ot["WPS_burst64"]["one_lane"]["env_steps_per_s"]),
    "B_ISO": f"{ot['WPS_burst64']['roofline']['isolated']['kernel_ms']:.2f}", "B_OVL": f"{ot['WPS_burst64']['roofline']['kernel_ms']:.2f}",
    "P_RUN": M(d["policy_in_loop_env_steps_per_s"]), "P_CALLS": f"{d['policy_calls_per_env_step']:.2f}", "P_GATE": M(d["policy_in_loop_to_the_gate"]["env_steps_per_s"]),
    "P_STEP": M(d["policy_in_loop_per_step_env_steps_per_s"]), "N_STEP": M(d["policy_in_loop_with_network"]["per_step"]["env_steps_per_s"]),
    "N_RUN": M(d["policy_in_loop_with_network"]["run_ahead"]["env_steps_per_s"]), "F_ONE": f"{d['facade_steps_per_s']:,.0f}", "F_BATCH": f"{d['facade_batch_env_steps_per_s']:,.0f}",
    "CPU": M(d["cpu_baseline"]["value"]),
}
for k, v in rep.items():
    t = t.replace(f"@@{k}@@", v)
sys.stdout.write(t)
