"""Shared checker of the token-builder parity tests (oracle on CPU, HIP path on the GPU)."""
import numpy as np


def check_tokens(tok_of, g, k, where):
    """tok_of(kind, max_tasks, max_agents) -> dict; compared bit for bit with the reference's arrays at sample k."""
    p = tok_of(0, 32, 16)
    for name, key in (("task_feats", "p_tf"), ("task_mask", "p_tm"), ("task_ids", "p_tid"), ("agent_feats", "p_af"),
                      ("agent_mask", "p_am"), ("agent_ids", "p_aid"), ("edge_valid", "p_ev")):
        assert np.array_equal(np.asarray(p[name]).astype(g[key].dtype), g[key][k]), f"{where}: pair {name}"
    assert p["n_urgent"] == int(g["p_nurg"][k]), f"{where}: n_urgent"
    r = tok_of(1, 32, 16)
    for name, key in (("task_feats", "r_tf"), ("agent_feats", "r_af"), ("edge_valid", "r_ev")):
        assert np.array_equal(r[name], g[key][k]), f"{where}: raw {name}"
    e = tok_of(2, int(g["e_max_tasks"]), int(g["e_max_agents"]))
    for name, key in (("task_feats", "e_tf"), ("task_mask", "e_tm"), ("task_ids", "e_tid"), ("agent_feats", "e_af"),
                      ("agent_mask", "e_am"), ("agent_ids", "e_aid"), ("edge_valid", "e_ev")):
        got, want = np.asarray(e[name]).astype(g[key].dtype), g[key][k]
        assert np.array_equal(got, want), f"{where}: escort {name}: {np.argwhere(got != want)[:4].tolist()}"
