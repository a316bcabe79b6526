"""MuavtaParams of a wide_config draw on a given tile (shared by the device fuzz and the N > 1 workers; no GPU, no oracle needed)."""
from muavta_amd.params import params_from_config


def params_of_wide(cfg, tile):
    c = dict(cfg)
    c["threats_list"] = [tuple(x) for x in c["threats_list"]]
    c["escort_agent_types"] = tuple(c["escort_agent_types"])
    return params_from_config(c, None, tile_agents=tile[0], tile_tasks=tile[1], tile_threats=tile[2])
