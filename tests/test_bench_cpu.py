"""bench.py's bookkeeping that needs no GPU: the roofline block of the JSON line (labels as VERDICT r4 item 6 asked) is well formed for the three
tiles, `frac` is exactly algorithmic bytes per launch / the kernel's launch duration / HBM peak, and the PMC-derived fields (measured HBM traffic, VALU-port
occupancy, lane utilisation) appear only when profiles/pmc_traffic.json was collected on exactly the kernel sources of this tree (source hash) — a figure
taken on other sources must never be attached to a new build's line."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

TILES = (("WPS_hard_x2", 4096, 16, 32, 16), ("WPS_escort24", 4096, 24, 48, 24), ("WPS_burst64", 1024, 64, 128, 64))


@pytest.mark.parametrize("case,envs,tile,max_tasks,n_agents", TILES)
def test_roofline_block_is_well_formed_and_follows_8d(case, envs, tile, max_tasks, n_agents):
    kernel_ms, iso_ms, per_step = 4.5, 2.5, 2.4
    r = bench.roofline(case, envs, tile, kernel_ms, iso_ms, per_step, 2, max_tasks, n_agents)
    json.dumps(r)  # (goes into the one JSON line)
    B = bench.ALGO_BYTES_PER_ENV_STEP[tile]
    assert r["bound"] == "issue" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS == 8000.0 and r["kernel"] == "k_rollout"
    assert r["achieved"] == pytest.approx(envs * bench.HORIZON * B / (kernel_ms * 1e-3) / 1e9, rel=1e-12) and r["frac"] == pytest.approx(r["achieved"] / 8000.0, rel=1e-12)
    assert r["isolated"]["frac"] == pytest.approx(envs * bench.HORIZON * B / (iso_ms * 1e-3) / 1e9 / 8000.0, rel=1e-12)
    assert r["device_frac"] == pytest.approx(envs * bench.HORIZON * B / (per_step * 1e-3) / 1e9 / 8000.0, rel=1e-12) and r["launches_in_flight"] == 2
    floor = max_tasks * 84 + max_tasks + n_agents * (36 + 4 * ((max_tasks + 31) // 32)) + 30 + 4 * n_agents
    assert r["hbm_floor_bytes_per_env_step"] == floor < B
    e = bench.pmc_entry(case, envs)
    if e is None:  # the PMC passes were taken on other kernel sources: nothing measured may be attached
        assert r["traffic"] is None and r["issue_frac"] is None and r["lane_util"] is None and r["hbm_measured_frac"] is None and r["issue"] is None
    else:
        assert e["source_hash"] == bench.source_hash() == r["issue"]["source_hash"]
        assert r["traffic"] == e["bytes_per_launch"] > 0 and 0.0 < r["issue_frac"] <= 1.0 and 0.0 < r["lane_util"] <= 1.0
        assert r["hbm_measured_frac"] == pytest.approx(e["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9 / 8000.0, rel=1e-12)
        assert os.path.isfile(os.path.join(ROOT, "profiles", e["profile"]))  # the summary the figure comes from is committed


def test_pmc_figures_of_other_kernel_sources_are_not_attached(monkeypatch):
    monkeypatch.setattr(bench, "source_hash", lambda: "0" * 16)
    for case, envs, tile, mt, na in TILES:
        assert bench.pmc_entry(case, envs) is None and bench.measured_traffic(case, envs) is None
        r = bench.roofline(case, envs, tile, 3.0, None, None, 1, mt, na)
        assert r["traffic"] is None and r["issue"] is None and r["frac"] is not None and "isolated" not in r and "device_frac" not in r
