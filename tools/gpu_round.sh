#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): parity suite, headline bench (with the other tiles), optional diagnostic builds and phase
# shares.  Output under gpurun_out/<tag>/.   usage: gpu_round.sh <tag> [notest] [nophase]
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
export TMPDIR=/tmp
if [[ " $* " != *" notest "* ]]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
  tail -5 "$OUT/pytest.log"
fi
timeout -k 10 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("headline %.1f M  kernel %.3f ms  frac %.3f | step_api %.1f fused_step %.1f wide %.1f obs_ring %.1f il %.1f" % (
        d["value"] / 1e6, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["step_api_env_steps_per_s"] / 1e6, d["fused_step_api_env_steps_per_s"] / 1e6,
        (d.get("fused_step_api_wide_env_steps_per_s") or 0) / 1e6, (d.get("obs_ring_env_steps_per_s") or 0) / 1e6, (d.get("il_samples_per_s") or 0) / 1e6))
    for k, v in d.get("other_tiles", {}).items():
        print("  %s %.2f M  kernel %.3f ms  frac %.3f" % (k, v["env_steps_per_s"] / 1e6, v["roofline"]["kernel_ms"], v["roofline"]["frac"]))
except Exception as e:
    print("bench parse failed:", e)
PY
if [[ " $* " != *" nophase "* ]] && [ -f tools/_build/libmuavta_prof.so ]; then
  for c in "WPS_hard_x2 4096" "WPS_escort24 4096" "WPS_burst64 1024"; do
    timeout -k 10 120 python tools/phase_profile.py $c > "$OUT/phase_${c%% *}.txt" 2>&1; echo "phase $c rc=$?"
  done
fi
