#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's closing evidence — parity suite, full bench line, phase shares with event
# counts, the seeding / sub-batch probes.  Output under gpurun_out/<tag>/ (copied into profiles/ by the builder).
set -u
TAG=${1:-r04final}
OUT=gpurun_out/$TAG
cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
timeout -k 10 400 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cut -c1-200 "$OUT/bench_full.json"
if [ -f tools/_build/libmuavta_prof.so ]; then
  for c in "WPS_hard_x2 4096" "WPS_escort24 4096" "WPS_burst64 1024"; do
    timeout -k 10 120 python tools/phase_profile.py $c > "$OUT/phase_${c%% *}.txt" 2>&1; echo "phase $c rc=$?"
  done
fi
timeout -k 10 120 python tools/seed_probe.py WPS_hard_x2 4096 > "$OUT/seed_probe.txt" 2>&1
{
  echo "# tools/parts_probe.py: 150 env steps of 4096 envs (config 2), one k_rollout(1 step) launch per sub-batch and step"
  echo "== default (4 hardware queues)"; timeout -k 10 120 python tools/parts_probe.py 2>&1 | grep parts
  echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python tools/parts_probe.py 2>&1 | grep parts
  echo "== GPU_MAX_HW_QUEUES=8, torch initialised in the process"; GPU_MAX_HW_QUEUES=8 PROBE_TORCH=1 timeout -k 10 120 python tools/parts_probe.py 2>&1 | grep parts
} > "$OUT/parts_probe.txt"
cat "$OUT/parts_probe.txt"
