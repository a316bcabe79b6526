#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): parity suite, headline bench, diagnostic builds, phase shares.  Output under gpurun_out/<tag>/.
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 420 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout -k 10 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"
for v in none; do
  if [ -f tools/_build/libmuavta_$v.so ]; then
    MUAVTA_SO=tools/_build/libmuavta_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"; echo "bench $v rc=$?"
    cat "$OUT/bench_$v.json"
  fi
done
if [ -f tools/_build/libmuavta_prof.so ]; then
  for c in "WPS_hard_x2 4096" "WPS_escort24 4096" "WPS_burst64 1024"; do
    timeout -k 10 120 python tools/phase_profile.py $c > "$OUT/phase_${c%% *}.txt" 2>&1; echo "phase $c rc=$?"
  done
fi
