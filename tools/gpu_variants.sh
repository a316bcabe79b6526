#!/bin/bash
# Runs ON THE GPU BOX: headline bench for the shipped library and for each diagnostic build under tools/_build/ named on the command line
# (a trailing "+tiles" also runs the 24x48 / 64x128 configs for that build).
set -u
TAG=${1:-var}; shift
OUT=gpurun_out/$TAG
cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "default rc=$?"; cut -c1-160 "$OUT/bench_default.json"
for v in "$@"; do
  b=${v%+tiles}
  MUAVTA_SO=tools/_build/libmuavta_$b.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > "$OUT/bench_$b.json" 2> "$OUT/bench_$b.err"; echo "$b rc=$?"; cut -c1-160 "$OUT/bench_$b.json"
  if [ "$v" != "$b" ]; then
    for ce in WPS_escort24:4096 WPS_burst64:1024; do
      MUAVTA_SO=tools/_build/libmuavta_$b.so timeout -k 10 200 python bench.py --case ${ce%%:*} --envs ${ce##*:} --steps 5 --no-cpu-baseline --no-extras > "$OUT/bench_${b}_${ce%%:*}.json" 2> "$OUT/bench_${b}_${ce%%:*}.err"; cut -c1-160 "$OUT/bench_${b}_${ce%%:*}.json"
    done
  fi
done
