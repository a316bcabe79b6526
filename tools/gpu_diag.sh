#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): parity suite, headline bench, and an instruction-cache counter pass per tile (is the
# ~270 KB k_rollout body thrashing the 64 KB instruction cache two CUs share?).  Output under gpurun_out/<tag>/.
set -u
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout -k 10 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cut -c1-600 "$OUT/bench.json"
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
NAMES=$(grep -o 'SQC_ICACHE_[A-Z_]*\|SQC_INST_[A-Z_]*' "$OUT/avail.txt" | sort -u | tr '\n' ' ')
echo "icache counters: $NAMES"
for ce in WPS_hard_x2:4096 WPS_escort24:4096 WPS_burst64:1024; do
  c=${ce%%:*}; n=${ce##*:}
  ARGS="bench.py --case $c --envs $n --no-cpu-baseline --no-extras --steps 3 --warmup 1"
  set -- $NAMES
  while [ $# -gt 0 ]; do
    G="$1 ${2:-} ${3:-} ${4:-}"; shift; shift 2>/dev/null; shift 2>/dev/null; shift 2>/dev/null
    D="$OUT/ic_${c}_$(echo $G | tr ' ' '_' | cut -c1-60)"
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$D" -- python3 $ARGS > /dev/null 2> "$D.err"
    f=$(ls $D/*/*_counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" "$c" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_rollout" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: sum(v) / len(v) for k, v in acc.items()})
PY
    rm -rf "$D"
  done
done 2>&1 | tee "$OUT/icache.txt"
