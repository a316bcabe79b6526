#!/bin/bash
# Runs ON THE GPU BOX: same-box comparison of up to three builds tools/_build/libmuavta_{A,B,C}.so, alternating, bench.py --no-extras
CASE=${1:-WPS_hard_x2}; ENVS=${2:-4096}; ROUNDS=${3:-3}; VARS=${4:-"A B C"}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    MUAVTA_SO=$PWD/tools/_build/libmuavta_$v.so timeout -k 10 200 python bench.py --case $CASE --envs $ENVS --steps 40 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $CASE', round(d['value']/1e6,2), 'M  kernel', round(d['roofline']['kernel_ms'],4), 'ms')"
  done
done
