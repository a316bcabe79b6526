#!/bin/bash
# Runs ON THE GPU BOX: A/B of two builds of the library on ONE box (box-to-box spread is +-1.5 %): tools/_build/libmuavta_A.so against
# libmuavta_B.so, alternating, bench.py --no-extras.  usage: ab_bench.sh [case envs] [rounds]
CASE=${1:-WPS_hard_x2}; ENVS=${2:-4096}; ROUNDS=${3:-3}
for r in $(seq 1 $ROUNDS); do
  for v in A B; do
    MUAVTA_SO=$PWD/tools/_build/libmuavta_$v.so timeout -k 10 200 python bench.py --case $CASE --envs $ENVS --steps 40 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v $CASE', round(d['value']/1e6,2), 'M  kernel', round(d['roofline']['kernel_ms'],4), 'ms')"
  done
done
