#!/bin/bash
# Runs ON THE GPU BOX: two rocprofv3 counter passes over tools/valu_meter.py's workload, then its report.
set -u
TAG=${1:-v}; CASE=${2:-WPS_hard_x2}; N=${3:-4096}   # MUAVTA_SO selects a diagnostic build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/valu; mkdir -p $OUT
if [ "${4:-}" = "traffic" ]; then  # FETCH_SIZE / WRITE_SIZE need a pass each (TCC slots); per-kernel bytes against known byte counts
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_c -- python3 tools/valu_meter.py drive $TAG $CASE $N > $OUT/${TAG}_c.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_d -- python3 tools/valu_meter.py drive $TAG $CASE $N > $OUT/${TAG}_d.log 2>&1 &&
  python3 tools/valu_meter.py report $TAG $CASE $N > $OUT/$TAG.txt 2>&1
  cat $OUT/$TAG.txt
  rm -rf $OUT/${TAG}_c $OUT/${TAG}_d
  exit 0
fi
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_a -- python3 tools/valu_meter.py drive $TAG $CASE $N > $OUT/${TAG}_a.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/${TAG}_b -- python3 tools/valu_meter.py drive $TAG $CASE $N > $OUT/${TAG}_b.log 2>&1 &&
python3 tools/valu_meter.py report $TAG $CASE $N > $OUT/$TAG.txt 2>&1
cat $OUT/$TAG.txt
rm -rf $OUT/${TAG}_a $OUT/${TAG}_b
