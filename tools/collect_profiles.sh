#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): for each BASELINE config's tile, a rocprofv3 kernel trace and the PMC passes the HBM
# guide prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass), each with --kernel-trace only and the program directly
# after `--`.  Output under gpurun_out/prof_<tag>/<case>/; tools/summarize_profiles.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r04}
CASES=${2:-"WPS_hard_x2:4096 WPS_escort24:4096 WPS_burst64:1024"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ce in $CASES; do
  c=${ce%%:*}; n=${ce##*:}
  OUT=gpurun_out/prof_$TAG/$c
  mkdir -p "$OUT"
  ARGS="bench.py --case $c --envs $n --no-cpu-baseline --no-extras"
  PARGS="$ARGS --lanes 1"  # counter passes: one state lane, launches in order (per-dispatch counters of overlapping launches are not separable)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $ARGS --steps 10 --warmup 2 > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $PARGS --steps 4 --warmup 1 > /dev/null 2> "$OUT/fetch.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $PARGS --steps 4 --warmup 1 > /dev/null 2> "$OUT/write.err"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq1" -- python3 $PARGS --steps 4 --warmup 1 > /dev/null 2> "$OUT/sq1.err"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq2" -- python3 $PARGS --steps 4 --warmup 1 > /dev/null 2> "$OUT/sq2.err"
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq3" -- python3 $PARGS --steps 4 --warmup 1 > /dev/null 2> "$OUT/sq3.err"
  rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 --output-format csv -d "$OUT/sq4" -- python3 $PARGS --steps 4 --warmup 1 > /dev/null 2> "$OUT/sq4.err"
  python3 $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "$c done: $(cut -c1-200 "$OUT/bench.json")"
done
