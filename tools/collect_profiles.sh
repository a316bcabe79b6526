#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + the two PMC passes the HBM guide prescribes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass), each with --kernel-trace only.  Output under gpurun_out/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq1" -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/sq1.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq2" -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/sq2.err"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
