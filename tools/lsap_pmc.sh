#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over the stand-alone register LSAP (tools/lsap_probe.py, 1024 problems = one wave per SIMD): what a
# scan step costs in instructions and in cycles when a wave runs alone.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03lsap; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/lsap_probe.py 1024 > $OUT/probe1.txt 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -- python3 tools/lsap_probe.py 1024 > /dev/null 2> $OUT/p2.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/r03lsap/{p}/*/*_counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if "k_lsap" in r["Kernel_Name"]:
            key = ("reg" if "Lb1" in r["Kernel_Name"] or "true" in r["Kernel_Name"] else "lds") + " disp " + r["Dispatch_Id"]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc, key=lambda s: int(s.split()[-1])):
    print(k, {c: sum(v) / len(v) for c, v in acc[k].items()})
PY
