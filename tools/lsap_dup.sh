#!/bin/bash
# Runs ON THE GPU BOX: kernel time of the stand-alone register LSAP (1024 problems, one wave per SIMD) for the shipped library and for
# builds that execute one piece of the scan step twice (-DMUAVTA_LSAP_DUP=1 wave minimum, 2 cost fetch, 3 row dual broadcast): the
# difference per scan step is what that piece costs on the dependent chain.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03dup; mkdir -p $OUT
for v in default dup1 dup2 dup3; do
  so=tools/_build/libmuavta_$v.so; [ $v = default ] && so=multi-uav-ta-gym-env_amd/libmuavta.so
  MUAVTA_SO=$so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 tools/lsap_probe.py 1024 > $OUT/$v.txt 2> $OUT/$v.err
done
python3 - <<'PY'
import csv, glob
for v in ("default", "dup1", "dup2", "dup3"):
    f = glob.glob(f"gpurun_out/r03dup/{v}/*/*_kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_lsap" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    # 4 launches per shape (1 warm + 3), shapes: 7x35, 21x60, 16x23 registers, 7x35, 21x60 lds
    per = [sum(d[i * 4 + 1:i * 4 + 4]) / 3 for i in range(len(d) // 4)]
    print(v, ["%.1f us" % x for x in per])
PY
