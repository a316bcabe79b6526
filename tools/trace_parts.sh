#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of bench.py and of tools/parts_probe.py, to compare the sub-batch launches (queue ids, durations, overlap)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03trace; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/bench -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
GPU_MAX_HW_QUEUES=8 PROBE_TORCH=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/probe -- python3 tools/parts_probe.py > $OUT/probe.txt 2> $OUT/probe.err
python3 - <<'PY'
import csv, glob, collections
for tag in ("bench", "probe"):
    f = glob.glob(f"gpurun_out/r03trace/{tag}/*/*_kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_rollout" in r["Kernel_Name"]]
    # launches of 2048 workgroups (two parts of 4096 envs): Grid_Size = 2048*64
    sel = [r for r in rows if int(r["Grid_Size"]) == 2048 * 64]
    print(tag, "k_rollout launches", len(rows), "of 2048 envs", len(sel))
    if not sel: continue
    sel.sort(key=lambda r: int(r["Start_Timestamp"]))
    sel = sel[len(sel) // 2:]  # second repetition
    q = collections.Counter(r["Queue_Id"] for r in sel)
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    ov = sum(1 for a, b in zip(sel, sel[1:]) if int(b["Start_Timestamp"]) < int(a["End_Timestamp"]))
    print("  queues", dict(q), "mean dur us %.1f" % (sum(dur) / len(dur) / 1e3), "span ms %.2f" % (span / 1e6), "overlapping successors", ov, "of", len(sel) - 1)
    print("  first 8 (start us, dur us, queue):", [(round((int(r["Start_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3, 1), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), r["Queue_Id"]) for r in sel[:8]])
PY
