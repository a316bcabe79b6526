#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in a built libmuavta.so, read from the code object's metadata
(builder's tool; no recompile).  usage: kernel_meta.py [path/to/libmuavta.so] [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def short(name):
    m = re.search(r"(k_\w+?)I4TileILi(\d+)ELi(\d+).*?EE(Lb(\d))?", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}x{m.group(3)}{',REC' if m.group(5) == '1' else ''}>"
    m = re.search(r"N_1\d+(k_\w+?)E", name)
    return m.group(1) if m else name[:60]


def main():
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-uav-ta-gym-env_amd", "libmuavta.so")
    subs = [a for a in sys.argv[1:] if not a.endswith(".so")]
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}", "--unbundle"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    print(f"{'kernel':34s} {'VGPR':>5s} {'vspill':>6s} {'SGPR':>5s} {'sspill':>6s} {'scratch B':>9s} {'LDS B':>7s}")
    for blk in notes.split("  - .agpr_count:")[1:]:
        def f(key):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = short(f("name"))
        if subs and not any(s in name for s in subs):
            continue
        print(f"{name:34s} {f('vgpr_count'):>5s} {f('vgpr_spill_count'):>6s} {f('sgpr_count'):>5s} {f('sgpr_spill_count'):>6s} {f('private_segment_fixed_size'):>9s} {f('group_segment_fixed_size'):>7s}")


if __name__ == "__main__":
    main()
