#!/usr/bin/env python3
"""Where the SGPR spills of a kernel sit (builder's tool): position of every spill store (v_writelane_b32 v, s, imm) and
reload (v_readlane_b32 s, v, imm) along the kernel's instruction stream, in bins of 1000 instructions, with the loop
structure read off the labels.  usage: spill_map.py file.s "<kernel short name, as tools/isa_stats.py prints it>" """
import collections
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import isa_stats  # noqa: E402


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else "k_rollout<16x40>"
    for name, body in isa_stats.functions(txt):
        if isa_stats.short(name) != want:
            continue
        n = 0
        hist = collections.Counter()
        for l in body:
            s = l.strip()
            if not s or s[0] in ";." or s.endswith(":"):
                continue
            n += 1
            if re.match(r"v_writelane_b32 v\d+, s\d+, \d+$", s):
                hist[("store", n // 1000)] += 1
            elif re.match(r"v_readlane_b32 s\d+, v\d+, \d+$", s):
                hist[("reload", n // 1000)] += 1
        print(f"{want}: {n} instructions; spill stores {sum(v for (k, _), v in hist.items() if k == 'store')}, "
              f"reloads {sum(v for (k, _), v in hist.items() if k == 'reload')} (static)")
        print("instr range   stores reloads")
        for b in range(n // 1000 + 1):
            if hist[("store", b)] or hist[("reload", b)]:
                print(f"{b * 1000:6d}-{b * 1000 + 999:<6d} {hist[('store', b)]:6d} {hist[('reload', b)]:7d}")


if __name__ == "__main__":
    main()
