#!/usr/bin/env python3
"""Static instruction statistics of the kernels in a hipcc -S dump (builder's tool: which kernel carries how many VALU /
SALU / LDS instructions, SGPR-spill lane moves, scratch accesses).  usage: isa_stats.py file.s [substring [out.s]]"""
import collections
import re
import sys


def functions(txt):
    lines = txt.split("\n")
    start = None
    name = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and name:
            yield name, lines[start + 1:i]
            name = None


def short(name):
    m = re.search(r"(k_\w+?)I4TileILi(\d+)ELi(\d+).*?EE(Lb(\d))?", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}x{m.group(3)}{',REC' if m.group(5) == '1' else ''}>"
    m = re.search(r"N_1\d+(k_\w+?)E", name)
    return m.group(1) if m else name[:60]


def stats(body):
    ops = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        ops[l.split()[0]] += 1
    return ops


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else None
    for name, body in functions(txt):
        s = short(name)
        if want and want not in s:
            continue
        ops = stats(body)
        tot = sum(ops.values())
        g = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
        print(f"{s:28s} total {tot:6d} valu {g('v_'):6d} salu {g('s_'):6d} lds {g('ds_'):5d} vmem {g('global_') + g('buffer_') + g('flat_'):5d} "
              f"readlane {ops['v_readlane_b32']:5d} writelane {ops['v_writelane_b32']:5d} scratch {g('scratch_'):4d} waitcnt {ops['s_waitcnt']:5d} "
              f"branch {g('s_cbranch') + ops['s_branch']:5d} sqrt {ops['v_sqrt_f64_e32'] + ops['v_rsq_f64_e32']:4d} div_fmas {ops['v_div_fmas_f64']:4d}")
        if want and len(sys.argv) > 3:
            open(sys.argv[3], "w").write("\n".join(body))


if __name__ == "__main__":
    main()
