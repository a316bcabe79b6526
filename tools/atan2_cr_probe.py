#!/usr/bin/env python3
"""How often the host libm's atan2 is NOT the correctly rounded value (mpmath, 200 bits): glibc >= 2.34 keeps only the first stage of
the IBM Accurate Mathematical Library's atan2, so about one result in a thousand is the other neighbour.  That is why the device
restates glibc's algorithm (csrc/muavta_atan2.h) instead of rounding correctly: core_sim's avoid_obstacles decides on the last bit
when an agent sits on an obstacle's axis.  Output kept as profiles/r05_atan2_glibc_vs_cr.txt."""
import math, random, mpmath
mpmath.mp.prec = 200
random.seed(5)
bad = 0; n = 60000
for i in range(n):
    if i % 3 == 0:
        y = random.uniform(-1, 1); x = random.uniform(-1, 1)
    elif i % 3 == 1:
        y = random.uniform(-800, 800); x = random.uniform(-800, 800)
    else:
        y = random.uniform(-1, 1) * 10 ** random.uniform(-3, 3); x = random.uniform(-1, 1) * 10 ** random.uniform(-3, 3)
    g = math.atan2(y, x)
    w = float(mpmath.atan2(mpmath.mpf(y), mpmath.mpf(x)))  # mpf -> float rounds to nearest
    if g != w:
        bad += 1
        print("diff", y.hex(), x.hex(), g.hex(), w.hex())
print("glibc atan2 vs correctly rounded:", bad, "of", n)
