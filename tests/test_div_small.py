"""Sim::div_small (csrc/sim/observe.inc): x / c by q = RN(x * inv), r = fma(-q, c, x), q' = fma(r, inv, q) must equal the IEEE
quotient bit for bit for the divisors the observation writer uses (1200, 6, max_time_steps, max_tasks).  The FMA is emulated
exactly with rationals, so this is a check of the algorithm the kernel relies on, run on the CPU."""
import math
import random
from fractions import Fraction

import pytest


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))  # one rounding, round-half-even


def div_small(x, c):
    inv = 1.0 / c
    q = x * inv
    return fma(fma(-q, c, x), inv, q)


DIVISORS = [1200.0, 6.0, 150.0, 200.0, 300.0, 1000.0, 20000.0, 40.0, 48.0, 128.0, 4096.0, 3.0, 7.0, 19999.0, 32767.0]


@pytest.mark.parametrize("c", DIVISORS)
def test_random_doubles_and_integers(c):
    rng = random.Random(int(c))
    xs = [rng.uniform(-1200.0, 1200.0) for _ in range(4000)] + [rng.uniform(0, 1) * 10.0 ** rng.randint(-8, 8) for _ in range(2000)]
    xs += [float(i) for i in range(-2100, 2101)] + [float(rng.randint(-40000, 40000)) for _ in range(2000)]
    for x in xs:
        assert div_small(x, c) == x / c, (x, c)


@pytest.mark.parametrize("c", DIVISORS)
def test_quotients_next_to_rounding_boundaries(c):
    """x chosen so that x / c lands as close as a double x allows to the midpoint between two neighbouring doubles."""
    rng = random.Random(1000 + int(c))
    for _ in range(3000):
        q = rng.uniform(0.001, 2.0) * 2.0 ** rng.randint(-10, 10)
        mid = Fraction(q) + Fraction(math.ulp(q)) / 2
        x0 = float(mid * Fraction(c))
        for x in (x0, math.nextafter(x0, math.inf), math.nextafter(x0, -math.inf), -x0):
            assert div_small(x, c) == x / c, (x, c)


def test_zero_and_signs():
    for c in DIVISORS:
        assert div_small(0.0, c) == 0.0
        assert div_small(-3.0, c) == -3.0 / c
