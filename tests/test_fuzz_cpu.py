"""A small slice of the wide differential fuzz inside the regular CPU suite (build container only: needs the read-only reference
checkout): the drivers themselves — tests/fuzz_reference.py, fuzz_reference_rl.py, fuzz_facade.py — so that they stay runnable and a
regression of the oracle or the facade against the REFERENCE on a random configuration shows up without anybody starting a fuzz run."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/mUAV_TA"), reason="reference checkout not present (GPU box)")


def _ok(results):
    bad = [r for r in results if r[1] not in ("ok", "skip")]
    assert not bad, bad
    assert sum(r[1] == "ok" for r in results) >= len(results) - 2


def test_reference_to_oracle_on_random_configurations():
    import fuzz_reference as F
    _ok([F.run_one(k) for k in list(range(700, 716)) + [1000003, 1000010]])


def test_reference_to_oracle_with_random_list_valued_actions(monkeypatch):
    import fuzz_reference as F
    monkeypatch.setattr(sys, "argv", ["fuzz_reference.py", "--lists"])
    _ok([F.run_one(k) for k in range(720, 732)])


def test_reference_rl_loop_to_oracle_on_random_configurations():
    import fuzz_reference_rl as F
    _ok([F.run_one(k) for k in range(740, 748)])


def test_reference_to_facade_with_out_of_step_mutators(monkeypatch):
    import fuzz_facade as F
    monkeypatch.setattr(sys, "argv", ["fuzz_facade.py", "--mutators"])
    _ok([F.run_one(k) for k in range(760, 766)])
