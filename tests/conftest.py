import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def host_libm_note():
    """None when the host's libm is the one csrc/muavta_math.h / muavta_atan2.h restate (glibc 2.35, the FMA + AVX2 ifunc variants), else why
    not: the bit-for-bit checks of the device's log / atan2 against the HOST's functions only mean something on that libm."""
    import platform

    lib, ver = platform.libc_ver()
    if lib != "glibc" or ver != "2.35":
        return f"host libc is {lib} {ver}, not glibc 2.35"
    try:
        flags = next(line for line in open("/proc/cpuinfo") if line.startswith("flags")).split()
    except (OSError, StopIteration):
        return "cannot read the CPU flags"
    if "fma" not in flags or "avx2" not in flags:
        return "host CPU lacks FMA + AVX2: its libm runs another ifunc variant"
    return None
