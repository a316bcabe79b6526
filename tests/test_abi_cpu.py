"""No-GPU checks of the drop-in boundary: the shared library loads, exports every symbol include/muavta.h
declares, agrees with the ctypes binding on struct layout, and fails LOUDLY without a device (no CPU
fallback).  No compute calls are made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from muavta_amd import native
from muavta_amd.params import MuavtaDims, MuavtaParams, params_for_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    native.build()
    return native.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "muavta.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|const char\*)\s+(muavta_\w+)\s*\(", text, flags=re.M)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/muavta.h but not exported by libmuavta.so"
    assert set(names) == set(native.EXPORTS), set(names) ^ set(native.EXPORTS)


def test_struct_layout_matches_header(lib):
    sizes = (C.c_int32 * 3)()
    assert lib.muavta_abi_sizes(C.byref(sizes)) == 0
    assert sizes[0] == C.sizeof(MuavtaParams) and sizes[1] == C.sizeof(MuavtaDims) and sizes[2] == 1


def test_no_device_is_a_loud_error_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = params_for_case("WPS_hard")
    h = C.c_void_p()
    rc = lib.muavta_create(C.byref(p), 4, 0, C.byref(h))
    assert rc == -2 and not h.value  # MUAVTA_E_NO_DEVICE
    assert b"no CPU fallback" in lib.muavta_last_error(None)
    from muavta_amd.batched import BatchedMultiUAVEnv, lsap
    from muavta_amd.native import MuavtaError
    with pytest.raises(MuavtaError):
        BatchedMultiUAVEnv(p, 4)
    with pytest.raises(MuavtaError):
        lsap(np.zeros((3, 3)))


def test_bad_arguments_are_rejected_before_touching_a_device(lib):
    h = C.c_void_p()
    assert lib.muavta_create(None, 4, 0, C.byref(h)) == -1
    assert lib.muavta_dims(None, None) == -1
    assert lib.muavta_abi_sizes(None) == -1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multi-uav-ta-gym-env_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "import orc" not in text and "oracle_backend" not in text, f
                assert not re.search(r'#include\s+"[^"]*oracle', text), f


def test_binding_structs_match_the_header_field_by_field(tmp_path):
    """MuavtaParams / MuavtaDims / MuavtaRecord / MuavtaScored / MuavtaRlStep / MuavtaRlRun: sizeof and the offset of every field of the ctypes
    binding against a C program that includes include/muavta.h (gcc: the header is plain C)."""
    import subprocess

    from muavta_amd.native import MuavtaRlRun, MuavtaRlStep, MuavtaScored
    from muavta_amd.params import MuavtaRecord

    structs = {"MuavtaParams": MuavtaParams, "MuavtaDims": MuavtaDims, "MuavtaRecord": MuavtaRecord, "MuavtaScored": MuavtaScored, "MuavtaRlStep": MuavtaRlStep, "MuavtaRlRun": MuavtaRlRun}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "muavta.h"', 'int main(void) {']
    for name, st in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field, *_ in st._fields_:
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, st in structs.items():
        assert int(got[name]) == C.sizeof(st), f"sizeof({name}): header {got[name]}, binding {C.sizeof(st)}"
        for field, *_ in st._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(st, field).offset, f"{name}.{field}: header offset {got[f'{name}.{field}']}, binding {getattr(st, field).offset}"
