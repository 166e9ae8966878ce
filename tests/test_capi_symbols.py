"""The C-ABI library loads without a GPU and exports exactly what include/vsc_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vsc_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(vsc_[a-z0-9_]+)\s*\(", text))


@pytest.fixture(scope="module")
def lib():
    from vsc_hip import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from vsc_hip import _lib
    declared = _declared()
    assert declared, "no declarations parsed from the header"
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (vsc_[a-z0-9_]+)", nm))
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but undeclared: {sorted(exported - declared)}"
    assert set(_lib.SIGNATURES) == declared, "python binding and header disagree"


def test_library_reports_version_and_no_device_without_gpu(lib):
    import torch
    assert b"gfx950" in lib.vsc_version()
    if not torch.cuda.is_available():
        assert lib.vsc_device_count() <= 0
