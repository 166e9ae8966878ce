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


def test_option_table_is_read_once_and_set_through_the_api(lib, monkeypatch):
    """vsc_set_option: known switches with or without the VSC_ prefix, NULL clears, unknown names are refused -- and a change
    of the environment after the first use is not seen (the library reads it once per process)."""
    from vsc_hip import _lib
    _lib.set_option("VSC_KNN_PATH", "bf16")
    _lib.set_option("KNN_PATH", None)
    with _lib.option("VSC_GEMM_V4", "0"):
        pass
    assert lib.vsc_set_option(b"VSC_NO_SUCH_SWITCH", b"1") != 0
    assert b"unknown switch" in lib.vsc_last_error()
    assert lib.vsc_set_option(None, b"1") != 0
    monkeypatch.setenv("VSC_NO_SUCH_SWITCH", "1")     # harmless: never consulted
    # every switch the sources use is in the table (a getenv() left on a launch path would not be)
    import glob
    import re
    src = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "vsc22-submission_amd", "csrc", "*.hip")) +
                  glob.glob(os.path.join(ROOT, "vsc22-submission_amd", "csrc", "*.h")))
    assert len(re.findall(r"\bgetenv\(", src)) == 1, "getenv outside the once-per-process loader in capi.hip"
    for name in set(re.findall(r"vsc_opt\(OPT_([A-Z0-9_]+)\)", src)):
        assert lib.vsc_set_option(("VSC_" + name).encode(), None) == 0, name
