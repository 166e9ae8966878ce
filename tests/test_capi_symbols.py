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


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def lib(request):
    """both builds of the library: bf16 operands (libvsc_hip.so) and fp16 operands (libvsc_hip_f16.so)"""
    from vsc_hip import _lib
    if not all(os.path.exists(p) for p in _lib.LIB_PATHS.values()):
        import __graft_entry__
        __graft_entry__.build()
    out = _lib.load(request.param)
    out.precision = request.param
    return out


def test_every_declared_symbol_is_exported(lib):
    from vsc_hip import _lib
    declared = _declared()
    assert declared, "no declarations parsed from the header"
    assert lib.vsc_operand_dtype().decode() == lib.precision
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATHS[lib.precision]], text=True)
    exported = set(re.findall(r" T (vsc_[a-z0-9_]+)", nm))
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but undeclared: {sorted(exported - declared)}"
    assert set(_lib.SIGNATURES) == declared, "python binding and header disagree"


def test_library_reports_version_and_no_device_without_gpu(lib):
    import torch
    assert b"gfx950" in lib.vsc_version()
    # the loaded library was built from THIS tree: vsc_version() carries the hash of csrc/* + include/vsc_hip.h (csrc/Makefile: HASHED).
    # .so files are git-ignored and travel to the GPU box as untracked artefacts -- a stale one must fail here, not be tested silently.
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "vsc22-submission_amd", "csrc")
    names = sorted([os.path.basename(p) for ext in ("*.hip", "*.h", "*.inc") for p in glob.glob(os.path.join(csrc, ext))] + ["Makefile"])
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(csrc, n), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "vsc_hip.h"), "rb").read())
    assert ("src " + h.hexdigest()[:16]).encode() in lib.vsc_version(), (lib.vsc_version(), h.hexdigest()[:16], "rebuild: make -C vsc22-submission_amd/csrc")
    if not torch.cuda.is_available():
        assert lib.vsc_device_count() <= 0


def test_option_table_is_read_once_and_set_through_the_api(lib, monkeypatch):
    """vsc_set_option: known switches with or without the VSC_ prefix, NULL clears, unknown names are refused -- and a change
    of the environment after the first use is not seen (the library reads it once per process)."""
    from vsc_hip import _lib
    _lib.set_option("VSC_KNN_PATH", "bf16")
    _lib.set_option("KNN_PATH", None)
    with _lib.option("VSC_GEMM_V4", "0"):
        assert _lib.get_option("GEMM_V4") == "0"
        with _lib.option("VSC_GEMM_V4", "1"):          # nested: the inner exit restores the OUTER value, not "unset"
            assert _lib.get_option("VSC_GEMM_V4") == "1"
        assert _lib.get_option("VSC_GEMM_V4") == "0"
    assert _lib.get_option("VSC_GEMM_V4") is None and _lib.get_option("VSC_NO_SUCH_SWITCH") is None
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


def test_swin_mlp_hidden_permutation_is_the_documented_bijection(lib):
    """vsc_swin_mlp_permute_hidden_f32 is host code (no GPU): dst[n][32 S + 8 g + 4 t + i] = src[n][32 S + 16 t + 4 g + i], a
    bijection of every row's hidden axis; other widths are refused."""
    import numpy as np
    from vsc_hip._lib import VscHipError, check
    for c in (128, 256):
        src = np.arange(c * 4 * c, dtype=np.float32).reshape(c, 4 * c)
        dst = np.full_like(src, -1.0)
        check(lib.vsc_swin_mlp_permute_hidden_f32(src.ctypes.data, dst.ctypes.data, c))
        k = np.arange(4 * c)
        S, t, g, i = k >> 5, (k >> 4) & 1, (k >> 2) & 3, k & 3
        assert np.array_equal(dst[:, 32 * S + 8 * g + 4 * t + i], src[:, k])
        assert np.array_equal(np.sort(dst, axis=1), src)
    # c = 512: chunk-major [64][512][32], the same order inside a 32-block (csrc/swin_mlp512.hip)
    c = 512
    src = np.arange(c * 4 * c, dtype=np.float32).reshape(c, 4 * c)
    dst = np.full(c * 4 * c, -1.0, dtype=np.float32)
    check(lib.vsc_swin_mlp_permute_hidden_f32(src.ctypes.data, dst.ctypes.data, c))
    k = np.arange(4 * c)
    ch, t, g, i = k >> 5, (k >> 4) & 1, (k >> 2) & 3, k & 3
    assert np.array_equal(dst.reshape(64, c, 32)[ch, :, 8 * g + 4 * t + i].T, src[:, k])
    assert np.array_equal(np.sort(dst), src.reshape(-1))
    with pytest.raises(VscHipError, match="unsupported"):
        buf = np.zeros((64, 256), dtype=np.float32)
        check(lib.vsc_swin_mlp_permute_hidden_f32(buf.ctypes.data, buf.copy().ctypes.data, 64))


def test_lds_layouts_are_conflict_free():
    """The LDS layouts the attention kernels and the fused Swin MLP read with ds_read_b128, checked exhaustively against the
    bank rule (tools/micro/lds_swizzle_check.py): one LDS cycle per 16-lane group."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools", "micro"))
    import lds_swizzle_check as L
    for C in (128, 256, 512):
        assert max(L.cycles(lambda l, j=j, ks=ks: (16 * j + (l & 15)) * 2 * C + (((4 * ks + (l >> 4)) ^ (l & 15)) << 4))
                   for j in range(4 if C < 512 else 2) for ks in range(C // 32)) == 4
    # swin_mlp512.hip, W2 chunk: 64-byte rows, fragment jo: row 32 (jo >> 1) + 8 (fr >> 2) + 4 (jo & 1) + (fr & 3), piece quad ^ ((row & 1) | ((row >> 3) & 1) << 1)
    def w2_512(l, jo):
        fr, quad = l & 15, l >> 4
        n = 32 * (jo >> 1) + 8 * (fr >> 2) + 4 * (jo & 1) + (fr & 3)
        return n * 64 + ((quad ^ ((n & 1) | (((n >> 3) & 1) << 1))) << 4)
    assert max(L.cycles(lambda l, jo=jo: w2_512(l, jo)) for jo in range(32)) == 4
    for tp in range(32, 321, 32):
        s = tp * 2 + 32
        assert max(L.cycles(lambda l, ct=ct, u=u: (ct * 16 + (l & 15)) * s + (32 * u + 8 * (l >> 4)) * 2)
                   for ct in range(2) for u in range(tp // 32)) == 4
    # and the old V^T row stride (2 TP + 8 bytes) is what a b128 read could NOT have used: misaligned rows
    assert (224 * 2 + 8) % 16 != 0


def test_operand_type_plumbing_refuses_what_does_not_exist():
    """`precision` names a build of the library: anything but "bf16" / "fp16" is refused before any device work, and a library whose
    vsc_operand_dtype() is not the requested one (a misplaced or stale build) is refused at load."""
    from vsc_hip import _lib, ops
    with pytest.raises(ValueError, match="precision must be one of"):
        _lib.load("fp8")
    with pytest.raises(AssertionError):
        ops.operands("fp8")
    assert set(_lib.LIB_PATHS) == {"bf16", "fp16"}
    saved = dict(_lib.LIB_PATHS)
    try:
        _lib._libs.pop("fp16", None)
        _lib.LIB_PATHS["fp16"] = saved["bf16"]          # the bf16 library under the fp16 name
        with pytest.raises(_lib.HipPathUnavailable, match="reports operand type bf16"):
            _lib.load("fp16")
    finally:
        _lib.LIB_PATHS.update(saved)
        _lib._libs.pop("fp16", None)
    assert _lib.load("fp16").vsc_operand_dtype() == b"fp16" and _lib.load("bf16").vsc_operand_dtype() == b"bf16"
    with ops.operands("fp16"):
        import torch
        assert ops.lp_dtype() == torch.float16
    assert ops.lp_dtype() == torch.bfloat16
    from src.model_zoo import DEFAULT_PRECISION
    assert DEFAULT_PRECISION == "fp16"
