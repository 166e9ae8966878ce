"""The committed fixtures are the REFERENCE's outputs: its own SwinTransformerV2 / CLIPModel / MS / SSCD-head classes,
instantiated from /root/reference (tests/golden/_reference_classes.py), reproduce tests/golden/*.npz.

These tests EXECUTE class definitions read from the untrusted reference tree, so they are opt-in and isolated:
    VSC_RUN_REFERENCE_CODE=1 python -m pytest tests/test_golden_vs_reference.py
runs every check in a child process with a scrubbed environment (no tokens, no proxy settings, HOME in a scratch directory) whose
working directory is that scratch directory; the loader itself refuses files whose SHA-256 is not the pinned one and executes no
`os` / `sys` / `subprocess` import.  Without the variable (the driver's CPU run, the GPU box) they are skipped."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
OPT_IN = "VSC_RUN_REFERENCE_CODE"

runs_reference_code = pytest.mark.skipif(not (os.path.isdir("/root/reference") and os.environ.get(OPT_IN) == "1"),
                                         reason=f"executes reference code: needs /root/reference and {OPT_IN}=1")


def _child(code, tmp_path):
    env = {"PATH": "/usr/bin:/bin", "HOME": str(tmp_path), "TMPDIR": str(tmp_path), OPT_IN: "1",
           "PYTHONPATH": os.pathsep.join([GOLD, ROOT, os.path.join(ROOT, "vsc22-submission_amd")]),
           "HF_HUB_OFFLINE": "1", "TRANSFORMERS_OFFLINE": "1", "PYTHONDONTWRITEBYTECODE": "1"}
    return subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)


@runs_reference_code
@pytest.mark.parametrize("kind,preset", [("swin", "tiny_swin"), ("swin", "tiny_swin_w8"), ("swin", "swinv2_base_256"), ("swin", "tiny_swin_w24"), ("swinoutlier", "swinv2_base_256"),
                                         ("clip", "tiny_clip"), ("vit", "tiny"), ("vit", "vit_b16_224"), ("vitoutlier", "vit_b16_224"), ("sscd", "vit_v68"), ("vsm", "tiny_vsm"), ("uape2e", "chain"), ("uape2e", "large")])
def test_fixture_equals_reference_class_output(kind, preset, tmp_path):
    r = _child(f"import check_golden_against_reference as chk; err = chk.check_{kind}({preset!r}); print('ERR', err); "
               f"raise SystemExit(0 if err <= chk.ATOL else 1)", tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@runs_reference_code
def test_reference_loader_executes_definitions_only(tmp_path):
    """The loader must not run the reference scripts' module-level statements (checkpoint paths, __main__ blocks), must not import
    `os` on their behalf, and must refuse a file whose hash is not the pinned one."""
    r = _child("import _reference_classes as refc\n"
               "ns = refc.load_definitions(refc.SWIN_SRC)\n"
               "assert 'SwinTransformerV2' in ns and 'CHECKPOINT_PATH' not in ns and 'build_model' not in ns\n"
               "assert 'os' not in refc.load_definitions(refc.CLIP_SRC)\n"
               "refc._PINNED[refc.SWIN_SRC] = '0' * 64\n"
               "try:\n"
               "    refc.load_definitions(refc.SWIN_SRC)\n"
               "    raise SystemExit(1)\n"
               "except RuntimeError as e:\n"
               "    assert 'pinned' in str(e)\n", tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_loader_is_opt_in():
    """Without the variable nothing is executed, whatever exists on disk."""
    sys.path.insert(0, GOLD)
    import _reference_classes as refc
    old = os.environ.pop(OPT_IN, None)
    try:
        assert not refc.available()
        with pytest.raises(RuntimeError, match="opt-in"):
            refc.load_definitions(refc.SWIN_SRC)
    finally:
        if old is not None:
            os.environ[OPT_IN] = old
