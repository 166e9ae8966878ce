"""Build-container helper: makes the REFERENCE's own model classes importable so that the committed fixtures can be
shown to be the reference's outputs (used by check_golden_against_reference.py and gen_sscd_golden.py; nothing on the
GPU box imports this — /root/reference does not exist there).

The reference files do not import here as modules: their headers pull packages this image lacks (timm, mmcv,
classy_vision) and the training-tree package `vsc.baseline.model_factory`.  None of those is on the numeric path of the
classes we need, so the file is parsed and only its class / function definitions and its torch / numpy / typing imports
are executed, with three helper names supplied:

* `DropPath(p)`       — stochastic depth; identity in eval mode (timm's implementation returns x when not training);
* `to_2tuple(x)`      — (x, x) for a scalar, tuple(x) otherwise;
* `trunc_normal_`     — parameter INIT only (every parameter is overwritten by the loaded state dict);
* `load_checkpoint`   — mmcv's loader, referenced by `init_weights` only, never by `forward`;
* `BACKBONES`         — the registry decorator of the training tree: registration only.
No reference source text is copied: it is read from /root/reference at run time.

The reference tree is UNTRUSTED public content and this loader executes definitions taken from it.  Hence:
* opt-in only: nothing here runs unless VSC_RUN_REFERENCE_CODE=1 is set (a present /root/reference is not consent);
* every file is pinned by SHA-256 (_PINNED): a tree that differs from the one these helpers were reviewed against is refused;
* the import whitelist holds no `os` / `sys` / `subprocess`: the one reference file that imports `os` (video/clip.py, for
  os.path.join in a checkpoint loader nobody calls) loses that import and the loader function raises NameError if ever called;
* tests/test_golden_vs_reference.py runs the checks in a child process with a scrubbed environment and a scratch working directory.
"""
import ast
import hashlib
import os

import torch
import torch.nn as nn

REFERENCE = "/root/reference"
SWIN_SRC = "VSC22-Descriptor-Track-1st/train/train_v115/torch2scripts.py"
CLIP_SRC = "VSC22-Descriptor-Track-1st/train/train_vid_score/video/clip.py"
SSCD_SRC = "VSC22-Descriptor-Track-1st/train/train_v68/vsc/baseline/model_factory/backbones/sscd.py"
VSM_SRC = "VSC22-Descriptor-Track-1st/train/train_vid_score/video/model.py"
VIT_SRC = "VSC22-Descriptor-Track-1st/train/train_v115/vsc/baseline/model_factory/backbones/vit.py"

_ALLOWED_IMPORT_ROOTS = {"torch", "numpy", "typing", "collections", "math", "transformers"}
OPT_IN = "VSC_RUN_REFERENCE_CODE"
# SHA-256 of the reference files whose definitions are executed (the tree this loader was reviewed against)
_PINNED = {
    SWIN_SRC: "e66fd2eafa44edc0b3e6aef8f7c397e4de034bc52e9b3d67c13d039afdd97605",
    CLIP_SRC: "0ed16e443efd547fba54959779a98a6cc0070865d072d0a6aaecd40781a95d03",
    SSCD_SRC: "daf573dca38b65a3c925b4f6cd85d482312270967c0f758e8a179323a81a9654",
    VSM_SRC: "c4e627e3df8564372f571683b602247e40c20682fef6da18cce4c3ab2dc7700d",
    VIT_SRC: "464fd0c3a398af1794673d2c5e05004845f4474db6afe61b7d46fd6b8e957163",
}


def available() -> bool:
    """The reference tree exists AND the caller opted in to executing definitions from it."""
    return os.path.isdir(REFERENCE) and os.environ.get(OPT_IN) == "1"


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training, "stub: eval only"
        return x


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _unused(*a, **k):
    raise RuntimeError("stubbed helper called on the numeric path")


def load_definitions(rel_path: str) -> dict:
    """Namespace with the classes / functions the reference file defines."""
    if os.environ.get(OPT_IN) != "1":
        raise RuntimeError(f"executing reference definitions is opt-in: set {OPT_IN}=1")
    path = os.path.join(REFERENCE, rel_path)
    text = open(path, "rb").read()
    digest = hashlib.sha256(text).hexdigest()
    if _PINNED.get(rel_path) != digest:
        raise RuntimeError(f"{path}: sha256 {digest} is not the pinned one -- the reference tree changed; review before running it")
    tree = ast.parse(text.decode(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)):
            keep.append(node)
        elif isinstance(node, ast.Import) and all(a.name.split(".")[0] in _ALLOWED_IMPORT_ROOTS for a in node.names):
            keep.append(node)
        elif isinstance(node, ast.ImportFrom) and node.level == 0 and (node.module or "").split(".")[0] in _ALLOWED_IMPORT_ROOTS:
            keep.append(node)
    ns = {"__name__": "reference_defs", "DropPath": DropPath, "to_2tuple": to_2tuple,
          "trunc_normal_": torch.nn.init.trunc_normal_, "load_checkpoint": _unused, "BACKBONES": _Registry(),
          "_load_weights": _unused}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns
