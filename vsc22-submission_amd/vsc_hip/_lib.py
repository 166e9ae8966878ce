"""ctypes binding of libvsc_hip.so (include/vsc_hip.h).

There is no CPU implementation behind these calls: if the shared library is
missing, or no gfx950 device is visible, they raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSC_HIP_LIB: another build of the library (same-box A/B runs of an experimental kernel build, tools/micro/*); default: the in-tree one
LIB_PATH = os.environ.get("VSC_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libvsc_hip.so")
# One library per 16-bit operand type of the encoders (vsc_operand_dtype): "bf16" is the configuration BASELINE.json names and the
# one the search / CNN / kernel-level wrappers use; "fp16" is the same kernels with 11-bit significands (csrc/common.h).
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.environ.get("VSC_HIP_LIB_F16") or os.path.join(os.path.dirname(_HERE), "lib", "libvsc_hip_f16.so")}

EPI_BF16, EPI_GELU_BF16, EPI_QGELU_BF16, EPI_RESADD_F32, EPI_PATCH_F32, EPI_F32 = range(6)
PROF_CLASSES = ("patchify", "gemm_patch", "layernorm", "gemm_qkv", "attention", "gemm_proj",
                "gemm_fc1", "gemm_fc2", "pool_head", "misc")


SWIN_PROF_CLASSES = 27
SWIN_PROF_STAGE0, SWIN_PROF_PER_STAGE = 3, 6
SWIN_PROF_KINDS = ("qkv", "attention", "proj_ln", "fc1", "fc2_ln", "merge")


class HipPathUnavailable(RuntimeError):
    """The HIP hot path cannot run here (library not built or no MI355X)."""


class VscHipError(RuntimeError):
    pass


class EncoderConfigC(ctypes.Structure):
    _fields_ = [
        ("image_size", c_int32), ("patch_size", c_int32), ("channels", c_int32),
        ("width", c_int32), ("layers", c_int32), ("heads", c_int32), ("mlp_dim", c_int32),
        ("out_dim", c_int32), ("ln_eps", c_float), ("act", c_int32), ("pre_ln", c_int32),
        ("patch_bias", c_int32), ("pool", c_int32), ("gem_p", c_float),
        ("max_batch", c_int32), ("l2_normalize", c_int32), ("head_conv_dim", c_int32), ("lanes", c_int32), ("fuse_ln", c_int32),
    ]


class SwinConfigC(ctypes.Structure):
    _fields_ = [
        ("image_size", c_int32), ("patch_size", c_int32), ("channels", c_int32), ("embed_dim", c_int32),
        ("stages", c_int32), ("depths", c_int32 * 4), ("heads", c_int32 * 4), ("window_size", c_int32),
        ("pretrained_window_sizes", c_int32 * 4), ("mlp_ratio", c_int32), ("out_dim", c_int32),
        ("ln_eps", c_float), ("gem_p", c_float), ("max_batch", c_int32), ("l2_normalize", c_int32),
    ]


# name -> (restype, argtypes); also the list the symbol test checks against the header
SIGNATURES = {
    "vsc_last_error": (c_char_p, []),
    "vsc_device_count": (c_int32, []),
    "vsc_version": (c_char_p, []),
    "vsc_operand_dtype": (c_char_p, []),
    "vsc_set_option": (c_int32, [c_char_p, c_char_p]),
    "vsc_get_option": (c_char_p, [c_char_p]),
    "vsc_encoder_create": (c_int32, [POINTER(EncoderConfigC), POINTER(c_void_p)]),
    "vsc_encoder_destroy": (None, [c_void_p]),
    "vsc_encoder_set_weight": (c_int32, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "vsc_encoder_finalize": (c_int32, [c_void_p]),
    "vsc_encoder_forward": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "vsc_encoder_forward_debug": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vsc_encoder_workspace_bytes": (c_int64, [c_void_p]),
    "vsc_encoder_set_profiling": (c_int32, [c_void_p, c_int32]),
    "vsc_encoder_get_profile": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "vsc_swin_create": (c_int32, [POINTER(SwinConfigC), POINTER(c_void_p)]),
    "vsc_swin_destroy": (None, [c_void_p]),
    "vsc_swin_set_weight": (c_int32, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "vsc_swin_finalize": (c_int32, [c_void_p]),
    "vsc_swin_forward": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "vsc_swin_forward_debug": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "vsc_swin_workspace_bytes": (c_int64, [c_void_p]),
    "vsc_swin_set_profiling": (c_int32, [c_void_p, c_int32]),
    "vsc_swin_get_profile": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "vsc_window_attention_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                            c_int32, c_int32, c_void_p]),
    "vsc_ln_residual_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                      c_float, c_void_p]),
    "vsc_gemm_ln_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int32, c_int32, c_float, c_void_p]),
    "vsc_swin_mlp_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_int32, c_float, c_void_p]),
    "vsc_swin_mlp_permute_hidden_f32": (c_int32, [c_void_p, c_void_p, c_int32]),
    "vsc_debug_mlp512_timing": (c_int32, [c_void_p]),
    "vsc_swin_proj_mlp_bf16": (c_int32, [c_void_p] * 13 + [c_int64, c_int32, ctypes.c_float, c_void_p]),
    "vsc_swin_proj_mlp_qkv_bf16": (c_int32, [c_void_p] * 15 + [c_int64, c_int32, ctypes.c_float, c_void_p]),
    "vsc_merge_gather_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "vsc_pair_similarity_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p,
                                          c_void_p, c_int64, c_void_p]),
    "vsc_video_pair_max_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int32,
                                         c_int32, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                         c_void_p]),
    "vsc_encoder_forward_u8": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vsc_swin_forward_u8": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vsc_debug_spin_ticks": (c_int32, [ctypes.c_uint64, c_void_p, c_void_p]),
    "vsc_knn_ip_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int64,
                                 c_void_p, c_void_p, c_void_p]),
    "vsc_knn_ip_floor_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vsc_knn_merge_parts_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "vsc_conv_packed_k": (c_int32, [c_int32, c_int32, c_int32]),
    "vsc_conv_pack_weight_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "vsc_conv2d_f32": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                 c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "vsc_dwconv2d_f32": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                   c_int32, c_int32, c_void_p, c_void_p]),
    "vsc_global_avgpool_f32": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "vsc_channel_scale_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "vsc_upsample_add_f32": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32,
                                       c_int32, c_int32, c_void_p]),
    "vsc_se_block_f32": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "vsc_upsample_sum_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int32,
                                       c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "vsc_conv_last_pipe": (c_int32, []),
    "vsc_attention_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "vsc_attention_f32_batch": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vsc_attention_f32_varlen": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vsc_knn_last_path": (c_int32, []),
    "vsc_search_release_scratch": (c_int64, []),
    "vsc_video_pair_max_last_path": (c_int32, []),
    "vsc_range_search_last_path": (c_int32, []),
    "vsc_knn_set_profiling": (None, [c_int32]),
    "vsc_knn_last_profile": (c_int32, [c_void_p]),
    "vsc_range_search_ip_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_float, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_int64, POINTER(c_int64), c_void_p]),
    "vsc_l2_normalize_f32": (c_int32, [c_void_p, c_int64, c_int32, c_void_p]),
    "vsc_gemm_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                c_int32, c_int32, c_int32, c_void_p]),
    "vsc_attention_bf16": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "vsc_layernorm_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                                    c_int32, c_void_p]),
    "vsc_patchify_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                    c_void_p]),
}

_libs = {}
_options = {}     # switches set through set_option so far: replayed into a library loaded later


def load(precision: str = "bf16") -> ctypes.CDLL:
    """Load the shared library of that operand type (no GPU needed for this step)."""
    lib = _libs.get(precision)
    if lib is None:
        if precision not in LIB_PATHS:
            raise ValueError(f"precision must be one of {sorted(LIB_PATHS)}, not {precision!r}")
        path = LIB_PATHS[precision]
        if not os.path.exists(path):
            raise HipPathUnavailable(
                f"{path} is not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vsc22-submission_amd/csrc`; there is no CPU fallback")
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.vsc_operand_dtype().decode()
        if got != precision:
            raise HipPathUnavailable(f"{path} reports operand type {got}, expected {precision}: stale or misplaced build")
        for name, value in _options.items():
            lib.vsc_set_option(name.encode(), None if value is None else str(value).encode())
        _libs[precision] = lib
    return lib


def require_device(precision: str = "bf16") -> ctypes.CDLL:
    lib = load(precision)
    n = lib.vsc_device_count()
    if n < 1:
        raise HipPathUnavailable(
            "no gfx950 (MI355X) device visible to HIP "
            f"(vsc_device_count() = {n}: {lib.vsc_last_error().decode()}); there is no CPU fallback")
    return lib


def set_option(name: str, value=None) -> None:
    """Set (or, with None, clear) a diagnostic switch of the library -- vsc_set_option in include/vsc_hip.h.  The library
    reads its VSC_* environment variables once per process; changing os.environ afterwards has no effect."""
    load()
    _options[name] = value
    for lib in list(_libs.values()):
        check(lib.vsc_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str):
    """Current value of a switch (str) or None -- vsc_get_option."""
    v = load().vsc_get_option(name.encode())
    return None if v is None else v.decode()


class option:
    """with option("VSC_KNN_PATH", "bf16"): ...  -- on exit the switch returns to the value it had on entry (from the
    environment at load, or from an enclosing `with option(...)`), not to "unset"."""

    def __init__(self, name: str, value):
        self.name, self.value, self._saved = name, value, None

    def __enter__(self):
        self._saved = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self._saved)
        return False


def check(rc: int) -> None:
    if rc != 0:
        msgs = [m for m in (lib.vsc_last_error().decode() for lib in _libs.values()) if m]
        raise VscHipError(f"libvsc_hip status {rc}: {' | '.join(msgs)}")


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libvsc_hip takes contiguous device tensors"
    return c_void_p(t.data_ptr())
