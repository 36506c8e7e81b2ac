"""ctypes binding of libvisrep_hip.so (the C ABI in include/visrep.h).

There is NO CPU fallback: if the library cannot be loaded (or built with hipcc), every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VISREP_LIB") or os.path.join(_PKG, "libvisrep_hip.so")   # VISREP_LIB: diagnostic builds only

ABI_VERSION = 600                 # include/visrep.h VISREP_VERSION this binding was written against (checked at load)
BF16, F32 = 0, 1
EPI_BIAS, EPI_ACT, EPI_RESID, EPI_VT, EPI_PATCH, EPI_F32 = range(6)
ACT = {"none": 0, "quick_gelu": 1, "gelu": 2, "gelu_erf": 2, "gelu_tanh": 3, "gelu_pytorch_tanh": 3}


class VitConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("image_size", "patch", "d", "heads", "mlp", "layers", "tokens", "has_cls", "pre_ln",
                                       "act", "kpad")] + [("eps", C.c_float), ("q_prescaled", C.c_int)]


class VitLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ls1", "ln2_g", "ln2_b", "w1", "b1",
                                          "w2", "b2", "ls2", "sqkv", "s1")]


class VitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("patch_w", "patch_b", "cls", "pos", "pre_ln_g", "pre_ln_b")] + [
        ("layers", C.POINTER(VitLayer))]


class JpegInfo(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int), ("hs", C.c_int * 3), ("vs", C.c_int * 3), ("hmax", C.c_int),
                ("vmax", C.c_int), ("mcus_w", C.c_int), ("mcus_h", C.c_int), ("blocks_w", C.c_int * 3), ("blocks_h", C.c_int * 3),
                ("comp_w", C.c_int * 3), ("comp_h", C.c_int * 3), ("restart_interval", C.c_int), ("progressive", C.c_int),
                ("unsupported", C.c_int), ("coef_count", C.c_long)]


# name -> (restype, argtypes); every symbol include/visrep.h declares
_vp, _i, _f, _sz, _l = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long
SIGNATURES = {
    "visrep_version": (_i, []),
    "visrep_last_error": (_sz, [C.c_char_p, _sz]),
    "visrep_set_gemm_variant": (_i, [_i]),
    "visrep_set_attn_variant": (_i, [_i]),
    "visrep_set_ascore_variant": (_i, [_i]),
    "visrep_debug_gemm_ablation": (_i, [_i]),
    "visrep_debug_gemm_timing_buffer": (_i, [_vp]),
    "visrep_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "visrep_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "visrep_mhsa_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "visrep_mhsa_cls_supported": (_i, [_i]),
    "visrep_mhsa_cls_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "visrep_set_scratch": (_i, [_vp, _sz]),
    "visrep_set_stream_scratch": (_i, [_vp, _vp, _sz]),
    "visrep_layernorm_stats": (_i, [_vp, _i, _vp, _i, _i, _f, _vp]),
    "visrep_gemm_bf16_ln": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "visrep_gemm_bf16_rows": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "visrep_gemm_bf16_resid_stats": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp]),
    "visrep_attention_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "visrep_groupnorm_workspace_bytes": (_sz, [_i, _i, _i]),
    "visrep_groupnorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "visrep_im2col3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "visrep_conv_gn_supported": (_i, [_i, _i, _i, _i]),
    "visrep_conv_gn_supported_epi": (_i, [_i, _i, _i, _i, _i]),
    "visrep_set_conv_halo_tile": (_i, [_i]),
    "visrep_conv3x3_c8_supported": (_i, [_i, _i, _i, _i]),
    "visrep_conv3x3_c8_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "visrep_conv_gn_partial_bytes": (_sz, [_i, _i, _i]),
    "visrep_conv3x3_bf16_gn": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "visrep_groupnorm_from_partials": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "visrep_conv3x3_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "visrep_conv3x3_halo_supported": (_i, [_i, _i, _i, _i, _i]),
    "visrep_conv3x3_bf16_halo": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "visrep_groupnorm_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "visrep_groupnorm_stats_from_partials": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "visrep_groupnorm_table_from_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "visrep_geglu": (_i, [_vp, _i, _vp, _i, _l, _i, _vp]),
    "visrep_softmax_rows": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _vp]),
    "visrep_nchw_to_tokens": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "visrep_resample_u8": (_i, [_vp, _vp, _l, _i, _i, _l, _l, _l, _l, _vp, _vp, _i, _vp]),
    "visrep_u8hwc_to_chw_norm": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _i, _vp]),
    "visrep_preprocess_u8_batch": (_i, [_vp, _i, C.c_long, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _i, _vp]),
    "visrep_resize_bilinear": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "visrep_sd_noisy_latents": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "visrep_mean_groups": (_i, [_vp, _vp, _i, _i, _l, _vp]),
    "visrep_im2col": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "visrep_cls_rows": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "visrep_cast_f32_bf16": (_i, [_vp, _vp, _l, _vp]),
    "visrep_vit_workspace_bytes": (_sz, [C.POINTER(VitConfig), _i]),
    "visrep_vit_forward": (_i, [C.POINTER(VitConfig), C.POINTER(VitWeights), _vp, _i, _vp, _i, _i, _vp, _vp]),
    "visrep_gemm_f32": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _i, C.POINTER(C.c_long), _vp]),
    "visrep_layernorm_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "visrep_softmax_rows_f32": (_i, [_vp, _i, _l, _i, _vp]),
    "visrep_vit_f32_workspace_bytes": (_sz, [C.POINTER(VitConfig), _i]),
    "visrep_vit_forward_f32": (_i, [C.POINTER(VitConfig), C.POINTER(VitWeights), _vp, _vp, _i, _i, _vp, _vp]),
    "visrep_split_bf16_planes": (_i, [_vp, _i, C.c_long, _i, _i, _vp, _vp]),
    "visrep_gemm_f32_split": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "visrep_vit_f32_split_supported": (_i, [C.POINTER(VitConfig)]),
    "visrep_vit_forward_f32_split": (_i, [C.POINTER(VitConfig), C.POINTER(VitWeights), C.POINTER(VitWeights), _i, _vp, _vp, _i, _i, _vp, _vp]),
    "visrep_debug_f32_attention": (_i, [_i]),
    "visrep_im2col3x3_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "visrep_groupnorm_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, _i, _vp]),
    "visrep_gram_pairs_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "visrep_row_rnorm_f32": (_i, [_vp, _l, _i, _f, _vp, _vp]),
    "visrep_masked_nn_min_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "visrep_device_cu_count": (_i, []),
    "visrep_set_reserved_cus": (_i, [_i]),
    "visrep_set_gemm_walk": (_i, [_i]),
    "visrep_debug_routes": (_i, [C.POINTER(C.c_long), _i]),
    "visrep_debug_mfma_probe": (_i, [_i, _i, _vp, C.POINTER(C.c_double), _vp]),
    "visrep_set_xcd_balance": (_i, [_i]),
    "visrep_debug_xcd_balance": (_i, [C.POINTER(C.c_float), C.POINTER(C.c_uint)]),
    "visrep_debug_xcd_split": (_i, [C.POINTER(C.c_float), _i, _i, C.POINTER(C.c_int)]),
    "visrep_mutual_nn_distance": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "visrep_ascore_workspace_bytes": (_sz, [_i, _i, _i]),
    "visrep_ascore_maxcos": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "visrep_ascore_row_scale": (_i, [_vp, C.c_long, _i, _i, _vp, _vp]),
    "visrep_ascore_maxcos_scaled": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "visrep_ascore_refarith_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "visrep_ascore_maxcos_refarith": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "visrep_vit_forward_cpu": (_i, [C.POINTER(VitConfig), C.POINTER(VitWeights), _vp, _vp, _i, _i, _i]),
    "visrep_ascore_maxcos_cpu": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "visrep_cscore_transfer_cpu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i]),
    "visrep_pck_count_cpu": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(C.c_float), _vp]),
    "visrep_cscore_transfer": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _vp]),
    "visrep_cscore_transfer_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "visrep_pck_count": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(C.c_float), _vp, _vp]),
    "visrep_jpeg_info": (_i, [_vp, _sz, C.POINTER(JpegInfo)]),
    "visrep_jpeg_entropy_decode": (_i, [_vp, _sz, _vp, _vp]),
    "visrep_jpeg_reconstruct": (_i, [_vp, _vp, _vp, _i, _l, _l, _vp, _vp, _vp]),
}

_lock = threading.Lock()
_lib = None


def load(build_if_missing: bool = True):
    """Load (building first if needed and possible) the HIP library; raises RuntimeError when impossible."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if build_if_missing and not os.environ.get("VISREP_LIB"):
            # (re)build when the library is missing OR older than its sources (content hash, see build.py); builders in other
            # processes - the ranks of one launch - are serialised there with a file lock.  Without hipcc an existing library is
            # used as it is (a deployment box), and a missing one is the error below.
            from . import build
            if build.have_sources() and (os.path.exists(build._hipcc_or_none() or "") or not os.path.exists(LIB_PATH)):
                build.build_lib()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m law_of_vision_representation_in_mllms_amd.build` "
                               "(there is no CPU fallback for the scoring path)")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
        lib.visrep_version.restype = C.c_int
        got = lib.visrep_version()
        if got != ABI_VERSION:                                  # a stale library (VISREP_LIB, a leftover build) would misread the arguments
            raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                               "(`python -m law_of_vision_representation_in_mllms_amd.build --force`)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


ROUTES = ("gemm_256", "gemm_128", "gemm_tail", "splitk", "conv_256", "conv_128", "conv_128_gn", "attn", "attn_wide", "attn_cls", "conv_halo", "conv_c8")


def xcd_balance() -> dict:
    """{"on": bool, "rel": [8 floats: smoothed time per round of tiles of each XCD relative to the mean], "updates": measurements folded in}
    of the current device's XCD-weighted tile split (visrep_debug_xcd_balance)."""
    rel = (C.c_float * 8)()
    upd = C.c_uint(0)
    on = load().visrep_debug_xcd_balance(rel, C.byref(upd))
    return {"on": bool(on), "rel": [round(float(v), 4) for v in rel], "updates": int(upd.value)}


def routes(reset: bool = False) -> dict:
    """This thread's launch counters per kernel family (visrep_debug_routes): {name: count}."""
    buf = (C.c_long * len(ROUTES))()
    n = load().visrep_debug_routes(buf, int(reset))
    assert n == len(ROUTES), "ROUTES is out of step with VISREP_ROUTE_COUNT"
    return dict(zip(ROUTES, list(buf)))


def last_error() -> str:
    buf = C.create_string_buffer(256)
    load().visrep_last_error(buf, 256)
    return buf.value.decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libvisrep_hip {what} failed (code {rc}): {last_error()}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible (torch.cuda.is_available() is False): the scoring path has no CPU fallback")
    return load()


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())
