"""ctypes binding of librichsem_msda.so (C ABI: include/richsem_msda.h).

There is no CPU fallback: if the library cannot be loaded the import of anything that needs
it raises, and CPU tensors are rejected exactly like the reference does
(reference models/richsem/ops/src/ms_deform_attn.h:38 "Not implemented on the CPU").
"""
import ctypes
import os

from . import _build

_lib = None

ERR_NAMES = {
    -1: "MSDA_ERR_NULL_POINTER", -2: "MSDA_ERR_BAD_DIMS", -3: "MSDA_ERR_IM2COL_STEP",
    -4: "MSDA_ERR_TOO_LARGE", -5: "MSDA_ERR_MISALIGNED", -6: "MSDA_ERR_NO_DEVICE", -7: "MSDA_ERR_BAD_OPTION",
    -8: "MSDA_ERR_NOT_ON_CPU",
}

ABI_VERSION = 8


def raw_stream(dev=None):
    """hipStream_t (as an int) of torch's current stream on ``dev``: the C call -- building a ``torch.cuda.Stream`` object per launch
    (``torch.cuda.current_stream(dev).cuda_stream``) costs ~4 us of host time, and an eager training step makes ~600 such calls"""
    import torch
    idx = None if dev is None else dev.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """context manager that makes ``dev`` the current device for a library call -- nothing at all when it already is (the usual case:
    ``torch.cuda.device`` costs two device switches and ~8 us of host time per launch)"""
    import torch
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


class WgradProblem(ctypes.Structure):
    """``msda_wgrad_problem`` (include/richsem_msda.h): one weight gradient of a grouped launch"""
    _fields_ = [("dz", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("dbias", ctypes.c_void_p)] + \
        [(k, ctypes.c_int) for k in ("N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad")]

# every symbol include/richsem_msda.h declares
SYMBOLS = [
    "msda_abi_version", "msda_last_error", "msda_set_option", "msda_get_option",
    "msda_profile_enable", "msda_profile_collect", "msda_tiled_plan", "msda_levelsum_plan", "msda_debug_stamps", "msda_debug_stats",
    "msda_forward_f32", "msda_forward_f64", "msda_backward_f32", "msda_backward_f64",
    "msda_forward_bf16", "msda_backward_bf16", "msda_forward_cpu", "msda_backward_cpu",
    "msda_prep_forward_f32", "msda_prep_forward_f64", "msda_prep_backward_f32", "msda_prep_backward_f64",
    "msda_prep_forward_bf16", "msda_prep_backward_bf16", "msda_forward_prep_f32", "msda_forward_prep_f64", "msda_forward_prep_bf16",
    "msda_mask_rows_f32", "msda_mask_rows_f64", "msda_mask_rows_bf16",
    "msda_dn_indices_i64", "msda_dn_attn_mask_u8", "msda_topk_f32", "msda_sine_embed_bf16", "msda_narrow_linear_backward_bf16", "msda_box_refine_forward", "msda_box_refine_backward", "msda_box_refine_backward_ref", "msda_box_pair_loss_f32", "msda_focal_pos_sum_f32", "msda_roi_align_forward_f32", "msda_roi_align_forward_f64",
    "msda_ffn_pack_w2_bf16", "msda_ffn_forward_bf16", "msda_ffn_debug_stamps", "msda_ffn_forward_train_bf16", "msda_ffn_ln_backward_bf16", "msda_add_layernorm_forward_bf16", "msda_lin256_pack_bf16", "msda_lin256_forward_bf16", "msda_lin256_pack_f32", "msda_lin256_forward_f32", "msda_lin256_forward_stacked_bf16",
    "msda_attn_workspace_bytes", "msda_attn_forward_bf16", "msda_attn_backward_bf16",
    "msda_matcher_cost_f32", "msda_matcher_cost_f64", "msda_focal_neg_sum_f32", "msda_focal_neg_grad_f32", "msda_attnpool_core_f32", "msda_attnpool_core_f64",
    "msda_cls_packed_elems", "msda_cls_pack", "msda_cls_max_scores",
    "msda_conv_set_tiling", "msda_conv_set_ring", "msda_conv_dgrad_fused_bf16", "msda_conv_packed_elems", "msda_conv_pack_weight", "msda_conv_forward_bf16", "msda_conv_dgrad_bf16", "msda_conv_forward_workspace_bytes", "msda_conv_forward_ws_bf16", "msda_conv_dgrad_workspace_bytes", "msda_conv_dgrad_ws_bf16", "msda_pool_nhwc_bf16", "msda_groupnorm8_nhwc_bf16", "msda_groupnorm8_backward_nhwc_bf16", "msda_conv_wgrad_workspace_bytes", "msda_conv_wgrad_bf16", "msda_conv_set_wgrad_ring", "msda_conv_wgrad_group_workspace_bytes", "msda_conv_wgrad_group_bf16",
]


class ProfileRecord(ctypes.Structure):
    """msda_profile_record (include/richsem_msda.h)"""
    _fields_ = [("kind", ctypes.c_int), ("variant", ctypes.c_int), ("dtype_bytes", ctypes.c_int),
                ("N", ctypes.c_int), ("S", ctypes.c_int), ("M", ctypes.c_int), ("D", ctypes.c_int),
                ("L", ctypes.c_int), ("Lq", ctypes.c_int), ("P", ctypes.c_int), ("kernel_ms", ctypes.c_float)]


def lib_path():
    """the product library -- or, for diagnostic runs only, the library RICHSEM_MSDA_LIB names (ablation builds are loaded from where they
    were built: nothing ever overwrites the product .so)"""
    return os.environ.get("RICHSEM_MSDA_LIB") or _build.LIB_PATH


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m richsem_amd._build` (hipcc --offload-arch=gfx950). "
            "richsem_amd has no CPU or PyTorch fallback for MSDeformAttn.")
    import torch  # noqa: F401  -- make torch's libamdhip64 (same SONAME) the one HIP runtime of this process
    L = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.msda_abi_version.restype = ci
    L.msda_last_error.restype = ctypes.c_char_p
    L.msda_set_option.argtypes = [ctypes.c_char_p, ci]
    L.msda_set_option.restype = ci
    L.msda_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci)]
    L.msda_get_option.restype = ci
    L.msda_tiled_plan.argtypes = [ci] * 7 + [vp, vp, ctypes.POINTER(ci)]
    L.msda_tiled_plan.restype = ci
    L.msda_levelsum_plan.argtypes = [ci] * 7 + [vp, vp, ctypes.POINTER(ci)]
    L.msda_levelsum_plan.restype = ci
    L.msda_debug_stamps.argtypes = [vp]
    L.msda_debug_stamps.restype = ci
    L.msda_debug_stats.argtypes = [vp]
    L.msda_debug_stats.restype = ci
    L.msda_profile_enable.argtypes = [ci]
    L.msda_profile_enable.restype = ci
    L.msda_profile_collect.argtypes = [ctypes.POINTER(ProfileRecord), ci, ctypes.POINTER(ci)]
    L.msda_profile_collect.restype = ci
    i64 = ctypes.c_int64
    L.msda_dn_indices_i64.argtypes = [vp, ci, i64, ci, i64, vp, vp, vp]
    L.msda_dn_indices_i64.restype = ci
    L.msda_dn_attn_mask_u8.argtypes = [vp, i64, i64, i64, vp]
    L.msda_dn_attn_mask_u8.restype = ci
    L.msda_topk_f32.argtypes = [vp, ci, ci, ci, vp, vp, vp]
    L.msda_topk_f32.restype = ci
    L.msda_sine_embed_bf16.argtypes = [vp, ci, ci, ci, ci, ctypes.c_float, vp, vp]
    L.msda_sine_embed_bf16.restype = ci
    L.msda_narrow_linear_backward_bf16.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, vp]
    L.msda_narrow_linear_backward_bf16.restype = ci
    L.msda_box_refine_forward.argtypes = [vp, ci, vp, ctypes.c_float, ctypes.c_int64, vp, vp]
    L.msda_box_refine_forward.restype = ci
    L.msda_box_refine_backward.argtypes = [vp, vp, ctypes.c_int64, vp, ci, vp]
    L.msda_box_refine_backward.restype = ci
    L.msda_box_refine_backward_ref.argtypes = [vp, vp, ctypes.c_int64, vp, ci, vp, ctypes.c_float, vp, vp]
    L.msda_box_refine_backward_ref.restype = ci
    L.msda_box_pair_loss_f32.argtypes = [vp, vp, vp, ci, ctypes.c_float, ctypes.c_float, vp, vp, vp]
    L.msda_box_pair_loss_f32.restype = ci
    L.msda_focal_pos_sum_f32.argtypes = [vp, vp, ci, ctypes.c_float, vp, vp, vp]
    L.msda_focal_pos_sum_f32.restype = ci
    for sfx in ("f32", "f64"):
        f = getattr(L, "msda_roi_align_forward_" + sfx)
        f.argtypes = [vp, vp] + [ci] * 7 + [ctypes.c_double, ci, ci, vp, vp]
        f.restype = ci
    for sfx in ("f32", "f64"):
        f = getattr(L, "msda_matcher_cost_" + sfx)
        f.argtypes = [vp] * 5 + [ci] * 3 + [i64] + [ctypes.c_double] * 4 + [vp, vp]
        f.restype = ci
    L.msda_focal_neg_sum_f32.argtypes = [vp, vp, i64, ci, ctypes.c_float, vp, ci, ctypes.POINTER(ci), vp]
    L.msda_focal_neg_sum_f32.restype = ci
    L.msda_focal_neg_grad_f32.argtypes = [vp, vp, i64, ci, ctypes.c_float, vp, vp, vp]
    L.msda_focal_neg_grad_f32.restype = ci
    for sfx in ("f32", "f64"):
        f = getattr(L, "msda_attnpool_core_" + sfx)
        f.argtypes = [vp] * 4 + [ci] * 5 + [vp, vp]
        f.restype = ci
    L.msda_cls_packed_elems.argtypes = [ci, ctypes.POINTER(i64)]
    L.msda_cls_packed_elems.restype = ci
    L.msda_cls_pack.argtypes = [vp, ci, vp, ci, vp, vp]
    L.msda_cls_pack.restype = ci
    L.msda_cls_max_scores.argtypes = [vp, ci, vp, ci, ci, ci, ctypes.c_float, ci, vp, vp]
    L.msda_cls_max_scores.restype = ci
    L.msda_conv_pack_weight.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    L.msda_conv_pack_weight.restype = ci
    L.msda_conv_set_tiling.argtypes = [ci, ci]
    L.msda_conv_set_tiling.restype = ci
    L.msda_conv_set_ring.argtypes = [ci]
    L.msda_conv_set_ring.restype = ci
    L.msda_conv_dgrad_fused_bf16.argtypes = [vp, vp] + [ci] * 11 + [vp, vp, vp, vp, vp]
    L.msda_conv_dgrad_fused_bf16.restype = ci
    L.msda_conv_packed_elems.argtypes = [ci] * 4 + [ctypes.POINTER(i64)]
    L.msda_conv_packed_elems.restype = ci
    L.msda_conv_forward_bf16.argtypes = [vp] * 5 + [ci] * 10 + [vp, vp]
    L.msda_conv_forward_bf16.restype = ci
    L.msda_conv_dgrad_bf16.argtypes = [vp, vp] + [ci] * 11 + [vp, vp]
    L.msda_conv_dgrad_bf16.restype = ci
    L.msda_conv_forward_workspace_bytes.argtypes = [ci] * 9 + [vp]
    L.msda_conv_forward_workspace_bytes.restype = ci
    L.msda_conv_forward_ws_bf16.argtypes = [vp] * 5 + [ci] * 10 + [vp, vp, vp]
    L.msda_conv_forward_ws_bf16.restype = ci
    L.msda_conv_dgrad_workspace_bytes.argtypes = [ci] * 11 + [vp]
    L.msda_conv_dgrad_workspace_bytes.restype = ci
    L.msda_conv_dgrad_ws_bf16.argtypes = [vp, vp] + [ci] * 11 + [vp, vp, vp]
    L.msda_conv_dgrad_ws_bf16.restype = ci
    L.msda_pool_nhwc_bf16.argtypes = [vp] + [ci] * 8 + [vp, vp]
    L.msda_pool_nhwc_bf16.restype = ci
    L.msda_groupnorm8_nhwc_bf16.argtypes = [vp, vp, vp, ctypes.c_float, ci, ci, ci, vp, vp, vp, vp]
    L.msda_groupnorm8_nhwc_bf16.restype = ci
    L.msda_groupnorm8_backward_nhwc_bf16.argtypes = [vp, vp, vp, ctypes.c_float, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    L.msda_groupnorm8_backward_nhwc_bf16.restype = ci
    L.msda_conv_wgrad_bf16.argtypes = [vp, vp] + [ci] * 9 + [vp, vp, vp, ci, vp, vp]
    L.msda_conv_wgrad_bf16.restype = ci
    L.msda_conv_wgrad_workspace_bytes.argtypes = [ci] * 9 + [ctypes.POINTER(i64)]
    L.msda_conv_wgrad_workspace_bytes.restype = ci
    L.msda_conv_set_wgrad_ring.argtypes = [ci]
    L.msda_conv_set_wgrad_ring.restype = ci
    L.msda_conv_wgrad_group_workspace_bytes.argtypes = [ctypes.POINTER(WgradProblem), ci, ctypes.POINTER(i64)]
    L.msda_conv_wgrad_group_workspace_bytes.restype = ci
    L.msda_conv_wgrad_group_bf16.argtypes = [ctypes.POINTER(WgradProblem), ci, vp, vp]
    L.msda_conv_wgrad_group_bf16.restype = ci
    L.msda_ffn_debug_stamps.argtypes = [vp]
    L.msda_ffn_debug_stamps.restype = ci
    L.msda_ffn_pack_w2_bf16.argtypes = [vp, ci, ci, vp, vp]
    L.msda_ffn_pack_w2_bf16.restype = ci
    L.msda_ffn_forward_bf16.argtypes = [vp] * 7 + [ctypes.c_float, ci, ci, ci, vp, vp]
    L.msda_ffn_forward_bf16.restype = ci
    L.msda_ffn_forward_train_bf16.argtypes = [vp] * 7 + [ctypes.c_float, ci, ci, ci, vp, vp, vp, vp]
    L.msda_ffn_forward_train_bf16.restype = ci
    L.msda_ffn_ln_backward_bf16.argtypes = [vp] * 4 + [ci, ci] + [vp] * 5
    L.msda_ffn_ln_backward_bf16.restype = ci
    L.msda_add_layernorm_forward_bf16.argtypes = [vp, vp, vp, vp, ctypes.c_float, ci, ci, vp, vp, vp, vp]
    L.msda_add_layernorm_forward_bf16.restype = ci
    L.msda_lin256_pack_bf16.argtypes = [vp, ci, ci, vp, vp]
    L.msda_lin256_pack_bf16.restype = ci
    L.msda_lin256_forward_bf16.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
    L.msda_lin256_forward_bf16.restype = ci
    L.msda_lin256_pack_f32.argtypes = [vp, ci, ci, vp, vp]
    L.msda_lin256_pack_f32.restype = ci
    L.msda_lin256_forward_f32.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
    L.msda_lin256_forward_f32.restype = ci
    L.msda_lin256_forward_stacked_bf16.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp]
    L.msda_lin256_forward_stacked_bf16.restype = ci
    L.msda_attn_workspace_bytes.argtypes = [ci, ci, ci]
    L.msda_attn_workspace_bytes.restype = ctypes.c_int64
    L.msda_attn_forward_bf16.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, vp, vp, vp]
    L.msda_attn_forward_bf16.restype = ci
    L.msda_attn_backward_bf16.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, ci, vp, ci, vp, ci, vp, vp]
    L.msda_attn_backward_bf16.restype = ci
    for sfx in ("f32", "f64", "bf16"):
        f = getattr(L, "msda_forward_" + sfx)
        f.argtypes = [vp] * 5 + [ci] * 8 + [vp, vp, vp, vp]
        f.restype = ci
        g = getattr(L, "msda_backward_" + sfx)
        g.argtypes = [vp] * 6 + [ci] * 8 + [vp, vp, vp, vp, vp, vp]
        g.restype = ci
    i64 = ctypes.c_int64
    for sfx in ("f32", "f64", "bf16"):
        f = getattr(L, "msda_prep_forward_" + sfx)
        f.argtypes = [vp, i64, vp, i64, vp, ci, vp] + [ci] * 5 + [vp, vp, vp]
        f.restype = ci
        g = getattr(L, "msda_prep_backward_" + sfx)
        g.argtypes = [vp, vp, vp, vp, i64, vp, ci, vp] + [ci] * 5 + [vp, i64, vp, i64, vp, vp]
        g.restype = ci
    for sfx in ("f32", "f64", "bf16"):
        f = getattr(L, "msda_forward_prep_" + sfx)
        f.argtypes = [vp, vp, vp, vp, i64, vp, i64, vp, ci] + [ci] * 8 + [vp, vp, vp, vp, vp, vp]
        f.restype = ci
    for sfx in ("f32", "f64", "bf16"):
        f = getattr(L, "msda_mask_rows_" + sfx)
        f.argtypes = [vp, vp, i64, ci, vp]
        f.restype = ci
    if L.msda_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {L.msda_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = L
    return L


def last_error():
    return load().msda_last_error().decode("utf-8", "replace")


def check(rc):
    """Turn a C-ABI return code into a RuntimeError (the reference raises RuntimeError via AT_ASSERTM)."""
    if rc == 0:
        return
    name = ERR_NAMES.get(rc, f"hipError {rc}" if rc > 0 else f"error {rc}")
    raise RuntimeError(f"richsem_msda: {last_error()} [{name}]")


def set_option(key, value):
    check(load().msda_set_option(key.encode(), int(value)))


def get_option(key):
    v = ctypes.c_int(0)
    check(load().msda_get_option(key.encode(), ctypes.byref(v)))
    return v.value


def profile_enable(capacity):
    """Log up to `capacity` calls (0 = off): HIP events around each call's main kernel, on its stream."""
    check(load().msda_profile_enable(int(capacity)))


def profile_collect(max_records=65536):
    """Synchronise the logged events; returns a list of dicts (kind 'fwd'/'bwd', variant, dims, kernel_ms)."""
    buf = (ProfileRecord * max_records)()
    n = ctypes.c_int(0)
    check(load().msda_profile_collect(buf, max_records, ctypes.byref(n)))
    out = []
    for r in buf[:n.value]:
        out.append(dict(kind="bwd" if r.kind else "fwd", variant=r.variant, dtype_bytes=r.dtype_bytes, N=r.N, S=r.S,
                        M=r.M, D=r.D, L=r.L, Lq=r.Lq, P=r.P, kernel_ms=float(r.kernel_ms)))
    return out


def levelsum_plan(N, S, M, D, L, Lq, P, shapes, lsi):
    """What the level-sum backward kernel would take for a problem (host-only)."""
    import numpy as np
    sh = np.ascontiguousarray(shapes, dtype=np.int64)
    ls = np.ascontiguousarray(lsi, dtype=np.int64)
    info = (ctypes.c_int * 8)()
    check(load().msda_levelsum_plan(N, S, M, D, L, Lq, P, sh.ctypes.data, ls.ctypes.data, info))
    keys = ("levels_mask", "windows", "slices", "lds_bytes", "grid", "max_rows")
    return dict(zip(keys, [int(v) for v in info]))


def tiled_plan(N, S, M, D, L, Lq, P, shapes, lsi):
    """Launch plan of the LDS-window kernels for a problem (host-only).  shapes: [(H, W)], lsi: [start]."""
    import numpy as np
    sh = np.ascontiguousarray(shapes, dtype=np.int64)
    ls = np.ascontiguousarray(lsi, dtype=np.int64)
    info = (ctypes.c_int * 8)()
    check(load().msda_tiled_plan(N, S, M, D, L, Lq, P, sh.ctypes.data, ls.ctypes.data, info))
    keys = ("applicable", "GY", "GX", "phases", "lds_bytes", "grid", "margin", "max_region_queries")
    return dict(zip(keys, [int(v) for v in info]))
