"""ctypes binding of ``libss_hotpath.so`` (C ABI declared in ``include/ss_hotpath.h``).

There is no fallback: if the shared library has not been built (``python -m silent_speech_amd.build``
or ``__graft_entry__.build()``), every product entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SS_HOTPATH_LIB selects the diagnostic build (tools/stamp_report.py); the product always loads the in-tree library
LIB_PATH = os.environ.get("SS_HOTPATH_LIB", os.path.join(_HERE, "libss_hotpath.so"))

_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_l = C.c_long
_u64 = C.c_uint64
_d = C.c_double

# name -> argtypes (restype is always int unless listed in _RESTYPES)
SIGNATURES = {
    "ss_abi_version": [],
    "ss_status_string": [_i],
    "ss_feature_fuse": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "ss_feature_fuse_stream": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "ss_roi_norm": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "ss_im2col3x3": [_vp, _i, _i, _i, _i, _vp, _i, _vp],
    "ss_relu_pool2": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "ss_relu_mean": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "ss_mask_scale": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "ss_col2im3x3": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "ss_pool2_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ss_roi_crop_idx": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ss_roi_cnn_fwd": [_vp, _i, _i, _i, _i] + [_vp] * 8 + [_i, _vp, _i, _vp],
    "ss_roi_cnn_fwd_stash": [_vp, _i, _i, _i, _i] + [_vp] * 8 + [_i, _vp, _i] + [_vp] * 6 + [_vp, _vp],
    "ss_roi_cnn_set_max_workgroups": [_i],
    "ss_roi_active_frames": [_vp, _i, _i, _vp, _vp, _i, _i, _vp],
    "ss_roi_cnn_fwd_frames": [_vp, _i, _i, _i, _i] + [_vp] * 8 + [_i, _vp, _i] + [_vp] * 6 + [_vp, _vp, _vp],
    "ss_roi_cnn_bwd_frames": [_vp, _i, _i, _i, _i] + [_vp] * 8 + [_i] + [_vp] * 6 + [_vp, _vp, _i] + [_vp] * 8 + [_vp, _vp],
    "ss_roi_cnn_stash_size": [_i, _i, _vp],
    "ss_roi_cnn_bwd": [_vp, _i, _i, _i, _i] + [_vp] * 8 + [_i] + [_vp] * 6 + [_vp, _vp, _i] + [_vp] * 8 + [_vp],
    "ss_gemm_f32": [_i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp],
    "ss_gemm_f32_batched": [_i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i,
                            _l, _l, _l, _l, _l, _vp],
    "ss_gemm_bf16_batched": [_i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _l, _l, _l, _l,
                             _vp],
    "ss_c5_conv1_fwd": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv_fwd": [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv_last_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "ss_c5_conv_wgrad": [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "ss_c5_conv_dgrad": [_i, _vp, _vp, _i, _vp, _vp, _vp],
    "ss_c5_conv_last_wgrad": [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv_last_dgrad": [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "ss_c5_conv1_wgrad": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv2_dgrad_conv1_wgrad": [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv12_fwd": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv_last_fwd_feat": [_vp, _i, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv_last_wgrad_df": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _l, _vp],
    "ss_c5_conv_wgrad_ws": [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _l, _vp],
    "ss_c5_conv2_wgrad_rc_ws": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _l, _vp],
    "ss_c5_conv_last_dgrad_df": [_vp, _vp, _i, _vp, _vp, _vp],
    "ss_c5_conv12_fwd_i1": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv2_dgrad_conv1_wgrad_i1": [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_c5_conv2_wgrad_rc": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "ss_gru_bf16_prep": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "ss_gru_bf16_ws_bytes": [_i, _i, _vp],
    "ss_gru_bf16_sync_bytes": [_i, _i, _i, _vp],
    "ss_gru_bf16_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _u64, _u64, _vp, _vp, _l, _vp],
    "ss_gru_bf16_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp],
    "ss_cvt_bf16_rows": [_vp, _i, _vp, _i, _l, _i, _f, _u64, _u64, _vp],
    "ss_gemm_splitk_ws_floats": [_i, _i, _i, _i, _i, _vp],
    "ss_gemm_splitk_reduce": [_vp, _i, _i, _i, _i, _i, _vp, _i, _l, _vp],
    "ss_gemm_splitk_group_ws_floats": [_vp, _i, _vp],
    "ss_gemm_f32_splitk_group": [_vp, _i, _vp, _l, _i, _vp],
    "ss_gemm_bf16_splitk_group_ws_floats": [_vp, _i, _vp],
    "ss_gemm_bf16_splitk_group": [_vp, _i, _vp, _l, _vp],
    "ss_colsum_f32": [_vp, _i, _i, _i, _vp, _vp],
    "ss_zero_f32x2": [_vp, _l, _vp, _l, _vp],
    "ss_train_prologue": [_vp, _l, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "ss_batch_gather_f32": [_vp, _i, _vp, _l, _vp, _vp, _f, _u64, _vp, _vp],
    "ss_batch_gather_u8": [_vp, _i, _vp, _l, _vp, _vp],
    "ss_crop_gray_resize": [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp],
    "ss_ring_tick": [_vp, _i, _vp, _vp],
    "ss_ring_push": [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_ring_window_map": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ss_mouth_openness": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "ss_mouth_gate": [_vp, _i, _vp, _d, _d, _d, _vp, _vp, _vp],
    "ss_clip_gate": [_vp, _i, _vp, _vp, _d, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_gru_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "ss_gru_fwd_drop": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _f, _u64, _u64, _vp, _vp],
    "ss_gru_sync_bytes": [_i, _i, _i, _vp],
    "ss_gru_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp],
    "ss_gru_bias_grad": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ss_attn_pool_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "ss_attn_pool_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "ss_tail_fwd": [_vp] * 11 + [_i] * 5 + [_f, _f, _u64, _u64, _f, _f] + [_vp] * 10 + [_vp],
    "ss_tail_bwd": [_vp] * 11 + [_i] * 5 + [_f, _u64, _u64] + [_vp] * 7 + [_vp],
    "ss_layernorm_fwd": [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp],
    "ss_layernorm_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "ss_dropout": [_vp, _vp, _l, _f, _u64, _u64, _vp, _vp],
    "ss_ce_ls_fwd_bwd": [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    "ss_softmax_topk": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "ss_sumsq_f32": [_vp, _l, _vp, _vp],
    "ss_adam_clip": [_vp, _vp, _vp, _vp, _l, _vp, _f, _f, _f, _f, _f, _f, _i, _vp],
    "ss_copy_rows_f32": [_vp, _i, _vp, _i, _i, _i, _vp],
}
_RESTYPES = {"ss_status_string": C.c_char_p}

_lib = None


def load():
    """Load the shared library and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"silent_speech_amd: {LIB_PATH} is missing. Build the HIP extension first "
            "(python -m silent_speech_amd.build). There is no CPU/PyTorch fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


class StashSizes(tuple):
    """Per-frame sizes of the six ROI-CNN stash arrays (a1 floats, a2 floats, i1 bytes, i2 bytes, m3 bytes, feat floats) as the
    library reports them; ``.ptr`` is the host int[6] the forward / backward entry points check against their own layout."""

    def __new__(cls, values):
        self = super().__new__(cls, values)
        self._arr = (C.c_int * 6)(*values)
        return self

    @property
    def ptr(self):
        return C.cast(self._arr, C.c_void_p)


def cnn_stash_sizes(H: int, W: int) -> StashSizes:
    arr = (C.c_int * 6)()
    st = load().ss_roi_cnn_stash_size(H, W, arr)
    if st != 0:
        raise RuntimeError(f"ROI size {H}x{W} is not one the CNN kernels are built for")
    return StashSizes(list(arr))


def gru_sync_bytes(B: int, T: int, H: int) -> int:
    """Bytes of the zero-initialised sync workspace the multi-CU recurrence wants for this shape (0: not used)."""
    n = C.c_long(0)
    st = load().ss_gru_sync_bytes(B, T, H, C.byref(n))
    if st != 0:
        raise RuntimeError(f"ss_gru_sync_bytes({B}, {T}, {H}) -> {st}")
    return n.value


class GemmProblem(C.Structure):
    """ss_gemm_problem of include/ss_hotpath.h."""
    _fields_ = [("a_kcontig", C.c_int), ("b_kcontig", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("a_group", C.c_int), ("a_gstride", C.c_int), ("a_off", C.c_int),
                ("B", C.c_void_p), ("ldb", C.c_int), ("b_group", C.c_int), ("b_gstride", C.c_int), ("b_off", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("splits", C.c_int), ("batch", C.c_int),
                ("stride_a", C.c_long), ("stride_b", C.c_long), ("stride_c", C.c_long)]


def gemm_group(problems):
    """-> (ctypes array, n) for ss_gemm_f32_splitk_group / ss_gemm_splitk_group_ws_floats."""
    arr = (GemmProblem * len(problems))(*problems)
    return arr, len(problems)


def gemm_group_ws_floats(problems, bf16: bool = False) -> int:
    arr, n = gemm_group(problems)
    out = C.c_long(0)
    name = "ss_gemm_bf16_splitk_group_ws_floats" if bf16 else "ss_gemm_splitk_group_ws_floats"
    st = getattr(load(), name)(arr, n, C.byref(out))
    if st != 0:
        raise RuntimeError(f"{name} -> {st}")
    return out.value


def gemm_splitk_ws_floats(M: int, N: int, K: int, splits: int, batch: int) -> int:
    n = C.c_long(0)
    st = load().ss_gemm_splitk_ws_floats(M, N, K, splits, batch, C.byref(n))
    if st != 0:
        raise RuntimeError(f"ss_gemm_splitk_ws_floats -> {st}")
    return n.value


def ptr(t):
    """Device pointer of a tensor (or NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def nbytes(t) -> int:
    """Bytes behind ``ptr(t)`` (0 for None): the size arguments the ABI checks workspace layouts against."""
    return 0 if t is None else t.numel() * t.element_size()


def stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing (bench.py): when PROFILE is a dict, every call is bracketed by HIP events on the
# stream the kernel is launched on (torch's current stream) and (start, end) pairs are collected per tag.
PROFILE = None


def call(name: str, *args, tag: str = None):
    lib = load()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = getattr(lib, name)(*args)
        e1.record()
        PROFILE.setdefault(tag or name, []).append((e0, e1))
    else:
        st = getattr(lib, name)(*args)
    if st != 0:
        msg = lib.ss_status_string(st).decode()
        raise RuntimeError(f"{name} failed: {msg} ({st})")
