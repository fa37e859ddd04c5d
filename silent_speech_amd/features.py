"""Host mirror of the recorder's per-frame numeric helpers, batched on the GPU.

``extract_features`` replaces ``extract_feature`` + ``mouth_width_px``
(/root/reference/record_landmarks_official.py:52-100; live variant live_infer_official.py:141-169) and
``crop_boxes`` the index arithmetic of ``crop_roi`` / ``crop_roi_gray`` (record…:106-114; live…:172-181).
Inputs are the MediaPipe-normalised (x, y) of the selected landmarks as a float32 ``(B,T,K,2)`` device tensor
(the landmark detector itself is upstream of the path).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib as L

# MediaPipe FaceLandmarker indices (data constants of the reference):
# the 88-point official set (record_landmarks_official.py:30-44, sorted) and the 40-point lip set
# (inactive/record_landmarks.py:23-34) that BASELINE.json's synthetic "40-landmark" configs refer to.
FIXED_IDXS_88 = (0, 13, 14, 17, 18, 32, 37, 39, 40, 42, 57, 61, 78, 81, 82, 83, 84, 87, 88, 91, 95, 146, 148, 149, 150,
                 152, 169, 176, 178, 181, 183, 185, 186, 194, 199, 200, 201, 202, 204, 208, 210, 211, 212, 214, 262,
                 267, 269, 270, 273, 287, 291, 308, 310, 311, 312, 313, 314, 317, 318, 321, 322, 324, 335, 364, 375,
                 377, 378, 379, 394, 396, 400, 402, 405, 406, 409, 410, 415, 416, 418, 421, 422, 424, 428, 430, 431,
                 432, 434, 436)
LIP_IDXS_40 = (0, 13, 14, 17, 37, 39, 40, 42, 61, 78, 81, 82, 84, 87, 88, 91, 95, 146, 178, 181, 183, 185, 267, 269,
               270, 291, 308, 310, 311, 312, 314, 317, 318, 321, 324, 375, 402, 405, 409, 415)
LEFT_CORNER, RIGHT_CORNER, UPPER_INNER, LOWER_INNER = 61, 291, 13, 14
MOUTH_W_MIN_PX, MOUTH_W_MAX_PX = 60, 150  # distance gate, record_landmarks_official.py:21-22,185

_VARIANTS = {"record": 0, "live": 1}


def anchor_positions(idxs: Sequence[int]):
    """Positions of landmarks 61 / 291 / 13 / 14 inside an index list."""
    idxs = [int(i) for i in idxs]
    return tuple(idxs.index(a) for a in (LEFT_CORNER, RIGHT_CORNER, UPPER_INNER, LOWER_INNER))


def extract_features(lm: torch.Tensor, w: int, h: int, idxs: Sequence[int] = FIXED_IDXS_88,
                     reset: Optional[torch.Tensor] = None, variant: str = "record", out: Optional[torch.Tensor] = None):
    """(B,T,K,2) float32 cuda -> X (B,T,2K+4), center (B,T,2) float32, fourth (B,T) float64.

    Row layout of X: ``[x0,y0,...,vel,mouth_open_px,mouth_w_px,mouth_aspect]``.  ``reset[b,t] != 0`` means
    ``prev_xy is None`` at that frame (t = 0 always).  ``fourth`` is the function's 4th return value: width + 1e-6
    for the recorder variant (what ``crop_roi`` receives), the width itself for the live variant."""
    if not lm.is_cuda:
        raise RuntimeError("extract_features runs on the HIP device only")
    B, T, K, two = lm.shape
    assert two == 2 and K == len(idxs) and lm.dtype == torch.float32
    lm = lm.contiguous()
    X = out if out is not None else torch.empty(B, T, 2 * K + 4, device=lm.device)
    center = torch.empty(B, T, 2, device=lm.device)
    fourth = torch.empty(B, T, device=lm.device, dtype=torch.float64)
    rs = reset.to(device=lm.device, dtype=torch.uint8).contiguous() if reset is not None else None
    L.call("ss_feature_fuse", lm.data_ptr(), L.ptr(rs), B, T, K, int(w), int(h), *anchor_positions(idxs),
           _VARIANTS[variant], X.data_ptr(), X.stride(1), center.data_ptr(), fourth.data_ptr(), L.stream())
    return X, center, fourth


def crop_boxes(center: torch.Tensor, scale: torch.Tensor, w: int, h: int, variant: str = "record") -> torch.Tensor:
    """(…,2) float32 centres + (…) float64 scales -> (…,5) int32 ``x1, x2, y1, y2, valid`` (bit-exact)."""
    if not center.is_cuda:
        raise RuntimeError("crop_boxes runs on the HIP device only")
    c = center.reshape(-1, 2).contiguous().float()
    s = scale.reshape(-1).contiguous().double()
    box = torch.empty(c.shape[0], 5, device=c.device, dtype=torch.int32)
    L.call("ss_roi_crop_idx", c.data_ptr(), s.data_ptr(), c.shape[0], int(w), int(h), _VARIANTS[variant], box.data_ptr(),
           L.stream())
    return box.view(*center.shape[:-1], 5)


def crop_rois(frames_bgr: torch.Tensor, boxes: torch.Tensor, roi_hw=(48, 96), variant: str = "record") -> torch.Tensor:
    """(N,h,w,3) uint8 BGR frames + (N,5) crop boxes -> (N,ROI_H,ROI_W) uint8 mouth ROIs: BGR2GRAY + resize, INTER_LINEAR
    for the recorder (record_landmarks_official.py:116-118), INTER_AREA for the live script (live_infer_official.py:184-186).
    Frames with an invalid box come back as zeros (the reference skips them)."""
    if not frames_bgr.is_cuda:
        raise RuntimeError("crop_rois runs on the HIP device only")
    f = frames_bgr.contiguous()
    N, h, w, _ = f.shape
    out = torch.empty(N, roi_hw[0], roi_hw[1], device=f.device, dtype=torch.uint8)
    L.call("ss_crop_gray_resize", f.data_ptr(), N, h, w, boxes.reshape(-1, 5).contiguous().data_ptr(), roi_hw[0], roi_hw[1],
           0 if variant == "record" else 1, out.data_ptr(), L.stream())
    return out


def in_distance_band(X: torch.Tensor, K: int) -> torch.Tensor:
    """The recorder's keep-frame gate 60 <= mouth_w <= 150 px (record_landmarks_official.py:185) from X's width column."""
    mw = X[..., 2 * K + 2]
    return (mw >= MOUTH_W_MIN_PX) & (mw <= MOUTH_W_MAX_PX)
