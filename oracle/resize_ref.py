"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the two OpenCV calls behind the reference's ROI crop:
``cv2.cvtColor(roi, COLOR_BGR2GRAY)`` and ``cv2.resize(roi, (ROI_W, ROI_H))`` with the default INTER_LINEAR
(/root/reference/record_landmarks_official.py:116-118) or INTER_AREA (/root/reference/live_infer_official.py:184-186).

PARITY UNPINNED: OpenCV (opencv-python==4.13.0.90, requirements.txt:17-18) is a third-party dependency that is neither in
/root/reference nor installed in the authoring container, and the reference holds no fixture for it.  What follows restates
OpenCV's published 8-bit algorithms (imgproc color conversion with 15-bit coefficients; resize.cpp's fixed-point linear
pass and DecimateAlpha area pass); an OpenCV wheel may route resize through IPP with +-1 differences, so the contract for
this row is +-1 grey level against cv2 and bit-exactness only between this file and the HIP kernel.
"""
from __future__ import annotations

import math

import numpy as np


def bgr2gray(img: np.ndarray) -> np.ndarray:
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def _sat_short(v: np.float32) -> int:
    return int(np.clip(np.rint(np.float32(v)), -32768, 32767))


def _linear_coord(d: int, ssize: int, dsize: int, area_mode: bool):
    scale, inv = ssize / dsize, dsize / ssize
    if not area_mode:
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = np.float32(f - np.float32(s))
    else:
        s = int(math.floor(d * scale))
        f = np.float32((d + 1) - (s + 1) * inv)
        f = np.float32(0.0) if f <= 0 else np.float32(f - np.float32(math.floor(f)))
    if s < 0:
        f, s = np.float32(0.0), 0
    if s >= ssize - 1:
        f, s = np.float32(0.0), ssize - 1
    return s, _sat_short(np.float32(np.float32(1.0) - f) * np.float32(2048.0)), _sat_short(f * np.float32(2048.0))


def _area_cells(d: int, scale: float, ssize: int):
    """DecimateAlpha entries (source index, float weight) of destination index d."""
    f1 = d * scale
    f2 = f1 + scale
    cell = min(scale, ssize - f1)
    s1, s2 = math.ceil(f1), min(math.floor(f2), ssize - 1)
    s1 = min(s1, s2)
    out = []
    if s1 - f1 > 1e-3:
        out.append((s1 - 1, np.float32((s1 - f1) / cell)))
    for s in range(s1, s2):
        out.append((s, np.float32(1.0 / cell)))
    if f2 - s2 > 1e-3:
        out.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
    return out


def resize_gray(g: np.ndarray, roi_h: int, roi_w: int, interp: str) -> np.ndarray:
    sh, sw = g.shape
    scx, scy = sw / roi_w, sh / roi_h
    out = np.zeros((roi_h, roi_w), np.uint8)
    gi = g.astype(np.int64)
    if not (interp == "area" and scx >= 1.0 and scy >= 1.0):
        am = interp == "area"
        xs = [_linear_coord(dx, sw, roi_w, am) for dx in range(roi_w)]
        ys = [_linear_coord(dy, sh, roi_h, am) for dy in range(roi_h)]
        for dy, (sy, b0, b1) in enumerate(ys):
            sy1 = min(sy + 1, sh - 1)
            for dx, (sx, a0, a1) in enumerate(xs):
                sx1 = min(sx + 1, sw - 1)
                S0 = int(gi[sy, sx]) * a0 + int(gi[sy, sx1]) * a1
                S1 = int(gi[sy1, sx]) * a0 + int(gi[sy1, sx1]) * a1
                out[dy, dx] = np.clip((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, 0, 255)
        return out
    isx, isy = int(scx), int(scy)
    if isx == scx and isy == scy:
        for dy in range(roi_h):
            for dx in range(roi_w):
                sm = int(gi[dy * isy:(dy + 1) * isy, dx * isx:(dx + 1) * isx].sum())
                out[dy, dx] = (sm + 2) >> 2 if (isx == 2 and isy == 2) else int(
                    np.clip(np.rint(np.float32(sm) * np.float32(np.float32(1.0) / np.float32(isx * isy))), 0, 255))
        return out
    xcells = [_area_cells(dx, scx, sw) for dx in range(roi_w)]
    ycells = [_area_cells(dy, scy, sh) for dy in range(roi_h)]
    for dy in range(roi_h):
        for dx in range(roi_w):
            total = None
            for sy, beta in ycells[dy]:
                buf = np.float32(0.0)
                for sx, alpha in xcells[dx]:
                    buf = np.float32(buf + np.float32(np.float32(g[sy, sx]) * alpha))
                v = np.float32(beta * buf)
                total = v if total is None else np.float32(total + v)
            out[dy, dx] = int(np.clip(np.rint(total), 0, 255))
    return out


def crop_gray_resize(frame_bgr: np.ndarray, box, roi_h: int, roi_w: int, interp: str) -> np.ndarray:
    """box = (x1, x2, y1, y2, valid) as produced by features_ref.crop_box / ss_roi_crop_idx."""
    x1, x2, y1, y2, valid = (int(v) for v in box)
    if not valid:
        return np.zeros((roi_h, roi_w), np.uint8)
    return resize_gray(bgr2gray(frame_bgr[y1:y2, x1:x2]), roi_h, roi_w, interp)
