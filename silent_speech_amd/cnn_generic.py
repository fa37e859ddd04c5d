"""ROI normalise + TinyROICNN for frame sizes outside the fused kernels' set (csrc/roi_cnn_generic.hip): the layer-by-layer form.

``engine.Workspace`` builds a ``GenericCnn`` when ``ss_roi_cnn_stash_size`` does not know the ROI size (the fused, LDS-resident
kernels exist for 64x64, 48x96 and 32x32); ``forward`` / ``backward`` below are then called in place of ``ss_roi_cnn_fwd_stash`` /
``ss_roi_cnn_bwd``.  Every 3x3 convolution is im2col + the f32 MFMA GEMM against ``nn.Conv2d``'s weight as it stands; the im2col
rows are kept for the weight gradients (MI355X has 288 GB of HBM: 96x96 frames at config 2's 7 680 frames per step are 20 GB).
Frames are independent, so the work is cut into chunks whose GEMMs stay inside the launch grid's limits.  Replaces
/root/reference/train_model_official.py:209-229 + :286-291 and their autograd for ANY ROI_H x ROI_W (multiples of 4); an order of
magnitude slower than the fused kernels -- the correctness path for unusual shapes, not a tuned one."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib as L

C1, C2, C3 = 8, 16, 24
MAX_GEMM_ROWS = 4 * 1024 * 1024  # rows of one im2col GEMM (grid.y = rows / 128 must stay below 65 536)
_NAMES = ("roi_cnn.net.0.weight", "roi_cnn.net.0.bias", "roi_cnn.net.3.weight", "roi_cnn.net.3.bias",
          "roi_cnn.net.6.weight", "roi_cnn.net.6.bias", "roi_cnn.fc.weight", "roi_cnn.fc.bias")


def fused_supported(H: int, W: int) -> bool:
    import ctypes as C

    arr = (C.c_int * 6)()
    return L.load().ss_roi_cnn_stash_size(H, W, arr) == 0


class GenericCnn:
    def __init__(self, N: int, H: int, W: int, device, train: bool):
        if H % 4 or W % 4 or H < 4 or W < 4:
            raise RuntimeError(f"ROI size {H}x{W}: the CNN needs H and W to be multiples of 4 (two 2x2 max-pools)")
        self.N, self.H, self.W, self.train = N, H, W, train
        self.HW, self.HW2, self.P = H * W, (H // 2) * (W // 2), (H // 4) * (W // 4)
        f32 = dict(device=device, dtype=torch.float32)
        u8 = dict(device=device, dtype=torch.uint8)
        self.chunk = max(1, min(N, MAX_GEMM_ROWS // self.HW))
        # what the backward pass reads again is kept for all N frames; the rest is sized for one chunk
        keep = N if train else self.chunk
        self.xn = torch.empty(self.chunk, self.HW, **f32)
        self.col1 = torch.empty(keep * self.HW, 12, **f32)      # 9 taps + 3 columns of alignment padding
        self.y1 = torch.empty(self.chunk * self.HW, C1, **f32)  # conv output (pixel-major), then d y1
        self.a1 = torch.empty(keep, C1, H // 2, W // 2, **f32)
        self.i1 = torch.empty(keep, C1, H // 2, W // 2, **u8)
        self.col2 = torch.empty(keep * self.HW2, 9 * C1, **f32)
        self.y2 = torch.empty(self.chunk * self.HW2, C2, **f32)
        self.a2 = torch.empty(keep, C2, H // 4, W // 4, **f32)
        self.i2 = torch.empty(keep, C2, H // 4, W // 4, **u8)
        self.col3 = torch.empty(keep * self.P, 9 * C2, **f32)
        self.y3 = torch.empty(self.chunk * self.P, C3, **f32)
        self.feat = torch.empty(N, C3, **f32)
        if train:
            self.m3 = torch.empty(N * self.P, C3, **u8)
            self.dfeat = torch.empty(self.chunk, C3, **f32)
            self.dcol = torch.empty(self.chunk * max(self.P * 9 * C2, self.HW2 * 9 * C1), **f32)
            self.da2 = torch.empty(self.chunk, C2, H // 4, W // 4, **f32)
            self.da1 = torch.empty(self.chunk, C1, H // 2, W // 2, **f32)

    # ------------------------------------------------------------------ forward
    def forward(self, P: Dict[str, torch.Tensor], R: torch.Tensor, standardize: bool, E: int, z_ptr: int, ld_z: int, stash: bool):
        from .engine import gemm

        if stash and not self.train:
            raise RuntimeError("stash=True needs a training workspace")
        s = L.stream()
        w1, b1, w2, b2, w3, b3, wfc, bfc = (P[k].data_ptr() for k in _NAMES)
        H, W, HW, HW2, Pp = self.H, self.W, self.HW, self.HW2, self.P
        for f0 in range(0, self.N, self.chunk):
            n = min(self.chunk, self.N - f0)
            k0 = f0 if self.train else 0  # frame offset into the buffers kept for the backward pass
            col1, a1, i1 = self.col1[k0 * HW:], self.a1[k0:], self.i1[k0:]
            col2, a2, i2 = self.col2[k0 * HW2:], self.a2[k0:], self.i2[k0:]
            col3 = self.col3[k0 * Pp:]
            L.call("ss_roi_norm", R.data_ptr() + f0 * HW, n, HW, int(standardize), self.xn.data_ptr(), None, s)
            L.call("ss_im2col3x3", self.xn.data_ptr(), n, 1, H, W, col1.data_ptr(), 12, s)
            gemm(1, 1, n * HW, C1, 9, col1.data_ptr(), 12, w1, 9, self.y1.data_ptr(), C1, bias=b1, tag="gemm_cnn_generic")
            L.call("ss_relu_pool2", self.y1.data_ptr(), n, C1, H, W, a1.data_ptr(), i1.data_ptr(), s)
            L.call("ss_im2col3x3", a1.data_ptr(), n, C1, H // 2, W // 2, col2.data_ptr(), 9 * C1, s)
            gemm(1, 1, n * HW2, C2, 9 * C1, col2.data_ptr(), 9 * C1, w2, 9 * C1, self.y2.data_ptr(), C2, bias=b2, tag="gemm_cnn_generic")
            L.call("ss_relu_pool2", self.y2.data_ptr(), n, C2, H // 2, W // 2, a2.data_ptr(), i2.data_ptr(), s)
            L.call("ss_im2col3x3", a2.data_ptr(), n, C2, H // 4, W // 4, col3.data_ptr(), 9 * C2, s)
            gemm(1, 1, n * Pp, C3, 9 * C2, col3.data_ptr(), 9 * C2, w3, 9 * C2, self.y3.data_ptr(), C3, bias=b3, tag="gemm_cnn_generic")
            feat = self.feat[f0:]
            L.call("ss_relu_mean", self.y3.data_ptr(), n, Pp, C3, feat.data_ptr(), self.m3[f0 * Pp:].data_ptr() if stash else None, s)
            gemm(1, 1, n, E, C3, feat.data_ptr(), C3, wfc, C3, z_ptr + 4 * f0 * ld_z, ld_z, bias=bfc, tag="gemm_cnn_generic")

    # ------------------------------------------------------------------ backward (parameter gradients only: the image has none)
    def backward(self, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor], E: int, dz_ptr: int, ld_dz: int):
        from .engine import gemm

        s = L.stream()
        w1, b1, w2, b2, w3, b3, wfc, bfc = (P[k].data_ptr() for k in _NAMES)
        g1, gb1, g2, gb2, g3, gb3, gfc, gbfc = (G[k].data_ptr() for k in _NAMES)
        H, W, HW, HW2, Pp = self.H, self.W, self.HW, self.HW2, self.P

        def wgrad(M, Nn, rows, dy, ld_dy, col, ld_col, dst):  # dst[M][Nn] += dy^T col over `rows` pixel rows (K slices + atomics)
            splits = max(1, min(rows // 256, 512))
            gemm(0, 0, M, Nn, rows, dy, ld_dy, col, ld_col, dst, Nn, accumulate=True, atomic=True, splits=splits, tag="gemm_cnn_generic")

        for f0 in range(0, self.N, self.chunk):
            n = min(self.chunk, self.N - f0)
            dz = dz_ptr + 4 * f0 * ld_dz
            feat = self.feat[f0:]
            # fc: d feat = d z . W_fc ; d W_fc += d z^T feat ; d b_fc += column sums of d z
            gemm(1, 0, n, C3, E, dz, ld_dz, wfc, C3, self.dfeat.data_ptr(), C3, tag="gemm_cnn_generic")
            gemm(0, 0, E, C3, n, dz, ld_dz, feat.data_ptr(), C3, gfc, C3, accumulate=True, atomic=True, a_colsum=gbfc, tag="gemm_cnn_generic")
            # conv3
            dy3 = self.y3.data_ptr()
            L.call("ss_mask_scale", self.m3[f0 * Pp:].data_ptr(), self.dfeat.data_ptr(), n, Pp, C3, dy3, s)
            L.call("ss_colsum_f32", dy3, n * Pp, C3, C3, gb3, s)
            col3 = self.col3[f0 * Pp:].data_ptr()
            wgrad(C3, 9 * C2, n * Pp, dy3, C3, col3, 9 * C2, g3)
            gemm(1, 0, n * Pp, 9 * C2, C3, dy3, C3, w3, 9 * C2, self.dcol.data_ptr(), 9 * C2, tag="gemm_cnn_generic")
            L.call("ss_col2im3x3", self.dcol.data_ptr(), 9 * C2, n, C2, H // 4, W // 4, self.da2.data_ptr(), s)
            # conv2
            dy2 = self.y2.data_ptr()
            L.call("ss_pool2_bwd", self.da2.data_ptr(), self.a2[f0:].data_ptr(), self.i2[f0:].data_ptr(), n, C2, H // 4, W // 4, dy2, s)
            L.call("ss_colsum_f32", dy2, n * HW2, C2, C2, gb2, s)
            wgrad(C2, 9 * C1, n * HW2, dy2, C2, self.col2[f0 * HW2:].data_ptr(), 9 * C1, g2)
            gemm(1, 0, n * HW2, 9 * C1, C2, dy2, C2, w2, 9 * C1, self.dcol.data_ptr(), 9 * C1, tag="gemm_cnn_generic")
            L.call("ss_col2im3x3", self.dcol.data_ptr(), 9 * C1, n, C1, H // 2, W // 2, self.da1.data_ptr(), s)
            # conv1 (no data gradient: the uint8 image has none)
            dy1 = self.y1.data_ptr()
            L.call("ss_pool2_bwd", self.da1.data_ptr(), self.a1[f0:].data_ptr(), self.i1[f0:].data_ptr(), n, C1, H // 2, W // 2, dy1, s)
            L.call("ss_colsum_f32", dy1, n * HW, C1, C1, gb1, s)
            wgrad(C1, 9, n * HW, dy1, C1, self.col1[f0 * HW:].data_ptr(), 12, g1)
