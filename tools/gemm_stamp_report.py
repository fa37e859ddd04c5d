#!/usr/bin/env python3
"""Diagnostic: where a weight-gradient GEMM workgroup spends its life (needs the -DSS_STAMP build).

Thread 0 of workgroups 0..255 accumulates clock64() deltas: prologue (first fetch -> LDS -> barrier), per k tile:
fetch issue + LDS reads + MFMAs, LDS store of the next tile, barrier; epilogue (atomics, drained).

The timers sit in the REGISTER-STAGED kernel (gemm_f32_kernel); the LDS-DMA ring kernel that the step uses carries none, so
this tool forces the register-staged one (SS_GEMM_NO_DMA=1): it explains that kernel, not the production GEMM times."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SS_GEMM_NO_DMA"] = "1"
os.environ.setdefault("SS_HOTPATH_LIB", os.path.join(ROOT, "silent_speech_amd", "libss_hotpath_stamp.so"))
from silent_speech_amd import _lib as L  # noqa: E402
from silent_speech_amd import engine as E  # noqa: E402

NAMES = {15: "prologue", 1: "fetch issue + LDS reads + MFMAs", 2: "LDS store of next tile", 3: "barrier", 4: "epilogue (drained)"}


def report(fn, title):
    buf = np.zeros(512 * 24, np.uint64)
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(512, 24).astype(np.float64)[:256, :16] / 3.0  # the table sums the three launches since the last read
    tot = t.sum(1).mean()
    print(f"{title}: {tot / 2400:.2f} us per workgroup (first 256 workgroups)")
    for k, nm in NAMES.items():
        print(f"   {nm:34s} {t[:, k].mean() / 2400:7.2f} us  {100 * t[:, k].mean() / tot:5.1f} %")


def main():
    dev = torch.device("cuda")
    B, T, H = 256, 30, 192
    N = B * T
    dG = torch.randn(2, N, 4 * H, device=dev)
    out = torch.randn(N, 2 * H, device=dev)
    gWh = torch.zeros(2, 3 * H, H, device=dev)
    lib = L.load()
    fn = lib.ss_debug_stamps_gemm
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    for target in (256, 768):
        splits = E.split_k(2 * H, H, B * (T - 1), 2, target)
        maps = dict(a_map=(T - 1, T, 1), b_map=(T - 1, T, 0))
        st = (N * 4 * H - 4 * H, H + 2 * H, 3 * H * H, 0, 0)
        for _ in range(3):
            E.gemm(0, 0, 2 * H, H, B * (T - 1), dG.data_ptr(), 4 * H, out.data_ptr(), 2 * H, gWh.data_ptr(), H,
                   accumulate=True, atomic=True, splits=splits, batch=2, strides=st, **maps)
        ktiles = -(-(B * (T - 1)) // splits) // 16
        report(fn, f"dW_hh 384x192xK, batch 2, splits {splits} (~{ktiles} k tiles per workgroup)")
    # the layer-1 input projection: (N x 384) . (576 x 384)^T per direction, K contiguous on both sides
    x = torch.randn(N, 2 * H, device=dev)
    w = torch.randn(2, 3 * H, 2 * H, device=dev)
    b = torch.randn(2, 3 * H, device=dev)
    gi = torch.empty(2, N, 3 * H, device=dev)
    for _ in range(3):
        E.gemm(1, 1, N, 3 * H, 2 * H, x.data_ptr(), 2 * H, w.data_ptr(), 2 * H, gi.data_ptr(), 3 * H, bias=b.data_ptr(), batch=2,
               strides=(0, 3 * H * 2 * H, N * 3 * H, 3 * H, 0))
    report(fn, "ih l1 7680x576x384, batch 2 (24 k tiles per workgroup, 1080 workgroups)")
    # d layer_in of layer 1: (N x 576) . (576 x 384) per direction, atomically summed over the directions
    dx = torch.zeros(N, 2 * H, device=dev)
    for _ in range(3):
        E.gemm(1, 0, N, 2 * H, 3 * H, dG.data_ptr(), 4 * H, w.data_ptr(), 2 * H, dx.data_ptr(), 2 * H, accumulate=True, atomic=True,
               batch=2, strides=(N * 4 * H, 3 * H * 2 * H, 0, 0, 0))
    report(fn, "dX l1 7680x384x576, batch 2 (36 k tiles per workgroup, 720 workgroups, atomic)")
    x0 = torch.randn(N, 116, device=dev)
    w0 = torch.randn(2, 3 * H, 116, device=dev)
    for _ in range(3):
        E.gemm(1, 1, N, 3 * H, 116, x0.data_ptr(), 116, w0.data_ptr(), 116, gi.data_ptr(), 3 * H, bias=b.data_ptr(), batch=2,
               strides=(0, 3 * H * 116, N * 3 * H, 3 * H, 0))
    report(fn, "ih l0 7680x576x116, batch 2 (8 k tiles per workgroup, 1080 workgroups)")


if __name__ == "__main__":
    main()
