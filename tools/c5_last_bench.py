#!/usr/bin/env python3
"""Diagnostic: launch time of the last conv layer's kernels against the number of frames (fixed cost per launch = weight staging /
gradient flush; slope = cycles per frame).  python tools/c5_last_bench.py   (SS_C5_LAST_WS=0: LDS-resident weights)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import _lib as L  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = "cuda"
    NMAX = 7680
    a3 = torch.randint(-300, 300, (NMAX, 12, 12, 64), device=dev, dtype=torch.int16)
    w4, b4 = torch.randn(96, 64, 3, 3, device=dev) / 24, torch.randn(96, device=dev) * 0.1
    m4 = torch.randint(0, 2, (NMAX, 144, 96), device=dev, dtype=torch.uint8)
    feat = torch.rand(NMAX, 96, device=dev)
    dfe = torch.randn(NMAX, 96, device=dev)
    da3 = torch.empty(NMAX, 12, 12, 64, device=dev, dtype=torch.int16)
    gw4, gb4 = torch.zeros(96, 64, 3, 3, device=dev), torch.zeros(96, device=dev)
    part = torch.empty(256 * 96 * 64 * 9, device=dev)
    s = L.stream()
    for N in (256, 512, 1024, 2560, 7680):
        tf = timeit(lambda: L.call("ss_c5_conv_last_fwd_feat", a3.data_ptr(), N, w4.data_ptr(), b4.data_ptr(), m4.data_ptr(), feat.data_ptr(), s))
        td = timeit(lambda: L.call("ss_c5_conv_last_dgrad_df", dfe.data_ptr(), m4.data_ptr(), N, w4.data_ptr(), da3.data_ptr(), s))
        tw = timeit(lambda: L.call("ss_c5_conv_last_wgrad_df", a3.data_ptr(), dfe.data_ptr(), m4.data_ptr(), N, gw4.data_ptr(), gb4.data_ptr(), part.data_ptr(), 0 if os.environ.get("SS_NO_PART") else part.numel(), s))
        print(f"N={N:5d} ({N // 256:2d} frames per workgroup): fwd {tf:7.1f} us   dgrad {td:7.1f} us   wgrad {tw:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
