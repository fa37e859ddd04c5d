#!/usr/bin/env python3
"""Diagnostic: launch time of every persistent config-5 CNN kernel at 1, 2, 4 and 30 frames per workgroup: the intercept is what a
launch costs before / after its frame walk (weight staging, gradient flush), the slope the time per frame.
python tools/c5_fixed_cost.py   (on the GPU box)"""
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
    NM = 7680
    i16 = lambda *s: torch.randint(-300, 300, s, device=dev, dtype=torch.int16)  # noqa: E731
    u8 = lambda hi, *s: torch.randint(0, hi, s, device=dev, dtype=torch.uint8)   # noqa: E731
    R = u8(256, NM, 96, 96)
    w1, b1 = torch.randn(16, 1, 3, 3, device=dev) / 3, torch.randn(16, device=dev) * 0.1
    w2, b2 = torch.randn(32, 16, 3, 3, device=dev) / 12, torch.randn(32, device=dev) * 0.1
    w3, b3 = torch.randn(64, 32, 3, 3, device=dev) / 17, torch.randn(64, device=dev) * 0.1
    a2, i2, st, i1 = i16(NM, 24, 24, 32), u8(5, NM, 24, 24, 32), torch.rand(NM, 2, device=dev) + 0.5, u8(5, NM, 48, 48, 16)
    a3, i3 = i16(NM, 12, 12, 64), u8(5, NM, 12, 12, 64)
    da2, da3 = i16(NM, 24, 24, 32), i16(NM, 12, 12, 64)
    gw1, gb1 = torch.zeros(16, 1, 3, 3, device=dev), torch.zeros(16, device=dev)
    gw2, gb2 = torch.zeros(32, 16, 3, 3, device=dev), torch.zeros(32, device=dev)
    gw3, gb3 = torch.zeros(64, 32, 3, 3, device=dev), torch.zeros(64, device=dev)
    part = torch.empty(256 * 96 * 64 * 9, device=dev)
    pn = 0 if os.environ.get("SS_NO_PART") else part.numel()  # SS_NO_PART=1: float atomics straight onto the gradients
    s = L.stream()
    P = lambda t: t.data_ptr()  # noqa: E731
    kernels = {
        "conv12_fwd (+i1)": lambda N: L.call("ss_c5_conv12_fwd_i1", P(R), N, 1, P(w1), P(b1), P(w2), P(b2), P(a2), P(i2), P(st), P(i1), s),
        "conv3_fwd": lambda N: L.call("ss_c5_conv_fwd", 3, P(a2), N, P(w3), P(b3), P(a3), P(i3), s),
        "conv3_wgrad": lambda N: L.call("ss_c5_conv_wgrad_ws", 3, P(a2), P(da3), P(i3), N, P(gw3), P(gb3), P(part), pn, s),
        "conv3_dgrad": lambda N: L.call("ss_c5_conv_dgrad", 3, P(da3), P(i3), N, P(w3), P(da2), s),
        "conv2_wgrad_rc": lambda N: L.call("ss_c5_conv2_wgrad_rc_ws", P(R), P(st), 1, P(w1), P(b1), P(da2), P(i2), N, P(gw2), P(gb2), P(part),
                                           pn, s),
        "conv2_dgrad_conv1_wgrad": lambda N: L.call("ss_c5_conv2_dgrad_conv1_wgrad_i1", P(da2), P(i2), N, P(w2), P(R), P(st), 1, P(w1), P(b1),
                                                    None, P(gw1), P(gb1), P(i1), s),
    }
    for name, fn in kernels.items():
        t = {N: timeit(lambda: fn(N)) for N in (256, 512, 1024, 7680)}
        slope = (t[7680] - t[1024]) / 26
        print(f"{name:26s} 1 / 2 / 4 / 30 frames per workgroup: {t[256]:7.1f} {t[512]:7.1f} {t[1024]:7.1f} {t[7680]:7.1f} us   "
              f"per frame {slope:5.2f} us, intercept {t[7680] - 30 * slope:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
