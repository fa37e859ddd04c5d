#!/usr/bin/env python3
"""Diagnostic: launch time of the persistent bf16 recurrence kernels against T (B = 256, H = 512): the intercept is what a launch
costs outside its time steps (weights into registers, XCD discovery, bias-gradient flush).  python tools/gru_bf16_fixed_cost.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import _lib as L  # noqa: E402

B, H = 256, 512


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
    lib = L.load()
    dev = "cuda"
    res = {}
    for T in (1, 2, 4, 30):
        N = B * T
        gi = torch.randn(2, N, 3 * H, device=dev) * 0.5
        whh = torch.randint(-100, 100, (2, 3 * H, H), device=dev, dtype=torch.int16)
        whht = torch.randint(-100, 100, (2, H, 3 * H), device=dev, dtype=torch.int16)
        bhh = torch.zeros(2, 3 * H, device=dev)
        lens = torch.full((B,), T, device=dev, dtype=torch.int32)
        out, save = torch.empty(N, 2 * H, device=dev), torch.empty(2, N, 4, H, device=dev)
        out_bf, out_dr = torch.empty(N, 2 * H, device=dev, dtype=torch.int16), torch.empty(N, 2 * H, device=dev, dtype=torch.int16)
        nb = C.c_long(0)
        lib.ss_gru_bf16_ws_bytes(B, H, C.byref(nb))
        ws = torch.empty(nb.value, device=dev, dtype=torch.uint8)
        lib.ss_gru_bf16_sync_bytes(B, T, H, C.byref(nb))
        sync = torch.zeros(nb.value // 4, device=dev, dtype=torch.int32)
        d_out = torch.randn(N, 2 * H, device=dev)
        dG_bf = torch.empty(2, N, 4, H, device=dev, dtype=torch.int16)
        gb = [torch.zeros(3 * H, device=dev) for _ in range(4)]
        s = L.stream()
        tf = timeit(lambda: L.call("ss_gru_bf16_fwd", gi.data_ptr(), whh.data_ptr(), bhh[0].data_ptr(), bhh[1].data_ptr(), lens.data_ptr(), B, T,
                                   H, out.data_ptr(), save.data_ptr(), out_bf.data_ptr(), out_dr.data_ptr(), 0.1, 1, 1 << 40, ws.data_ptr(),
                                   sync.data_ptr(), L.nbytes(sync), s))
        tb = timeit(lambda: L.call("ss_gru_bf16_bwd", d_out.data_ptr(), out.data_ptr(), save.data_ptr(), whht.data_ptr(), lens.data_ptr(), B, T,
                                   H, None, dG_bf.data_ptr(), 0.1, 1, 1 << 40, *[t_.data_ptr() for t_ in gb], ws.data_ptr(), sync.data_ptr(), L.nbytes(sync), s))
        res[T] = (tf, tb)
        print(f"T={T:3d}: forward {tf:7.1f} us   backward {tb:7.1f} us", flush=True)
    for k, nm in ((0, "forward"), (1, "backward")):
        slope = (res[30][k] - res[4][k]) / 26
        print(f"{nm}: {slope:.2f} us per step, intercept {res[30][k] - 30 * slope:.1f} us")


if __name__ == "__main__":
    main()
