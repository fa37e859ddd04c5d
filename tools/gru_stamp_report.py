#!/usr/bin/env python3
"""Diagnostic: per-stage time of the multi-CU GRU kernels (needs the -DSS_STAMP build, see stamp_report.py).

Thread 0 (owner wave of the first unit tile) of every workgroup accumulates clock64() deltas (shader clock, ~2.4 GHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SS_HOTPATH_LIB", os.path.join(ROOT, "silent_speech_amd", "libss_hotpath_stamp.so"))
from silent_speech_amd import _lib as L  # noqa: E402

FWD = {15: "loop top", 0: "sweep h_prev granules", 1: "-> LDS panel + barrier", 2: "MFMA", 3: "k-split reduce",
       4: "gates + publish", }
BWD = {15: "loop top", 4: "gate grads -> LDS panel + barrier", 2: "MFMA + publish partials", 0: "d_g stores, prefetch, sweep partials",
       1: "sum -> LDS + barrier"}


def main():
    dev = torch.device("cuda")
    H, B, T = 192, 256, 30
    N = B * T
    gi = torch.randn(2, N, 3 * H, device=dev) * 0.5
    w = [torch.randn(3 * H, H, device=dev) * 0.07 for _ in range(2)]
    b = [torch.randn(3 * H, device=dev) * 0.07 for _ in range(2)]
    lens = torch.full((B,), T, device=dev, dtype=torch.int32)
    out = torch.empty(N, 2 * H, device=dev)
    save = torch.empty(2, N, 4, H, device=dev)
    dout = torch.randn(N, 2 * H, device=dev)
    dg = torch.empty(2, N, 4, H, device=dev)
    sync_ws = torch.zeros(L.gru_sync_bytes(B, T, H) // 4, device=dev, dtype=torch.int32)
    s = L.stream()
    lib = L.load()
    fn = lib.ss_debug_stamps_gru
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int

    def report(name, names):
        buf = np.zeros(512 * 24, np.uint64)
        assert fn(buf.ctypes.data) == 0
        t = buf.reshape(512, 24).astype(np.float64)[:192, :16] / 3.0  # the table sums the three launches since the last read
        tot = t.sum(1).mean()
        print(f"{name}: {tot / T / 2400:.2f} us per step (mean over workgroups; clock64 taken as 2.4 GHz)")
        for k, nm in names.items():
            print(f"   {nm:36s} {t[:, k].mean() / T / 2400:6.2f} us/step  {100 * t[:, k].mean() / tot:5.1f} %")

    for _ in range(3):
        L.call("ss_gru_fwd", gi.data_ptr(), w[0].data_ptr(), w[1].data_ptr(), b[0].data_ptr(), b[1].data_ptr(),
               lens.data_ptr(), B, T, H, out.data_ptr(), save.data_ptr(), sync_ws.data_ptr(), s)
    report("gru_split_fwd", FWD)
    for _ in range(3):
        L.call("ss_gru_bwd", dout.data_ptr(), out.data_ptr(), save.data_ptr(), w[0].data_ptr(), w[1].data_ptr(),
               lens.data_ptr(), B, T, H, dg.data_ptr(), 0.0, 0, 0, None, None, None, None, sync_ws.data_ptr(), s)
    report("gru_split_bwd", BWD)


if __name__ == "__main__":
    main()
