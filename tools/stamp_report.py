#!/usr/bin/env python3
"""Diagnostic: per-stage cycle shares of the persistent ROI-CNN kernels (needs the -DSS_STAMP build).

    python -m silent_speech_amd.build --stamp
    SS_HOTPATH_LIB=silent_speech_amd/libss_hotpath_stamp.so python tools/stamp_report.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SS_HOTPATH_LIB", os.path.join(ROOT, "silent_speech_amd", "libss_hotpath_stamp.so"))
import silent_speech_amd as ss  # noqa: E402
from silent_speech_amd import _lib as L  # noqa: E402

FWD = {15: "loop top", 0: "stats+normalise", 1: "conv1 (MFMA)", 2: "a1 stash copy + conv2 (MFMA)", 3: "a2 stash copy",
       4: "conv3 (MFMA)", 5: "feat + fc"}
BWD = {15: "loop top (barrier E skew)", 0: "frame top: pixels / argmaxes out of staging", 3: "S1 dW3 (+ pooled-1 DMA issue)", 4: "S2 da2",
       10: "T: dy2 scatter from registers", 5: "T: a1 DMA wait, barrier", 6: "S3 dW2 (+ normalised frame, i1, d feat per wave)", 9: "S4 da1",
       7: "S5 dW1 + prefetch + the next frame's front"}


def main():
    B, T = 256, 30
    dev = torch.device("cuda")
    m = ss.BiGRUClassifier(84, 5, use_roi=True).to(dev).train()
    tr = ss.Trainer(m)
    X = torch.randn(B, T, 84, device=dev)
    hh, ww = (int(v) for v in os.environ.get("AB_ROI", "64,64").split(","))  # 48,96 = the reference's ROI
    R = torch.randint(0, 256, (B, T, hh, ww), device=dev, dtype=torch.uint8)
    lengths = torch.full((B,), T, device=dev)
    y = torch.randint(0, 5, (B,), device=dev)
    lib = L.load()
    if os.environ.get("SS_STAMP_MAX_WGS"):  # e.g. 128: one forward workgroup per CU (the cap counts CUs) -- a workgroup's solo time
        L.call("ss_roi_cnn_set_max_workgroups", int(os.environ["SS_STAMP_MAX_WGS"]))
    for k in range(3):
        if k == 2:  # the tables accumulate over launches and a read clears them: measure the last step only
            for nm in ("ss_debug_stamps_fwd", "ss_debug_stamps_bwd"):
                fn = getattr(lib, nm)
                fn.argtypes, fn.restype = [C.c_void_p], C.c_int
                scratch = np.zeros(512 * 24, np.uint64)
                assert fn(scratch.ctypes.data) == 0
        tr.step(X, lengths, R, y)
    for which, names in ((0, FWD), (1, BWD)):
        nwg = 512
        buf = np.zeros(nwg * 24, np.uint64)
        fn = getattr(lib, "ss_debug_stamps_fwd" if which == 0 else "ss_debug_stamps_bwd")
        fn.argtypes, fn.restype = [C.c_void_p], C.c_int
        assert fn(buf.ctypes.data) == 0
        t = buf.reshape(nwg, 24).astype(np.float64)
        used = 512 if which == 0 else 256   # forward: 512 workgroups of 256 threads (two per CU), backward: 256 of 512
        if os.environ.get("SS_STAMP_MAX_WGS"):
            used = int(os.environ["SS_STAMP_MAX_WGS"]) * (2 if which == 0 else 1)
        t = t[:used]
        stage = t[:, list(names)]
        tot = stage.sum(1).mean()
        fpw = B * T // used
        print(f"kernel {'roi_cnn_fwd' if which == 0 else 'roi_cnn_bwd'}: {tot / fpw:.0f} cycles per frame (mean over {used} workgroups, {fpw} frames each)")
        for k, name in names.items():
            print(f"   {name:28s} {t[:, k].mean() / fpw:9.0f} cyc/frame  {100 * t[:, k].mean() / tot:5.1f} %")
        loop = stage.sum(1)
        print(f"   frame loop per workgroup: min {loop.min():.0f}  mean {loop.mean():.0f}  max {loop.max():.0f} cycles; before the loop: mean {t[:, 14].mean():.0f} max {t[:, 14].max():.0f}")
        w0, w1 = t[:, 12], t[:, 13]
        print(f"   wall clock (100 MHz ticks): first entry -> last entry {w0.max() - w0.min():.0f}, first entry -> first exit {w1.min() - w0.min():.0f}, "
              f"-> last exit {w1.max() - w0.min():.0f}; lifetime mean {np.mean(w1 - w0):.0f}")


if __name__ == "__main__":
    main()
