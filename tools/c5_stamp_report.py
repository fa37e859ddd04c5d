#!/usr/bin/env python3
"""Diagnostic: per-stage cycle shares of three config-5 CNN kernels (needs the -DSS_STAMP build: python -m silent_speech_amd.build
--stamp).  Thread 0 of every workgroup accumulates clock64() deltas between the stamps; cycles per frame = table / frames walked."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SS_HOTPATH_LIB", os.path.join(ROOT, "silent_speech_amd", "libss_hotpath_stamp.so"))
from silent_speech_amd import _lib as L  # noqa: E402

N = 7680
FPW = N // 256


def read(fn):
    buf = np.zeros(512 * 24, np.uint64)
    assert fn(buf.ctypes.data) == 0
    return buf.reshape(512, 24).astype(np.float64)[:256, :16]


def report(title, t, names, launches):
    t = t / launches / FPW
    tot = sum(t[:, k].mean() for k in names)
    print(f"{title}: {tot:.0f} cycles per frame (stamped stages)")
    for k, nm in names.items():
        print(f"   {nm:64s} {t[:, k].mean():8.0f}  {100 * t[:, k].mean() / tot:5.1f} %")


def main():
    lib = L.load()
    ffwd, fbwd = lib.ss_debug_stamps_c5_fwd, lib.ss_debug_stamps_c5_bwd
    for f in (ffwd, fbwd):
        f.argtypes, f.restype = [C.c_void_p], C.c_int
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    R = torch.randint(0, 256, (N, 96, 96), device=dev, dtype=torch.uint8, generator=g)
    w1, b1 = torch.randn(16, 1, 3, 3, device=dev) / 3, torch.randn(16, device=dev) * 0.1
    w2, b2 = torch.randn(32, 16, 3, 3, device=dev) / 12, torch.randn(32, device=dev) * 0.1
    a2 = torch.empty(N, 24, 24, 32, device=dev, dtype=torch.int16)
    i2 = torch.empty(N, 24, 24, 32, device=dev, dtype=torch.uint8)
    st = torch.empty(N, 2, device=dev)
    read(ffwd); read(fbwd)
    reps = 3
    for _ in range(reps):
        L.call("ss_c5_conv12_fwd", R.data_ptr(), N, 1, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), a2.data_ptr(), i2.data_ptr(),
               st.data_ptr(), L.stream())
    report("conv12_fwd", read(ffwd), {15: "frame top", 0: "statistics, table, bf16 image (3 barriers inside)", 1: "conv1 rows", 2: "barrier",
                                      3: "conv2 half: MFMAs + epilogue", 4: "barrier", 5: "copy-out of the half", 6: "barrier"}, reps)
    da2 = torch.randint(-300, 300, (N, 24, 24, 32), device=dev, dtype=torch.int16)
    gw2, gb2 = torch.zeros(32, 16, 3, 3, device=dev), torch.zeros(32, device=dev)
    for _ in range(reps):
        L.call("ss_c5_conv2_wgrad_rc", R.data_ptr(), st.data_ptr(), 1, w1.data_ptr(), b1.data_ptr(), da2.data_ptr(), i2.data_ptr(), N,
               gw2.data_ptr(), gb2.data_ptr(), L.stream())
    report("conv2_wgrad_rc", read(fbwd), {15: "frame top", 0: "frame image (RC)", 1: "dy expand commit", 2: "conv1 rows (RC) / a_in commit", 3: "barrier",
                                          4: "issue of the next band's loads", 5: "MFMAs", 6: "barrier"}, reps)
    gw1, gb1 = torch.zeros(16, 1, 3, 3, device=dev), torch.zeros(16, device=dev)
    for _ in range(reps):
        L.call("ss_c5_conv2_dgrad_conv1_wgrad", da2.data_ptr(), i2.data_ptr(), N, w2.data_ptr(), R.data_ptr(), st.data_ptr(), 1, w1.data_ptr(),
               b1.data_ptr(), None, gw1.data_ptr(), gb1.data_ptr(), L.stream())
    report("conv2_dgrad_conv1_wgrad", read(fbwd), {15: "frame top", 0: "frame images", 9: "wait for the band's loads", 1: "dy expand commit", 2: "barrier", 3: "issue + d a1 MFMAs + oa stores",
                                                   4: "conv1 rows (pool winners)", 5: "barrier", 6: "slot-masked images of a half band", 7: "barrier",
                                                   8: "conv1-wgrad MFMAs of a half band", 9: "barrier"}, reps)
    # ---- the last layer's weight gradient (12 x 12 x 64 -> 96): per-frame stages
    a3 = torch.randint(-300, 300, (N, 12, 12, 64), device=dev, dtype=torch.int16)
    dz = torch.randn(N, 64, device=dev)
    wfc = torch.randn(64, 96, device=dev) / 10
    m4 = torch.randint(0, 2, (N, 144, 96), device=dev, dtype=torch.uint8)
    feat = torch.rand(N, 96, device=dev)
    gw4, gb4 = torch.zeros(96, 64, 3, 3, device=dev), torch.zeros(96, device=dev)
    gwf, gbf = torch.zeros(64, 96, device=dev), torch.zeros(64, device=dev)
    read(fbwd)
    for _ in range(reps):
        L.call("ss_c5_conv_last_wgrad", a3.data_ptr(), dz.data_ptr(), 64, 64, wfc.data_ptr(), m4.data_ptr(), feat.data_ptr(), N, gw4.data_ptr(),
               gb4.data_ptr(), gwf.data_ptr(), gbf.data_ptr(), L.stream())
    report("conv_last_wgrad", read(fbwd), {15: "frame top", 0: "-", 1: "d feat, fc gradients, dy image from the sign mask", 2: "a_in commit", 3: "barrier",
                                           4: "issue of the next frame's loads", 5: "MFMAs", 6: "barrier"}, reps)
    dfe = dz @ wfc
    part4 = torch.empty(256 * 96 * 64 * 9, device=dev)
    for _ in range(reps):
        L.call("ss_c5_conv_last_wgrad_df", a3.data_ptr(), dfe.data_ptr(), m4.data_ptr(), N, gw4.data_ptr(), gb4.data_ptr(), part4.data_ptr(), part4.numel(),
               L.stream())
    tt = read(fbwd)
    print(f"   conv_last_wgrad_df per LAUNCH (cycles, mean over workgroups): before the frame loop {tt[:, 14].mean() / reps:.0f}, weight-gradient flush "
          f"{tt[:, 10].mean() / reps:.0f}, bias flush {tt[:, 11].mean() / reps:.0f}")
    report("conv_last_wgrad_df", tt, {15: "frame top", 0: "-", 1: "d feat, fc gradients, dy image from the sign mask", 2: "a_in commit", 3: "barrier",
                                           4: "issue of the next frame's loads", 5: "MFMAs", 6: "barrier"}, reps)
    # ---- generic forward / data-gradient kernels: layers 3 and 4
    fnames = {15: "frame top", 0: "commit of the frame image", 1: "barrier", 2: "issue of the next frame", 3: "units: MFMAs + epilogue", 4: "barrier",
              5: "copy-out / average"}
    dnames = {15: "band top", 0: "dy image (expand / mask)", 1: "barrier", 2: "issue of the next band", 3: "units: MFMAs + stores", 4: "barrier",
              5: "copy-out"}
    a2 = torch.randint(-300, 300, (N, 24, 24, 32), device=dev, dtype=torch.int16)
    w3, b3 = torch.randn(64, 32, 3, 3, device=dev) / 17, torch.randn(64, device=dev) * 0.1
    w4, b4 = torch.randn(96, 64, 3, 3, device=dev) / 24, torch.randn(96, device=dev) * 0.1
    a3o = torch.empty(N, 12, 12, 64, device=dev, dtype=torch.int16)
    i3 = torch.empty(N, 12, 12, 64, device=dev, dtype=torch.uint8)
    read(ffwd)
    for _ in range(reps):
        L.call("ss_c5_conv_fwd", 3, a2.data_ptr(), N, w3.data_ptr(), b3.data_ptr(), a3o.data_ptr(), i3.data_ptr(), L.stream())
    report("conv3_fwd (2 workgroups per CU: x2)", read(ffwd), fnames, reps)
    for _ in range(reps):
        L.call("ss_c5_conv_last_fwd_feat", a3.data_ptr(), N, w4.data_ptr(), b4.data_ptr(), m4.data_ptr(), feat.data_ptr(), L.stream())
    report("conv_last_fwd_feat (weight-stationary: consumer wave 0)", read(ffwd), {15: "pass top", 0: "multiply + epilogue", 1: "barrier (wait for the producers)"}, reps)
    da3 = torch.randint(-300, 300, (N, 12, 12, 64), device=dev, dtype=torch.int16)
    da2o = torch.empty(N, 24, 24, 32, device=dev, dtype=torch.int16)
    read(fbwd)
    for _ in range(reps):
        L.call("ss_c5_conv_dgrad", 3, da3.data_ptr(), i3.data_ptr(), N, w3.data_ptr(), da2o.data_ptr(), L.stream())
    report("conv3_dgrad (2 workgroups per CU: x2)", read(fbwd), dnames, reps)
    for _ in range(reps):
        L.call("ss_c5_conv_last_dgrad_df", dfe.data_ptr(), m4.data_ptr(), N, w4.data_ptr(), a3o.data_ptr(), L.stream())
    report("conv_last_dgrad_df", read(fbwd), dnames, reps)
    gw3, gb3 = torch.zeros(64, 32, 3, 3, device=dev), torch.zeros(64, device=dev)
    for _ in range(reps):
        L.call("ss_c5_conv_wgrad", 3, a2.data_ptr(), da3.data_ptr(), i3.data_ptr(), N, gw3.data_ptr(), gb3.data_ptr(), L.stream())
    report("conv3_wgrad", read(fbwd), {15: "frame top", 0: "-", 1: "dy expand commit", 2: "a_in commit", 3: "barrier", 4: "issue of the next band's loads",
                                       5: "MFMAs", 6: "barrier"}, reps)


if __name__ == "__main__":
    main()
