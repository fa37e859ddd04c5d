#!/usr/bin/env python3
"""Diagnostic: HIP-event launch times of roi_cnn_fwd / roi_cnn_bwd inside the config-2 step, for the library named by
SS_HOTPATH_LIB (default: the in-tree build).  Run it once per variant inside ONE gpurun call: launch times move by a few
per cent between boxes, not between processes on one box.

    SS_HOTPATH_LIB=silent_speech_amd/libA.so python tools/cnn_ab.py && SS_HOTPATH_LIB=.../libB.so python tools/cnn_ab.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import silent_speech_amd as ss  # noqa: E402
from silent_speech_amd import _lib as L  # noqa: E402
from silent_speech_amd import engine as E  # noqa: E402


def main():
    B, T = 256, 30
    steps = int(os.environ.get("AB_STEPS", "20"))
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = ss.BiGRUClassifier(84, 5, use_roi=True).to(dev).train()
    tr = ss.Trainer(m)
    X = torch.randn(B, T, 84, device=dev)
    hh, ww = (int(v) for v in os.environ.get("AB_ROI", "64,64").split(","))  # 64,64 (BASELINE.json) | 48,96 (the reference's ROI) | 32,32
    R = torch.randint(0, 256, (B, T, hh, ww), device=dev, dtype=torch.uint8)
    lengths = torch.full((B,), T, device=dev)
    y = torch.randint(0, 5, (B,), device=dev)
    for _ in range(5):
        tr.step(X, lengths, R, y)
    torch.cuda.synchronize()
    E.USE_SIDE_STREAM = False
    L.PROFILE = {}
    for _ in range(steps):
        tr.step(X, lengths, R, y)
    torch.cuda.synchronize()
    prof, L.PROFILE = L.PROFILE, None
    out = []
    for k in ("ss_roi_cnn_fwd_stash", "ss_roi_cnn_bwd"):
        ts = sorted(a.elapsed_time(b) for a, b in prof[k])
        out.append(f"{k} median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f}")
    print(os.path.basename(os.environ.get("SS_HOTPATH_LIB", "libss_hotpath.so")), " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
