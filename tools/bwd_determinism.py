#!/usr/bin/env python3
"""Diagnostic: run-to-run spread of the ROI-CNN parameter gradients on fixed inputs (float atomics reorder sums: ~1e-6
relative is the floor; anything larger is a race).  SS_HOTPATH_LIB selects the library; DET_B sets the batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import silent_speech_amd as ss  # noqa: E402


def main():
    B, T = int(os.environ.get("DET_B", "3")), 30
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = ss.BiGRUClassifier(84, 5, use_roi=True).to(dev).train()
    X = torch.randn(B, T, 84, device=dev)
    hh, ww = (int(v) for v in os.environ.get("AB_ROI", "64,64").split(","))  # 48,96 = the reference's ROI
    R = torch.randint(0, 256, (B, T, hh, ww), device=dev, dtype=torch.uint8)
    lengths = torch.full((B,), T, device=dev)
    y = torch.randint(0, 5, (B,), device=dev)
    names = [n for n, _ in m.named_parameters() if n.startswith("roi_cnn")]
    ref = None
    worst = {n: 0.0 for n in names}
    for it in range(int(os.environ.get("DET_RUNS", "40"))):
        m.zero_grad(set_to_none=False)
        m._step_seed = 0  # the same dropout masks every run
        logits = m(X, lengths, R)
        loss = torch.nn.functional.cross_entropy(logits, y)
        loss.backward()
        g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if n in worst}
        if ref is None:
            ref = g
        else:
            for n in names:
                d = float((g[n] - ref[n]).abs().max() / (ref[n].abs().max() + 1e-30))
                worst[n] = max(worst[n], d)
    print(os.path.basename(os.environ.get("SS_HOTPATH_LIB", "in-tree")), f"B={B}", {n.replace('roi_cnn.', ''): f"{v:.1e}" for n, v in worst.items()}, flush=True)


if __name__ == "__main__":
    main()
