#!/usr/bin/env python3
"""Diagnostic: per-tensor relative L2 error of the bf16 path's GRU weight gradients against the bf16 restatement and the f32
oracle, for 2 and 3 GRU layers (landmark-only model, H = 512, B = 64 so that the grouped weight-gradient launch is taken)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights as W  # noqa: E402
from oracle import model_ref as MR  # noqa: E402
from oracle import model_ref_bf16 as MB  # noqa: E402
import silent_speech_amd as ss  # noqa: E402


def main():
    for layers, T in ((2, 3), (3, 3), (3, 6)):
        B, H = 64, 512
        sd = W.make_state_dict(8, 84, 7, False, hidden=H, gru_layers=layers)
        X, Lh, R, y = W.make_inputs(8, B, T, 84, 7, None, lengths=[T] * B)
        m = ss.BiGRUClassifier(84, 7, use_roi=False, hidden=H, gru_layers=layers, precision="bf16")
        m.load_state_dict(sd)
        m.cuda().train()
        tr = ss.Trainer(m, dropout=False)
        loss, _ = tr.step(X.cuda(), Lh.cuda(), None, y.cuda())
        torch.cuda.synchronize()
        l_emu, _, g_emu = MB.loss_and_grads(sd, X, Lh, None, y)
        l_f32, _, g_f32 = MR.loss_and_grads(sd, X, Lh, None, y)
        G = m._views_of(m.flat_grads)
        print(f"layers {layers} T {T}: loss {float(loss):.6f} emu {float(l_emu):.6f} f32 {float(l_f32):.6f}")
        for k in g_emu:
            if not k.startswith("gru."):
                continue
            got = G[k].detach().cpu()
            e1 = float((got - g_emu[k]).norm() / g_emu[k].norm())
            e2 = float((got - g_f32[k]).norm() / g_f32[k].norm())
            e3 = float((g_emu[k] - g_f32[k]).norm() / g_f32[k].norm())
            print(f"   {k:32s} vs emu {e1:.2e}  vs f32 {e2:.2e}  emu vs f32 {e3:.2e}  |g| {float(g_f32[k].norm()):.3e}")
        m.check_health()


if __name__ == "__main__":
    main()
