#!/usr/bin/env python3
"""Soak of the fused train step under changing clip lengths: every step draws new lengths on the device (full clips, ragged, all
length 1, one long clip among short ones), so the frame list, the listed / unlisted kernel choice of the backward pass (a guess from
the last finished batch), the shared side stream and the multi-CU recurrences are exercised in every order for tens of thousands
of steps.  Every CHECK steps: loss and gradient norm finite, ``check_health()`` (lost-partner channel of the recurrences), and the
logits of that batch on a model that never uses the list -- bit-equal.

    python tools/train_soak.py [steps] [roi_h,roi_w]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import silent_speech_amd as ss  # noqa: E402
from silent_speech_amd import engine as E  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    hw = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (64, 64)
    CHECK = 2000
    dev = torch.device("cuda")
    B, T, D, C = 96, 24, 84, 5
    g = torch.Generator(device=dev).manual_seed(3)
    torch.manual_seed(0)
    m = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).train()
    tr = ss.Trainer(m, dropout=False)
    X = torch.randn(B, T, D, device=dev, generator=g)
    R = torch.randint(0, 256, (B, T) + hw, device=dev, dtype=torch.uint8, generator=g)
    y = torch.randint(0, C, (B,), device=dev, generator=g)
    E.SKIP_PADDED_FRAMES = False  # (read when a workspace is built: m0's inference workspace is built here, without a list)
    m0 = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).eval()
    with torch.no_grad():
        m0(X, torch.full((B,), T, device=dev, dtype=torch.int64), R)
    E.SKIP_PADDED_FRAMES = True
    t0 = time.time()
    kinds = [0, 0, 0]
    for k in range(1, steps + 1):
        r = k % 7
        if r in (0, 1):
            lengths = torch.full((B,), T, device=dev, dtype=torch.int64)
        elif r == 2:
            lengths = torch.ones(B, device=dev, dtype=torch.int64)
            lengths[k % B] = T
        else:
            lengths = torch.randint(1, T + 1, (B,), device=dev, generator=g, dtype=torch.int64)
        loss, _ = tr.step(X, lengths, R, y)
        ws = m._workspace(X, R, train=True, slot=0)
        kinds[0 if ws.walk_listed else 1] += 1
        if k % CHECK == 0:
            torch.cuda.synchronize()
            gn = float(tr.grad_norm())
            assert torch.isfinite(loss) and gn == gn and gn < 1e6, (k, float(loss), gn)
            m.check_health()
            # the same parameters on a model that walks every frame: bit-equal logits
            m0.load_state_dict(m.state_dict())
            m.eval()
            with torch.no_grad():
                la, lb = m(X, lengths, R), m0(X, lengths, R)
            m.train()
            assert torch.equal(la, lb), (k, float((la - lb).abs().max()))
            print(f"step {k}: loss {float(loss):.5f} grad norm {gn:.4f}; listed {kinds[0]} / unlisted {kinds[1]} steps; "
                  f"{(time.time() - t0) / k * 1e3:.3f} ms per step", flush=True)
    torch.cuda.synchronize()
    print("soak ok:", steps, "steps")


if __name__ == "__main__":
    main()
