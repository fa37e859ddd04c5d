#!/usr/bin/env python3
"""Diagnostic: does a hipGraph replay shorten the per-step gap of the bf16 recurrence (one launch per time step)?
Times ss_gru_bf16_fwd (B = 256, T = 30, H = 512: 30 dependent launches) eagerly and as a captured graph."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import _lib as L  # noqa: E402


def main():
    B, T, H = 256, 30, 512
    N = B * T
    dev = "cuda"
    torch.manual_seed(0)
    w = torch.randn(2, 3 * H, H, device=dev) * 0.05
    wb = torch.empty(2, 3 * H, H, device=dev, dtype=torch.int16)
    wtb = torch.empty(2, H, 3 * H, device=dev, dtype=torch.int16)
    L.call("ss_gru_bf16_prep", w[0].data_ptr(), w[1].data_ptr(), H, wb.data_ptr(), wtb.data_ptr(), L.stream())
    nb = ctypes.c_long(0)
    assert L.load().ss_gru_bf16_ws_bytes(B, H, ctypes.byref(nb)) == 0
    ws = torch.empty(nb.value, device=dev, dtype=torch.uint8)
    gi = torch.randn(2, N, 3 * H, device=dev)
    bf_, br_ = torch.zeros(3 * H, device=dev), torch.zeros(3 * H, device=dev)
    lens = torch.full((B,), T, device=dev, dtype=torch.int32)
    out = torch.empty(N, 2 * H, device=dev)
    save = torch.empty(2, N, 4, H, device=dev)

    def run():
        L.call("ss_gru_bf16_fwd", gi.data_ptr(), wb.data_ptr(), bf_.data_ptr(), br_.data_ptr(), lens.data_ptr(), B, T, H,
               out.data_ptr(), save.data_ptr(), None, None, 0.0, 0, 0, ws.data_ptr(), None, 0, L.stream())

    def timed(fn, reps=20):
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

    t_eager = timed(run)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            run()
    torch.cuda.synchronize()
    t_graph = timed(g.replay)
    print(f"ss_gru_bf16_fwd, 30 steps: eager {t_eager:.1f} us ({t_eager / T:.2f} per step), graph replay {t_graph:.1f} us ({t_graph / T:.2f} per step)")


if __name__ == "__main__":
    main()
