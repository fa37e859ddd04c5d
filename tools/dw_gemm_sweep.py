#!/usr/bin/env python3
"""Diagnostic: time the split-K weight-gradient GEMMs of one GRU layer for several workgroup targets."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import engine as E  # noqa: E402


def main():
    dev = torch.device("cuda")
    B, T, H = 256, 30, 192
    N = B * T
    dG = torch.randn(2, N, 4 * H, device=dev)
    out = torch.randn(N, 2 * H, device=dev)
    for K in (116, 384):
        lin = torch.randn(N, K, device=dev)
        gWi = torch.zeros(2, 3 * H, K, device=dev)
        gWh = torch.zeros(2, 3 * H, H, device=dev)
        for target in (64, 128, 192, 256, 384, 512, 768):
            def run():
                E.gemm(0, 0, 3 * H, K, N, dG.data_ptr(), 4 * H, lin.data_ptr(), K, gWi.data_ptr(), K, accumulate=True,
                       atomic=True, splits=E.split_k(3 * H, K, N, 2, target), batch=2, strides=(N * 4 * H, 0, 3 * H * K, 0, 0))
                maps = dict(a_map=(T - 1, T, 1), b_map=(T - 1, T, 0))
                st = (N * 4 * H - 4 * H, H + 2 * H, 3 * H * H, 0, 0)
                E.gemm(0, 0, 2 * H, H, B * (T - 1), dG.data_ptr(), 4 * H, out.data_ptr(), 2 * H, gWh.data_ptr(), H,
                       accumulate=True, atomic=True, splits=E.split_k(2 * H, H, B * (T - 1), 2, target), batch=2, strides=st, **maps)
                E.gemm(0, 0, H, H, B * (T - 1), dG.data_ptr() + 3 * H * 4, 4 * H, out.data_ptr(), 2 * H,
                       gWh.data_ptr() + 2 * H * H * 4, H, accumulate=True, atomic=True,
                       splits=E.split_k(H, H, B * (T - 1), 2, target), batch=2, strides=st, **maps)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"K={K:4d} target_wgs={target:4d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us for the three dW GEMMs", flush=True)


if __name__ == "__main__":
    main()
