#!/usr/bin/env python3
"""Diagnostic: the grouped weight-gradient launch of one GRU layer (ss_gemm_f32_splitk_group) alone on the chip, config-2 shapes.
    SS_GEMM_DW_WIDE=0 python tools/dw_group_bench.py ; SS_GEMM_DW_WIDE=1 python tools/dw_group_bench.py"""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import _lib as L  # noqa: E402
from silent_speech_amd import engine as E  # noqa: E402


WIDE = int(os.environ.get("SS_GEMM_DW_WIDE", "1"))


def main():
    dev = torch.device("cuda")
    B, T, H = 256, 30, 192
    N = B * T
    cfg = SimpleNamespace(hidden=H, in_dim=116)
    for l, K in ((1, 2 * H), (0, 116)):
        ws = SimpleNamespace(B=B, T=T, dG=[torch.randn(2, N, 4 * H, device=dev)] * 2, out=[torch.randn(N, 2 * H, device=dev)] * 2)
        lin = torch.randn(N, K, device=dev)
        G = {}
        flat = torch.zeros(2 * 3 * H * K + 2 * 3 * H * H, device=dev)
        G[f"gru.weight_ih_l{l}"] = flat[:3 * H * K].view(3 * H, K)
        G[f"gru.weight_ih_l{l}_reverse"] = flat[3 * H * K:2 * 3 * H * K].view(3 * H, K)
        o = 2 * 3 * H * K
        G[f"gru.weight_hh_l{l}"] = flat[o:o + 3 * H * H].view(3 * H, H)
        G[f"gru.weight_hh_l{l}_reverse"] = flat[o + 3 * H * H:].view(3 * H, H)
        pr = E.dw_problems(ws, G, cfg, l, lin.data_ptr(), K)
        wsf = torch.empty(L.gemm_group_ws_floats(pr), device=dev)
        arr, n = L.gemm_group(pr)
        for _ in range(3):
            L.call("ss_gemm_f32_splitk_group", arr, n, wsf.data_ptr(), wsf.numel(), WIDE, L.stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.call("ss_gemm_f32_splitk_group", arr, n, wsf.data_ptr(), wsf.numel(), WIDE, L.stream())
        e1.record()
        torch.cuda.synchronize()
        gf = sum(2.0 * q.M * q.N * q.K * q.batch for q in pr) / 1e9
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"layer {l}: {us:7.1f} us per grouped launch (+ reduce), {gf:.2f} GFLOP = {gf / us * 1e3:.1f} TFLOP/s; scratch {wsf.numel() * 4 / 1e6:.1f} MB; "
              f"SS_GEMM_DW_WIDE={os.environ.get('SS_GEMM_DW_WIDE', '1')}", flush=True)


if __name__ == "__main__":
    main()
