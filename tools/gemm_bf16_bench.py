#!/usr/bin/env python3
"""Diagnostic: the config-5 GRU-layer GEMM shapes of ss_gemm_bf16_batched one at a time (bf16 operands in HBM), with and
without K slices / float atomics.  python tools/gemm_bf16_bench.py   (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import _lib as L  # noqa: E402

INT_MAX = 2**31 - 1
N, H, T, B = 7680, 512, 30, 256


def run(name, akc, bkc, M, Nn, K, lda, ldb, flags, splits, batch, a_map=(INT_MAX, 0, 0), b_map=(INT_MAX, 0, 0), sa=0, sb=0, sc=0, reps=20):
    dev = "cuda"
    a_rows = (M if akc else K) + 64
    b_rows = (Nn if bkc else K) + 64
    A = torch.randint(-200, 200, (batch * a_rows * 2, lda), device=dev, dtype=torch.int16)
    Bm = torch.randint(-200, 200, (batch * b_rows * 2, ldb), device=dev, dtype=torch.int16)
    C = torch.zeros(batch, M, Nn, device=dev)
    if flags & 16:
        gf_scale = 1
    sa = sa or a_rows * lda
    sb = sb or b_rows * ldb
    args = (akc, bkc, M, Nn, K, A.data_ptr(), lda, *a_map, Bm.data_ptr(), ldb, *b_map, C.data_ptr(), Nn, None, flags | 8, splits, batch,
            sa, sb, M * Nn, 0, L.stream())
    for _ in range(3):
        L.call("ss_gemm_bf16_batched", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("ss_gemm_bf16_batched", *args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = 2.0 * M * Nn * K * batch / 1e9
    print(f"{name:34s} M={M:5d} N={Nn:5d} K={K:5d} x{batch} splits={splits:2d} flags={flags}: {ms * 1e3:7.1f} us  {gf / ms:7.1f} TFLOP/s", flush=True)


def run_group(name, l, reps=20):
    """The weight gradients of GRU layer l (config-5 shapes) as one grouped stream-K launch + its reduce launch."""
    from types import SimpleNamespace

    from silent_speech_amd import engine_bf16 as E

    cfg = SimpleNamespace(hidden=H, in_dim=148)
    Kp = 152 if l == 0 else 2 * H
    dG = torch.randint(-200, 200, (2 * N + 64, 4 * H), device="cuda", dtype=torch.int16)
    lin = torch.randint(-200, 200, (N + 64, Kp), device="cuda", dtype=torch.int16)
    hp = torch.randint(-200, 200, (N + 64, 2 * H), device="cuda", dtype=torch.int16)
    K = 148 if l == 0 else 2 * H
    g_ih = torch.zeros(2, 3 * H, K, device="cuda")
    g_hh = torch.zeros(2, 3 * H, H, device="cuda")
    pr = E.dw_problems(cfg, B, T, l, Kp, dG.data_ptr() + 4 * H * 2, lin.data_ptr(), hp.data_ptr(), (g_ih.data_ptr(), 3 * H * K),
                       (g_hh.data_ptr(), 3 * H * H))
    ws = torch.empty(L.gemm_group_ws_floats(pr, bf16=True), device="cuda")
    arr, n = L.gemm_group(pr)
    for _ in range(3):
        L.call("ss_gemm_bf16_splitk_group", arr, n, ws.data_ptr(), ws.numel(), L.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("ss_gemm_bf16_splitk_group", arr, n, ws.data_ptr(), ws.numel(), L.stream())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = sum(2.0 * q.M * q.N * q.K * q.batch for q in pr) / 1e9
    print(f"{name:34s} {len(pr)} problems, ws {ws.numel() * 4 / 1e6:.0f} MB: {ms * 1e3:7.1f} us  {gf / ms:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "xcd":  # tiles dealt to the XCDs in contiguous ranges (default) against as dispatched (flag 2048)
        for rep in range(2):
            for fl in (0, 2048):
                run("ih l1 (store)", 1, 1, N, 3 * H, 2 * H, 2 * H, 2 * H, fl, 1, 2)
                run("dX l1 (K-concatenated, store)", 1, 0, N, 2 * H, 3 * H, 4 * H, 2 * H, 16 | fl, 1, 2)
                run("dW_ih l1 (2 slices, atomics)", 0, 0, 3 * H, 2 * H, N, 4 * H, 2 * H, 5 | fl, 2, 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        run_group("dW group l1", 1)
        run_group("dW group l0", 0)
        run("ih l1 (store)", 1, 1, N, 3 * H, 2 * H, 2 * H, 2 * H, 0, 1, 2)
        run("ih l0 (K=152)", 1, 1, N, 3 * H, 152, 152, 152, 0, 1, 2)
        run("dX l1 (K-concatenated, store)", 1, 0, N, 2 * H, 3 * H, 4 * H, 2 * H, 16, 1, 2)
        run("dX l0 (N=68, K-concatenated)", 1, 0, N, 68, 3 * H, 4 * H, 152, 16, 1, 2)
        for sp in (1, 2, 4, 8):
            run("dX l0 (N=68, atomics)", 1, 0, N, 68, 3 * H, 4 * H, 152, 5, sp, 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "diag":
        for rep in range(2):
            for fl in (5, 5 | 256, 5 | 512, 5 | 768):
                for st in (0, 1024):
                    run("dW_ih l1", 0, 0, 3 * H, 2 * H, N, 4 * H, 2 * H, fl | st, 1, 2)
            for fl in (0, 256, 512, 768):
                for st in (0, 1024):
                    run("ih l1 (store)", 1, 1, N, 3 * H, 2 * H, 2 * H, 2 * H, fl | st, 1, 2)
            for st in (0, 1024):
                run("dX l1 (K-concatenated, store)", 1, 0, N, 2 * H, 3 * H, 4 * H, 2 * H, 16 | st, 1, 2)
        sys.exit(0)
    run("ih l1 (store)", 1, 1, N, 3 * H, 2 * H, 2 * H, 2 * H, 0, 1, 2)
    run("ih l1 (one direction)", 1, 1, N, 3 * H, 2 * H, 2 * H, 2 * H, 0, 1, 1)
    run("dX l1 (atomic)", 1, 0, N, 2 * H, 3 * H, 4 * H, 2 * H, 5, 1, 2)
    run("dX l1 (plain accumulate, 1 dir)", 1, 0, N, 2 * H, 3 * H, 4 * H, 2 * H, 1, 1, 1)
    run("dX l1 (K-concatenated, store)", 1, 0, N, 2 * H, 3 * H, 4 * H, 2 * H, 16, 1, 2)
    for sp in (1, 2, 3, 5, 8):
        run("dW_ih l1", 0, 0, 3 * H, 2 * H, N, 4 * H, 2 * H, 5, sp, 2)
    for sp in (1, 4, 8, 16):
        run("dW_hh rz (remap)", 0, 0, 2 * H, H, B * (T - 1), 4 * H, 2 * H, 5, sp, 2, a_map=(T - 1, T, 1), b_map=(T - 1, T, 0))
    for sp in (4, 8, 16, 32):
        run("dW_hh n (remap)", 0, 0, H, H, B * (T - 1), 4 * H, 2 * H, 5, sp, 2, a_map=(T - 1, T, 1), b_map=(T - 1, T, 0))
    run("dW_ih l0 (N=148)", 0, 0, 3 * H, 148, N, 4 * H, 152, 5, 8, 2)
