#!/usr/bin/env python3
"""Diagnostic: achieved TFLOP/s of ss_gemm_f32 on the shapes of the path and on a large square problem."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from silent_speech_amd import engine as E  # noqa: E402


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
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = torch.device("cuda")
    cases = [  # name, a_kc, b_kc, M, N, K, splits, atomic
        ("square 4096 NT (kc,kc)", 1, 1, 4096, 4096, 4096, 1, False),
        ("square 4096 NN (kc,rows)", 1, 0, 4096, 4096, 4096, 1, False),
        ("square 4096 TN (rows,rows)", 0, 0, 4096, 4096, 4096, 1, False),
        ("ih l0  7680x1152x116", 1, 1, 7680, 1152, 116, 1, False),
        ("ih l1  7680x1152x384", 1, 1, 7680, 1152, 384, 1, False),
        ("dX l1  7680x384x576", 1, 0, 7680, 384, 576, 1, False),
        ("dX l0  7680x116x576", 1, 0, 7680, 116, 576, 1, False),
        ("dWih1  576x384x7680 s21 plain", 0, 0, 576, 384, 7680, 21, False),
        ("dWih1  576x384x7680 s21 atomic", 0, 0, 576, 384, 7680, 21, True),
        ("dWih1  576x384x7680 s8 atomic", 0, 0, 576, 384, 7680, 8, True),
        ("dWhh   384x192x7424 s42 atomic", 0, 0, 384, 192, 7424, 42, True),
        ("dWhh   384x192x7424 s14 atomic", 0, 0, 384, 192, 7424, 14, True),
    ]
    cases += [(f"dWih1  576x384x7680 s{k} ws", 0, 0, 576, 384, 7680, k, "ws") for k in (4, 6, 8, 12, 16, 21)]
    cases += [(f"dWhh   384x192x7424 s{k} ws", 0, 0, 384, 192, 7424, k, "ws") for k in (8, 14, 21, 42)]
    cases += [("dX l0  7680x32x576", 1, 0, 7680, 32, 576, 1, False)]
    scratch = torch.empty(64 << 20, device=dev)
    for name, akc, bkc, M, N, K, splits, atomic in cases:
        A = torch.randn((M, K) if akc else (K, M), device=dev)
        B = torch.randn((N, K) if bkc else (K, N), device=dev)
        Cm = torch.zeros(M, N, device=dev)
        lda = K if akc else M
        ldb = K if bkc else N

        def run():
            E.gemm(akc, bkc, M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, Cm.data_ptr(), N,
                   accumulate=bool(atomic) or splits > 1, atomic=atomic is True, splits=splits,
                   splitk_ws=scratch if atomic == "ws" else None)

        t = timed(run)
        print(f"{name:34s} {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
