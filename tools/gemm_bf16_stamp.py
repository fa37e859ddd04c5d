#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the bf16-operand GEMM spends its life (needs the -DSS_STAMP build:
python -m silent_speech_amd.build --stamp).  Thread 0 of workgroups 0..255 accumulates clock64() deltas."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SS_HOTPATH_LIB", os.path.join(ROOT, "silent_speech_amd", "libss_hotpath_stamp.so"))
from silent_speech_amd import _lib as L  # noqa: E402

NAMES = {15: "prologue", 1: "load issue + LDS reads + MFMAs", 2: "LDS store of the next tile (waits for its loads)", 3: "barrier", 4: "epilogue"}
INT_MAX = 2**31 - 1


def report(fn, title, reps):
    buf = np.zeros(512 * 24, np.uint64)
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(512, 24).astype(np.float64)[:64, :16] / reps
    tot = t.sum(1).mean()
    print(f"{title}: {tot:.0f} cycles per workgroup")
    for k, nm in NAMES.items():
        print(f"   {nm:52s} {t[:, k].mean():9.0f} cycles  {100 * t[:, k].mean() / tot:5.1f} %")


def main():
    lib = L.load()
    fn = lib.ss_debug_stamps_gemm_bf16
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    dev = "cuda"
    buf = np.zeros(512 * 24, np.uint64)
    for name, akc, bkc, M, Nn, K, lda, ldb, flags, splits, batch in (
            ("dW_hh rz, 64 workgroups alone (116 k tiles each)", 0, 0, 1024, 512, 7424, 2048, 1024, 5, 1, 2),
            ("dW_ih l1, splits 2 (60 k tiles each, 2 workgroups per CU)", 0, 0, 1536, 1024, 7680, 2048, 1024, 5, 2, 2),
            ("ih l1 (16 k tiles each)", 1, 1, 7680, 1536, 1024, 1024, 1024, 0, 1, 2)):
        a_rows, b_rows = (M if akc else K) + 64, (Nn if bkc else K) + 64
        A = torch.randint(-200, 200, (batch * a_rows, lda), device=dev, dtype=torch.int16)
        Bm = torch.randint(-200, 200, (batch * b_rows, ldb), device=dev, dtype=torch.int16)
        Cm = torch.zeros(batch, M, Nn, device=dev)
        fn(buf.ctypes.data)  # clear
        for _ in range(3):
            L.call("ss_gemm_bf16_batched", akc, bkc, M, Nn, K, A.data_ptr(), lda, INT_MAX, 0, 0, Bm.data_ptr(), ldb, INT_MAX, 0, 0,
                   Cm.data_ptr(), Nn, None, flags | 8, splits, batch, a_rows * lda, b_rows * ldb, M * Nn, 0, L.stream())
        report(fn, name, 3)


if __name__ == "__main__":
    main()
