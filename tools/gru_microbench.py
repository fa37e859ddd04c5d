#!/usr/bin/env python3
"""Micro-benchmark of the GRU recurrence kernels (forward with / without the backward stash, BPTT)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from silent_speech_amd import _lib as L  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000.0


def main():
    dev = torch.device("cuda")
    H = 192
    # (T = 1, 2, 4: the intercept of launch time against T is what a launch costs outside its steps)
    for B, T, split in ((256, 30, False), (256, 1, True), (256, 2, True), (256, 4, True), (256, 30, True), (128, 30, True), (16, 30, True), (256, 60, True)):
        N = B * T
        nb = L.gru_sync_bytes(B, T, H) if split else 0
        sync_ws = torch.zeros(nb // 4, device=dev, dtype=torch.int32) if nb else None
        gi = torch.randn(2, N, 3 * H, device=dev) * 0.5
        w = [torch.randn(3 * H, H, device=dev) * 0.07 for _ in range(2)]
        b = [torch.randn(3 * H, device=dev) * 0.07 for _ in range(2)]
        lens = torch.full((B,), T, device=dev, dtype=torch.int32)
        out = torch.empty(N, 2 * H, device=dev)
        save = torch.empty(2, N, 4, H, device=dev)
        dout = torch.randn(N, 2 * H, device=dev)
        dg = torch.empty(2, N, 4, H, device=dev)
        s = L.stream()

        def fwd(sv):
            L.call("ss_gru_fwd", gi.data_ptr(), w[0].data_ptr(), w[1].data_ptr(), b[0].data_ptr(), b[1].data_ptr(),
                   lens.data_ptr(), B, T, H, out.data_ptr(), sv, L.ptr(sync_ws), s)

        def bwd():
            L.call("ss_gru_bwd", dout.data_ptr(), out.data_ptr(), save.data_ptr(), w[0].data_ptr(), w[1].data_ptr(),
                   lens.data_ptr(), B, T, H, dg.data_ptr(), 0.0, 0, 0, None, None, None, None, L.ptr(sync_ws), s)

        t1, t2, t3 = timed(lambda: fwd(save.data_ptr())), timed(lambda: fwd(None)), timed(bwd)
        errs = int(sync_ws[2]) if sync_ws is not None else 0
        fast = int(sync_ws[3]) if sync_ws is not None else 0
        print(f"B={B:4d} T={T} {'multi-CU' if sync_ws is not None else 'one-CU  '} (wait time-outs {errs}, same-XCD workgroup launches {fast}): fwd+stash {t1:7.1f} us ({t1 / T:5.2f}/step)  fwd {t2:7.1f} us ({t2 / T:5.2f}/step)  "
              f"bwd {t3:7.1f} us ({t3 / T:5.2f}/step)")


if __name__ == "__main__":
    main()
