#!/usr/bin/env python3
"""Diagnostic: where a time step of the persistent bf16 recurrence goes (needs the -DSS_STAMP build).  Thread 0 of every workgroup
accumulates clock64() deltas; cycles per step = table / (launches x T)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SS_HOTPATH_LIB", os.path.join(ROOT, "silent_speech_amd", "libss_hotpath_stamp.so"))
from silent_speech_amd import _lib as L  # noqa: E402

B, T, H = 256, 30, 512
N = B * T


def main():
    lib = L.load()
    fn = lib.ss_debug_stamps_gru_bf16
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    dev = "cuda"
    gi = torch.randn(2, N, 3 * H, device=dev) * 0.5
    whh = torch.randint(-100, 100, (2, 3 * H, H), device=dev, dtype=torch.int16)
    whht = torch.randint(-100, 100, (2, H, 3 * H), device=dev, dtype=torch.int16)
    bhh = torch.zeros(2, 3 * H, device=dev)
    lens = torch.full((B,), T, device=dev, dtype=torch.int32)
    out, save = torch.empty(N, 2 * H, device=dev), torch.empty(2, N, 4, H, device=dev)
    out_bf, out_dr = torch.empty(N, 2 * H, device=dev, dtype=torch.int16), torch.empty(N, 2 * H, device=dev, dtype=torch.int16)
    nb = C.c_long(0)
    lib.ss_gru_bf16_ws_bytes(B, H, C.byref(nb))
    ws = torch.empty(nb.value, device=dev, dtype=torch.uint8)
    lib.ss_gru_bf16_sync_bytes(B, T, H, C.byref(nb))
    sync = torch.zeros(nb.value // 4, device=dev, dtype=torch.int32)
    buf = np.zeros(512 * 24, np.uint64)
    reps = 3

    def report(title, names):
        assert fn(buf.ctypes.data) == 0
        t = buf.reshape(512, 24).astype(np.float64)[:256, :16] / reps / T
        tot = sum(t[:, k].mean() for k in names)
        print(f"{title}: {tot:.0f} cycles per step")
        for k, nm in names.items():
            print(f"   {nm:60s} {t[:, k].mean():8.0f}  {100 * t[:, k].mean() / tot:5.1f} %   (slowest workgroup {t[:, k].max():6.0f})")

    fn(buf.ctypes.data)
    for _ in range(reps):
        L.call("ss_gru_bf16_fwd", gi.data_ptr(), whh.data_ptr(), bhh[0].data_ptr(), bhh[1].data_ptr(), lens.data_ptr(), B, T, H, out.data_ptr(),
               save.data_ptr(), out_bf.data_ptr(), out_dr.data_ptr(), 0.1, 1, 1 << 40, ws.data_ptr(), sync.data_ptr(), L.nbytes(sync), L.stream())
    report("forward", {15: "loop top + gi loads issued", 0: "sweep of the previous state", 1: "panel write + barrier", 2: "fragment reads + 48 MFMAs",
                       3: "gates + publish", 4: "out / bf16 copies / save stores"})
    d_out = torch.randn(N, 2 * H, device=dev)
    dG_bf = torch.empty(2, N, 4, H, device=dev, dtype=torch.int16)
    gb = [torch.zeros(3 * H, device=dev) for _ in range(4)]
    for _ in range(reps):
        L.call("ss_gru_bf16_bwd", d_out.data_ptr(), out.data_ptr(), save.data_ptr(), whht.data_ptr(), lens.data_ptr(), B, T, H, None,
               dG_bf.data_ptr(), 0.1, 1, 1 << 40, *[t_.data_ptr() for t_ in gb], ws.data_ptr(), sync.data_ptr(), L.nbytes(sync), L.stream())
    report("backward", {15: "loop top", 5: "gate gradients + d_pre panel write", 6: "barrier", 7: "fragment reads + 48 MFMAs + publish", 8: "d_g stores + sums",
                        9: "next step's inputs (loads, Philox)", 10: "sweep of the partial sums", 11: "sum + LDS + barrier + d h"})
    torch.cuda.synchronize()
    assert int(sync[2]) == 0


if __name__ == "__main__":
    main()
