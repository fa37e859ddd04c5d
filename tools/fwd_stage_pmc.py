#!/usr/bin/env python3
"""Diagnostic: launches roi_cnn_fwd (with stash) a few times on config-2's 7 680 frames and nothing else, for counter runs on
variant libraries built with -DSS_FWD_STOP=k (a frame ends behind stage k; roi_cnn.hip).  Differences between the variants'
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS are one stage's share.

    tools/mkvariant.sh stop1 silent_speech_amd/csrc/roi_cnn.hip roi_cnn "-DSS_FWD_STOP=1"
    export SS_HOTPATH_LIB=silent_speech_amd/_ab/libstop1.so
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d out -o x -- python3 tools/fwd_stage_pmc.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from silent_speech_amd import _lib as L  # noqa: E402

KEYS = [(8, 1, 3, 3), (8,), (16, 8, 3, 3), (16,), (24, 16, 3, 3), (24,), (32, 24), (32,)]


def main():
    H, W = (int(v) for v in os.environ.get("AB_ROI", "64,64").split(","))
    N = int(os.environ.get("AB_FRAMES", "7680"))
    L.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    P = [(torch.randn(*s, device=dev, generator=g) * (0.3 if len(s) > 1 else 0.05)).contiguous() for s in KEYS]
    R = torch.randint(0, 256, (N, H, W), device=dev, dtype=torch.uint8, generator=g)
    sizes = L.cnn_stash_sizes(H, W)
    n_a1, n_a2, n_i1, n_i2, n_m3, n_feat = sizes
    st = [torch.zeros(N, n_a1, device=dev), torch.zeros(N, n_i1, device=dev, dtype=torch.uint8), torch.zeros(N, n_a2, device=dev),
          torch.zeros(N, n_i2, device=dev, dtype=torch.uint8), torch.zeros(N, n_m3, device=dev, dtype=torch.uint8),
          torch.zeros(N, n_feat, device=dev)]
    out = torch.zeros(N, 32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for k in range(6):
        if k == 1:
            ev[0].record()
        L.call("ss_roi_cnn_fwd_stash", R.data_ptr(), N, H, W, 1, *[p.data_ptr() for p in P], 32, out.data_ptr(), 32,
               *[s.data_ptr() for s in st], sizes.ptr, L.stream())
    ev[1].record()
    torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("SS_HOTPATH_LIB", "libss_hotpath.so")), f"{ev[0].elapsed_time(ev[1]) / 5 * 1e3:.1f} us per launch", flush=True)


if __name__ == "__main__":
    main()
