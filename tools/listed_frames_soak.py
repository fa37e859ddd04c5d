#!/usr/bin/env python3
"""Soak for the listed-frame walk of the two fused ROI-CNN kernels (ss_roi_cnn_fwd_frames / _bwd_frames): random frame lists
(empty, one frame, everything, sparse, dense), random grid caps (so that workgroups get 0, 1 or many listed frames and the
prefetch pipeline runs over gaps), all three frame sizes; every round is compared with the full walk whose d_out is zero on the
frames left out.  Prints one line per size; exits 1 on the first disagreement.

    python tools/listed_frames_soak.py [rounds]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from silent_speech_amd import _lib as L  # noqa: E402

KEYS = [(8, 1, 3, 3), (8,), (16, 8, 3, 3), (16,), (24, 16, 3, 3), (24,), (32, 24), (32,)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    L.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    for (H, W) in ((64, 64), (48, 96), (32, 32)):
        N = 700
        P = [(torch.randn(*s, device=dev, generator=g) * (0.3 if len(s) > 1 else 0.05)).contiguous() for s in KEYS]
        R = torch.randint(0, 256, (N, H, W), device=dev, dtype=torch.uint8, generator=g)
        sizes = L.cnn_stash_sizes(H, W)
        n_a1, n_a2, n_i1, n_i2, n_m3, n_feat = sizes
        st = [torch.zeros(N, n_a1, device=dev), torch.zeros(N, n_i1, device=dev, dtype=torch.uint8), torch.zeros(N, n_a2, device=dev),
              torch.zeros(N, n_i2, device=dev, dtype=torch.uint8), torch.zeros(N, n_m3, device=dev, dtype=torch.uint8),
              torch.zeros(N, n_feat, device=dev)]
        worst = 0.0
        for r in range(rounds):
            kind = r % 6
            if kind == 0:
                pick = torch.zeros(N, dtype=torch.bool, device=dev)
            elif kind == 1:
                pick = torch.zeros(N, dtype=torch.bool, device=dev)
                pick[int(torch.randint(0, N, (1,), device=dev, generator=g))] = True
            elif kind == 2:
                pick = torch.ones(N, dtype=torch.bool, device=dev)
            else:
                pick = torch.rand(N, device=dev, generator=g) < (0.05, 0.5, 0.95)[kind - 3]
            listed = torch.nonzero(pick).flatten().to(torch.int32)
            frames = torch.cat([torch.tensor([len(listed)], dtype=torch.int32, device=dev), listed,
                                torch.full((N - len(listed),), 2 ** 30, dtype=torch.int32, device=dev)])
            d_out = torch.randn(N, 32, device=dev, generator=g)
            d_mask = (d_out * pick[:, None]).contiguous()
            cap = (0, 1, 3, 16, 64, 200)[int(torch.randint(0, 6, (1,), device=dev, generator=g))]
            res = []
            for fr, d in ((None, d_mask), (frames, d_out)):
                out = torch.full((N, 32), -3.0, device=dev)
                G = [torch.zeros_like(p) for p in P]
                L.call("ss_roi_cnn_set_max_workgroups", cap if fr is not None else 0)
                L.call("ss_roi_cnn_fwd_frames", R.data_ptr(), N, H, W, 1, *[p.data_ptr() for p in P], 32, out.data_ptr(), 32,
                       *[s.data_ptr() for s in st], sizes.ptr, L.ptr(fr), L.stream())
                L.call("ss_roi_cnn_bwd_frames", R.data_ptr(), N, H, W, 1, *[p.data_ptr() for p in P], 32, *[s.data_ptr() for s in st],
                       sizes.ptr, d.data_ptr(), 32, *[gg.data_ptr() for gg in G], L.ptr(fr), L.stream())
                L.call("ss_roi_cnn_set_max_workgroups", 0)
                res.append((out, G))
            torch.cuda.synchronize()
            (o0, G0), (o1, G1) = res
            ok = torch.equal(o1[pick], o0[pick]) and bool(torch.all(o1[~pick] == -3.0))
            for a, b in zip(G1, G0):
                e = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)
                worst = max(worst, e if len(listed) else float(a.abs().max()))
                ok = ok and (e < 3e-5 if len(listed) else float(a.abs().max()) == 0.0)
            if not ok:
                print(f"{H}x{W} round {r} kind {kind} cap {cap} listed {len(listed)}: MISMATCH (worst {worst:.2e})", flush=True)
                sys.exit(1)
        print(f"{H}x{W}: {rounds} rounds agree, worst gradient difference {worst:.2e} of the largest entry", flush=True)


if __name__ == "__main__":
    main()
