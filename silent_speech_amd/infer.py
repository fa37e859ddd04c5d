"""Forward-only serving path: a fixed-shape forward captured once into a hipGraph and replayed.

The shipped live script classifies one press-to-record clip at batch 1 (/root/reference/live_infer_official.py:344-359);
the sliding-window variant (inactive/live_feed.py:155-213) pads a rolling ``deque(maxlen=max_t)`` and recomputes the
whole window every two frames, with no state reuse (a BiGRU's reverse direction cannot reuse state across shifted
windows).  Served at scale that is many independent fixed-length windows per call -- one static shape, which is what a
graph wants: ~25 kernel launches collapse into one replay (BASELINE.json config 4: T=60, B=4096, forward-only).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import engine as E
from .model import BiGRUClassifier


class GraphedInference:
    """``g = GraphedInference(model, B, T, roi_hw); logits = g(X, lengths, R)``.

    Inputs are copied into static device buffers (or written there directly through ``g.X / g.lengths / g.R``) and
    the captured forward is replayed; ``logits`` is the static output buffer (clone it to keep it across calls)."""

    def __init__(self, model: BiGRUClassifier, B: int, T: int, roi_hw=None, warmup: int = 2, topk: int = 0):
        if model.flat_params is None or not model.flat_params.is_cuda:
            raise RuntimeError("GraphedInference needs the model on a HIP device")
        if model.use_roi and roi_hw is None:
            raise RuntimeError("use_roi=True needs roi_hw=(H, W)")
        self.model, self.B, self.T = model, B, T
        dev = model.flat_params.device
        cfg = model.cfg
        self.X = torch.zeros(B, T, cfg.x_dim, device=dev)
        self.lengths = torch.full((B,), T, device=dev, dtype=torch.int32)
        self.R = torch.zeros(B, T, *roi_hw, device=dev, dtype=torch.uint8) if model.use_roi else None
        self.ws = E.make_workspace(cfg, B, T, tuple(roi_hw) if roi_hw else None, dev, train=False)
        # topk > 0: softmax + the k most probable classes of every window are part of the captured graph
        # (live_infer_official.py:223-226 for every window), read from ``top_probs`` / ``top_idx`` after a call
        self.topk = min(int(topk), cfg.num_classes)
        self.top_probs = torch.zeros(B, self.topk, device=dev) if self.topk else None
        self.top_idx = torch.zeros(B, self.topk, device=dev, dtype=torch.int32) if self.topk else None
        self._P = model._param_dict()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture: first-launch attribute setup, allocator settling
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.logits = self._run()

    def _run(self) -> torch.Tensor:
        self.ws.lengths.copy_(self.lengths)
        logits = E.forward(self._P, self.model.cfg, self.ws, self.X, self.R, train=False, stash=False)
        if self.topk:
            E.L.call("ss_softmax_topk", logits.data_ptr(), self.B, self.model.cfg.num_classes, self.topk,
                     self.top_probs.data_ptr(), self.top_idx.data_ptr(), E.L.stream())
        return logits

    def __call__(self, X: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None,
                 R: Optional[torch.Tensor] = None) -> torch.Tensor:
        if X is not None:
            self.X.copy_(X, non_blocking=True)
        if lengths is not None:
            self.lengths.copy_(lengths.to(torch.int32), non_blocking=True)
        if R is not None:
            self.R.copy_(R, non_blocking=True)
        self.graph.replay()
        return self.logits
