"""Deterministic state_dict / input generators shared by ``make_golden.py`` (which feeds them to
the reference classes) and by the tests (which feed them to the oracle and the HIP path).

Keeping only a seed in each fixture instead of 1.08 M weights keeps the fixtures small; the
CPU generator of one torch version is deterministic, and the fixture stores a checksum of the
generated weights so a generator drift would be caught rather than silently mis-compared.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch


def param_shapes(x_dim: int, num_classes: int, use_roi: bool, roi_emb: int = 32, hidden: int = 192,
                 gru_layers: int = 2, cnn_channels=(8, 16, 24)) -> "OrderedDict[str, tuple]":
    """Key -> shape in the reference's state_dict order.  ``cnn_channels`` other than the reference's
    (8, 16, 24) gives the build-defined wider CNN of BASELINE config 5 (conv blocks at Sequential indices 0, 3, 6, 9, ...)."""
    s = OrderedDict()
    if use_roi:
        chans = (1,) + tuple(cnn_channels)
        for i in range(len(cnn_channels)):
            s[f"roi_cnn.net.{3 * i}.weight"] = (chans[i + 1], chans[i], 3, 3)
            s[f"roi_cnn.net.{3 * i}.bias"] = (chans[i + 1],)
        s["roi_cnn.fc.weight"] = (roi_emb, chans[-1])
        s["roi_cnn.fc.bias"] = (roi_emb,)
    in_dim = x_dim + (roi_emb if use_roi else 0)
    for l in range(gru_layers):
        lin = in_dim if l == 0 else 2 * hidden
        for suf in ("", "_reverse"):
            s[f"gru.weight_ih_l{l}{suf}"] = (3 * hidden, lin)
            s[f"gru.weight_hh_l{l}{suf}"] = (3 * hidden, hidden)
            s[f"gru.bias_ih_l{l}{suf}"] = (3 * hidden,)
            s[f"gru.bias_hh_l{l}{suf}"] = (3 * hidden,)
    s["pool.score.weight"] = (1, 2 * hidden)
    s["pool.score.bias"] = (1,)
    s["head.0.weight"] = (2 * hidden,)
    s["head.0.bias"] = (2 * hidden,)
    s["head.1.weight"] = (128, 2 * hidden)
    s["head.1.bias"] = (128,)
    s["head.4.weight"] = (num_classes, 128)
    s["head.4.bias"] = (num_classes,)
    return s


def make_state_dict(seed: int, x_dim: int, num_classes: int, use_roi: bool, roi_emb: int = 32,
                    hidden: int = 192, gru_layers: int = 2, gain: float = 1.5, cnn_channels=(8, 16, 24)) -> "OrderedDict[str, torch.Tensor]":
    """U(-a, a) with a = gain/sqrt(fan_in) per tensor (biases use the matching weight's fan-in,
    LayerNorm gamma is 1 + U(-.2,.2)).  ``gain`` > 1 keeps gates away from the trivial regime so
    parity errors are visible in the logits."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    shapes = param_shapes(x_dim, num_classes, use_roi, roi_emb, hidden, gru_layers, cnn_channels)
    for k, shp in shapes.items():
        if k.startswith("gru."):
            fan = hidden
        elif k.endswith("weight"):
            fan = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
        else:
            wk = k[: -len("bias")] + "weight"
            fan = int(np.prod(shapes[wk][1:])) if len(shapes[wk]) > 1 else shapes[wk][0]
        a = gain / np.sqrt(fan)
        t = (torch.rand(shp, generator=g) * 2 - 1) * a
        if k == "head.0.weight":
            t = 1.0 + (torch.rand(shp, generator=g) * 2 - 1) * 0.2
        sd[k] = t.float().contiguous()
    return sd


def checksum(sd) -> float:
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def make_inputs(seed: int, B: int, T: int, x_dim: int, num_classes: int, roi_hw=None, lengths=None):
    g = torch.Generator().manual_seed(seed + 7919)
    X = torch.randn(B, T, x_dim, generator=g) * 0.7
    y = torch.randint(0, num_classes, (B,), generator=g)
    if lengths is None:
        lengths = torch.randint(1, T + 1, (B,), generator=g)
        lengths[0] = T
    lengths = torch.as_tensor(lengths, dtype=torch.int64)
    R = None
    if roi_hw is not None:
        H, W = roi_hw
        R = torch.randint(0, 256, (B, T, H, W), generator=g, dtype=torch.uint8)
        # smooth-ish structure so conv responses are not pure noise: blend with a ramp
        ramp = (torch.arange(W).view(1, 1, 1, W) * 255 // max(1, W - 1)).to(torch.int32)
        R = ((R.to(torch.int32) + ramp) // 2).to(torch.uint8)
    return X, lengths, R, y


def reduce_tensor(t: torch.Tensor, full_below: int = 4096, stride: int = 53) -> np.ndarray:
    """Small tensors in full; big ones as [sum, l2, strided sample...] (float64)."""
    f = t.detach().double().reshape(-1)
    if f.numel() <= full_below:
        return f.numpy().copy()
    return np.concatenate([[float(f.sum()), float(f.norm())], f[::stride].numpy()])
