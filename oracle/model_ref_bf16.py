"""CPU restatement of the bf16-MFMA path of BASELINE config 5 (TEST INFRASTRUCTURE ONLY, see ``oracle/__init__.py``).

``model_ref.py`` is the f32 statement of the model (the reference's arithmetic; for the build-defined config-5 widths the
same module with one more conv block).  This file evaluates the SAME function with the operands of
every matrix product rounded to bfloat16 (nearest even) where the HIP kernels round them, accumulating in f32:

  * ROI CNN: the normalised frame, every conv weight and every pooled map between the layers are bf16 values
    (cnn_bf16.hip keeps them as bf16 in LDS / HBM); bias, ReLU, pooling, the global average and the fc layer are f32;
  * GRU: both operands of ``W_ih x`` and of ``W_hh h`` are rounded (gemm_bf16.hip, gru_bf16.hip); gates, the blend
    ``(1-z) n + z h`` (with the un-rounded previous state) and the biases are f32;
  * AttnPool, LayerNorm, head and the loss are f32 (the f32 kernels of tail.hip).

It separates two questions the GPU tests ask: "do the kernels compute this function?" (tight tolerance against this file)
and "how far is this function from the f32 model?" (the tolerance DESIGN.md states for config 5, measured on the CPU in
tests/test_oracle_bf16.py).  Parity of the config-5 MODEL against the reference is by construction only -- the reference has no
such model -- while every building block is pinned through model_ref.py at the reference's own widths.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import model_ref as MR

SD = MR.SD


def bf(x: torch.Tensor) -> torch.Tensor:
    """Round to bfloat16 and back: identity gradient (the rounding points are treated as straight-through)."""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def roi_cnn_bf16(r: torch.Tensor, sd: SD, prefix: str = "roi_cnn.") -> torch.Tensor:
    B, T, C, H, W = r.shape
    x = bf(r.reshape(B * T, C, H, W))
    n_conv = 0
    while f"{prefix}net.{3 * n_conv}.weight" in sd:
        n_conv += 1
    for i in range(n_conv):
        x = F.relu(F.conv2d(x, bf(sd[f"{prefix}net.{3 * i}.weight"]), sd[f"{prefix}net.{3 * i}.bias"], padding=1))
        if i < n_conv - 1:
            x = bf(F.max_pool2d(x, 2))
    x = x.mean(dim=(2, 3))
    x = F.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    return x.reshape(B, T, -1)


def gru_direction_bf16(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse: bool):
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi_all = F.linear(bf(x), bf(w_ih), b_ih)
    wb = bf(w_hh)
    h = x.new_zeros(B, H)
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        valid = (lengths > t).to(x.dtype).unsqueeze(1)
        gh = F.linear(bf(h), wb) + b_hh
        gi = gi_all[:, t]
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h_new = (1.0 - z) * n + z * h
        h = valid * h_new + (1.0 - valid) * h
        outs[t] = valid * h
    return torch.stack(outs, dim=1)


def forward(sd: SD, X: torch.Tensor, lengths: torch.Tensor, R: Optional[torch.Tensor] = None, *, roi_standardize: bool = True,
            gru_dropout_masks: Optional[list] = None, head_dropout_mask: Optional[torch.Tensor] = None):
    use_roi = "roi_cnn.fc.weight" in sd
    if use_roi:
        Z = torch.cat([X, roi_cnn_bf16(MR.roi_normalise(R, roi_standardize), sd)], dim=2)
    else:
        Z = X
    x = Z
    layers = MR.gru_layers_of(sd)
    for l in range(layers):
        p = f"gru."
        f = gru_direction_bf16(x, lengths, sd[f"{p}weight_ih_l{l}"], sd[f"{p}weight_hh_l{l}"], sd[f"{p}bias_ih_l{l}"],
                               sd[f"{p}bias_hh_l{l}"], reverse=False)
        b = gru_direction_bf16(x, lengths, sd[f"{p}weight_ih_l{l}_reverse"], sd[f"{p}weight_hh_l{l}_reverse"],
                               sd[f"{p}bias_ih_l{l}_reverse"], sd[f"{p}bias_hh_l{l}_reverse"], reverse=True)
        x = torch.cat([f, b], dim=2)
        if gru_dropout_masks is not None and l < layers - 1:
            x = x * gru_dropout_masks[l]
    out = x[:, : int(lengths.max())]
    pooled = MR.attn_pool(out, lengths, sd)
    return MR.head(pooled, sd, head_dropout_mask)


def loss_and_grads(sd: SD, X, lengths, R, y, *, roi_standardize: bool = True, label_smoothing: float = 0.05):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = forward(leaves, X, lengths, R, roi_standardize=roi_standardize)
    loss = MR.ce_label_smoothing(logits, y, label_smoothing)
    keys = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys])
    return loss.detach(), logits.detach(), dict(zip(keys, grads))
