"""Fused training step: forward -> CE(label smoothing) -> backward -> [RCCL all-reduce] -> clip -> Adam.

Counterpart of the loop body of /root/reference/train_model_official.py:433-443 with
``Adam(lr=3e-4)`` (:403), ``CrossEntropyLoss(label_smoothing=0.05)`` (:405) and
``clip_grad_norm_(.., 1.0)`` (:438).  No autograd graph, no host synchronisation inside a step: the
loss and the hit count stay on the device until the caller asks for them.

Data parallelism (the reference has none): one process per GPU, each rank holds a full replica and
takes its shard of the clips; the only exchange is ONE all-reduce (sum) of the flat fp32 gradient
bucket between backward and clip, so the global-norm clip sees the gradient of the whole global
batch exactly as a single process would.  ``torch.distributed`` backend "nccl" is RCCL on ROCm.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import engine as E
from .model import BiGRUClassifier


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous clip shard of rank ``rank``: [lo, hi).  Equal shards when world | n_items."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def allreduce_flat_grads(flat: torch.Tensor, group=None) -> None:
    """Sum the flat gradient bucket over ranks (one collective per step)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


class Trainer:
    def __init__(self, model: BiGRUClassifier, lr: float = 3e-4, max_norm: float = 1.0,
                 label_smoothing: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8, world_size: int = 1,
                 process_group=None, dropout: bool = True):
        if model.flat_params is None or not model.flat_params.is_cuda:
            raise RuntimeError("Trainer needs the model on a HIP device")
        L.load()
        self.model, self.lr, self.max_norm, self.ls = model, lr, max_norm, label_smoothing
        self.betas, self.eps = betas, eps
        self.world, self.group = world_size, process_group
        self.dropout = dropout
        dev = model.flat_params.device
        self.m = torch.zeros_like(model.flat_params)
        self.v = torch.zeros_like(model.flat_params)
        # [loss_sum, sumsq] fp32 and [correct] int32 live on the device
        self.scal = torch.zeros(2, device=dev, dtype=torch.float32)
        self.correct = torch.zeros(1, device=dev, dtype=torch.int32)
        self.step_count = 0
        self.G = model._views_of(model.flat_grads)
        model.attach_flat_grads()

    def step(self, X: torch.Tensor, lengths: torch.Tensor, R: Optional[torch.Tensor], y: torch.Tensor):
        """One optimiser step on this rank's shard.  Returns (loss, correct) device tensors:
        loss = this shard's contribution to the global mean loss (sum over ranks = global loss)."""
        model, cfg = self.model, self.model.cfg
        s = L.stream()
        B = X.shape[0]
        ws = model._workspace(X, R, train=True)
        ws.lengths.copy_(lengths.to(torch.int32), non_blocking=True)
        P = model._param_dict()
        self.step_count += 1
        seed = self.step_count
        model.flat_grads.zero_()
        self.scal.zero_()
        self.correct.zero_()
        train = self.dropout and model.training
        logits = E.forward(P, cfg, ws, X, R, train=train, stash=True, seed=seed)
        L.call("ss_ce_ls_fwd_bwd", logits.data_ptr(), y.data_ptr(), B, cfg.num_classes, self.ls,
               float(B * self.world), ws.d_logits.data_ptr(), self.scal.data_ptr(), self.correct.data_ptr(), s)
        E.backward(P, self.G, cfg, ws, X, R, ws.d_logits, train=train, seed=seed)
        if self.world > 1:
            allreduce_flat_grads(model.flat_grads, self.group)
        n = model.flat_grads.numel()
        L.call("ss_sumsq_f32", model.flat_grads.data_ptr(), n, self.scal.data_ptr() + 4, s)
        # d_logits already carries 1/(B*world), so the summed bucket IS the global-mean gradient
        L.call("ss_adam_clip", model.flat_params.data_ptr(), model.flat_grads.data_ptr(), self.m.data_ptr(),
               self.v.data_ptr(), n, self.scal.data_ptr() + 4, 1.0, self.max_norm, self.lr, self.betas[0],
               self.betas[1], self.eps, self.step_count, s)
        return self.scal[0], self.correct[0]

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the last step's (pre-clip) gradient."""
        return self.scal[1].sqrt()
