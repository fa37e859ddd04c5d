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


def allreduce_flat_grads(flat: torch.Tensor, group=None, always: bool = False) -> None:
    """Sum the flat gradient bucket over ranks (one collective per step).  ``always`` issues the collective on a
    one-rank group too (a sum over one rank: the identity) -- how the RCCL leg is exercised on a one-GPU box."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and (always or dist.get_world_size(group) > 1):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


class Trainer:
    """``micro_batches`` > 1 splits this rank's clips into that many equal slices which travel through forward and
    backward on their own HIP streams.  The GRU recurrence is latency-bound and occupies only one CU per
    (16-clip slice, direction); with two slices in flight the ROI-CNN / GEMM kernels of one slice fill the CUs
    the other slice's recurrence leaves idle.  The arithmetic is unchanged: every slice divides its loss by the
    global batch and adds its gradient into the same flat bucket (float atomics).

    Measured on MI355X at B=256: 2 staggered slices gain 1 % (3.66 vs 3.71 ms/step) -- the overlap is
    real (the second slice's CNN forward runs under the first slice's recurrence) but the half-size persistent CNN
    launches on 224 CUs lose what it wins -- so the default stays 1."""

    # CUs left to the other micro-batch's recurrence while a persistent ROI-CNN kernel runs (2 directions x 8 slices)
    CNN_RESERVED_CUS = 32

    def __init__(self, model: BiGRUClassifier, lr: float = 3e-4, max_norm: float = 1.0,
                 label_smoothing: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8, world_size: int = 1,
                 process_group=None, dropout: bool = True, micro_batches: int = 1, always_allreduce: bool = False):
        if model.flat_params is None or not model.flat_params.is_cuda:
            raise RuntimeError("Trainer needs the model on a HIP device")
        L.load()
        self.model, self.lr, self.max_norm, self.ls = model, lr, max_norm, label_smoothing
        self.betas, self.eps = betas, eps
        self.world, self.group = world_size, process_group
        self.always_allreduce = always_allreduce
        # bench.py: when this is a list, every step appends a (start, end) HIP-event pair around the gradient all-reduce
        self.allreduce_events = None
        self.dropout = dropout
        dev = model.flat_params.device
        self.m = torch.zeros_like(model.flat_params)
        self.v = torch.zeros_like(model.flat_params)
        # [loss_sum, sumsq] fp32 and [correct] int32 live on the device
        self.scal = torch.zeros(2, device=dev, dtype=torch.float32)
        self.correct = torch.zeros(1, device=dev, dtype=torch.int32)
        self.step_count = 0
        self.rank = 0
        if world_size > 1:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                self.rank = dist.get_rank(process_group)
        self._bind_bucket()
        self.micro_batches = max(1, int(micro_batches))
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.micro_batches)] if self.micro_batches > 1 else []
        self.ev_start = torch.cuda.Event()
        self.ev_done = [torch.cuda.Event() for _ in self.streams]

    def _bind_bucket(self):
        """(Re)resolve everything that aliases the model's flat buckets.  ``model.to()/.cuda()/.float()`` keep the bucket
        when nothing moves; when they do re-allocate it (``model._bucket_version`` moves) the gradient views are taken
        again and the Adam moments follow the parameters to their device, so training carries on instead of
        accumulating into an orphaned bucket."""
        model = self.model
        flat = model.flat_params
        if self.m.device != flat.device or self.m.numel() != flat.numel():
            if self.m.numel() != flat.numel():
                raise RuntimeError("the model's parameter layout changed under the Trainer")
            self.m, self.v = self.m.to(flat.device), self.v.to(flat.device)
            self.scal, self.correct = self.scal.to(flat.device), self.correct.to(flat.device)
        self.G = model._views_of(model.flat_grads)
        model.attach_flat_grads()
        self._bucket_version = model._bucket_version

    def step(self, X: torch.Tensor, lengths: torch.Tensor, R: Optional[torch.Tensor], y: torch.Tensor,
             global_batch: Optional[int] = None):
        """One optimiser step on this rank's shard.  Returns (loss, correct) device tensors:
        loss = this shard's contribution to the global mean loss (sum over ranks = global loss).
        ``global_batch`` = clips of ALL ranks in this step; needed only when the shards are unequal (``shard_range``
        with a world size that does not divide the batch): the loss is divided by it, so the summed bucket is the
        global-mean gradient whatever each rank holds.  Default: B * world_size (equal shards)."""
        model, cfg = self.model, self.model.cfg
        if model._bucket_version != self._bucket_version:
            self._bind_bucket()
        B = X.shape[0]
        self.step_count += 1
        train = self.dropout and model.training
        # CE is divided by the GLOBAL clip count, so the summed bucket is the global-mean gradient
        denom = float(global_batch if global_batch is not None else B * self.world)
        # dropout streams must differ between ranks (clip b of every rank would otherwise draw the same Philox masks)
        base_seed = self.step_count * self.world + self.rank
        M = self.micro_batches if (self.micro_batches > 1 and B % self.micro_batches == 0 and B // self.micro_batches >= 16) else 1
        fused_prologue = (M == 1 and lengths.dtype == torch.int64 and lengths.is_cuda and lengths.is_contiguous()
                          and X.is_contiguous())
        if fused_prologue:
            # zero_grad, the scalars, the int32 lengths and the landmark half of the GRU input: one launch instead of six
            ws = model._workspace(X, R, train=True, slot=0)
            T = X.shape[1]
            L.call("ss_train_prologue", model.flat_grads.data_ptr(), model.flat_grads.numel(), self.scal.data_ptr(), 2,
                   self.correct.data_ptr(), lengths.data_ptr(), ws.lengths.data_ptr(), B,
                   X.data_ptr() if cfg.use_roi else None, cfg.x_dim, L.ptr(ws.Z), cfg.in_dim, B * T, cfg.x_dim,
                   L.ptr(getattr(ws, "frames", None)), cfg.roi_emb if cfg.use_roi else 0, L.stream())
        else:
            model.flat_grads.zero_()
            self.scal.zero_()
            self.correct.zero_()
        if M == 1:
            self._fwd_bwd(X, lengths, R, y, denom, train, seed=base_seed, slot=0, phase="both",
                          prologue_done=fused_prologue)
        else:
            n = B // M
            cur = torch.cuda.current_stream()
            self.ev_start.record(cur)
            L.call("ss_roi_cnn_set_max_workgroups", torch.cuda.get_device_properties(X.device).multi_processor_count - self.CNN_RESERVED_CUS)
            parts = [(X[m * n:(m + 1) * n], lengths[m * n:(m + 1) * n], None if R is None else R[m * n:(m + 1) * n],
                      y[m * n:(m + 1) * n]) for m in range(M)]
            # issue order fwd(0), fwd(1), ..., bwd(0), bwd(1), ...: the host never runs far ahead on one stream
            for phase in ("fwd", "bwd"):
                for m, (Xm, Lm, Rm, ym) in enumerate(parts):
                    with torch.cuda.stream(self.streams[m]):
                        if phase == "fwd":
                            self.streams[m].wait_event(self.ev_start)
                            model._workspace(Xm, Rm, train=True, slot=m + 1).stagger = True  # the next slice waits for its CNN
                            if m > 0 and cfg.use_roi:
                                # stagger: slice m starts its ROI-CNN when slice m-1 has left it for the recurrence, so the
                                # chip-filling kernels of one slice run beside the 16-CU recurrence of the other
                                prev = model._workspace(parts[m - 1][0], parts[m - 1][2], train=True, slot=m)
                                self.streams[m].wait_event(prev.ev_cnn_fwd)
                        self._fwd_bwd(Xm, Lm, Rm, ym, denom, train, seed=base_seed * M + m, slot=m + 1, phase=phase)
                        if phase == "bwd":
                            self.ev_done[m].record(self.streams[m])
            for ev in self.ev_done:
                cur.wait_event(ev)
            L.call("ss_roi_cnn_set_max_workgroups", 0)
        s = L.stream()
        if self.world > 1 or self.always_allreduce:
            if self.allreduce_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            allreduce_flat_grads(model.flat_grads, self.group, always=self.always_allreduce)
            if self.allreduce_events is not None:
                e1.record()
                self.allreduce_events.append((e0, e1))
        n_el = model.flat_grads.numel()
        L.call("ss_sumsq_f32", model.flat_grads.data_ptr(), n_el, self.scal.data_ptr() + 4, s)
        # d_logits already carries 1/(B*world), so the summed bucket IS the global-mean gradient
        L.call("ss_adam_clip", model.flat_params.data_ptr(), model.flat_grads.data_ptr(), self.m.data_ptr(),
               self.v.data_ptr(), n_el, self.scal.data_ptr() + 4, 1.0, self.max_norm, self.lr, self.betas[0],
               self.betas[1], self.eps, self.step_count, s)
        return self.scal[0], self.correct[0]

    def _fwd_bwd(self, X, lengths, R, y, denom, train, seed, slot, phase, prologue_done=False):
        """Forward + CE ("fwd"), backward ("bwd") or both of one micro-batch on the current stream."""
        model, cfg = self.model, self.model.cfg
        ws = model._workspace(X, R, train=True, slot=slot)
        P = model._param_dict()
        if phase in ("fwd", "both"):
            if not prologue_done:
                ws.lengths.copy_(lengths.to(torch.int32), non_blocking=True)
            E.forward(P, cfg, ws, X, R, train=train, stash=True, seed=seed, x_in_place=prologue_done,
                      ce=(y.data_ptr(), self.ls, denom, self.scal.data_ptr(), self.correct.data_ptr()))
        if phase in ("bwd", "both"):
            E.backward(P, self.G, cfg, ws, X, R, ws.d_logits, train=train, seed=seed)

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the last step's (pre-clip) gradient."""
        return self.scal[1].sqrt()
