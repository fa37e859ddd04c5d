"""Kernel orchestration of the hot path: explicit forward / backward over raw device buffers.

This is the host side of the C ABI (``include/ss_hotpath.h``): it owns the workspace layout in
HBM and issues the HIP kernels on torch's current stream.  torch is used for device memory and
streams only -- there is no aten compute op on this path and no CPU fallback.

Data layout (fp32 row-major unless noted; N = B*T frames):
    Z      (N, x_dim+E)      landmark features | ROI embedding   (the torch.cat, fused)
    gi_l   (2, N, 3H)        W_ih x + b_ih per direction, layer l
    out_l  (N, 2H)           GRU layer output, forward | reverse halves
    save_l (2, N, 4, H)      r, z, n, W_hn h + b_hn                (training only)
    dG_l   (2, N, 4, H)      d gi_r, d gi_z, d gi_n, d(W_hn h+b_hn) (training only)
    CNN stash: pooled maps a1 (N,8,H/2,W/2), a2 (N,16,H/4,W/4), pool argmaxes (u8), conv3 sign
    mask (u8), averaged features (N,24)                            (training only)
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib as L

INT_MAX = 2**31 - 1
# bench.py turns this off for its per-kernel timing pass: HIP-event pairs only bracket a kernel's own run time when
# every launch sits on one stream
JOIN_BEFORE_CNN_BWD = True
# weight-gradient GEMMs of layer l > 0 are queued behind its d layer_in GEMM, so they run beside the recurrence of layer l - 1
# (latency-bound) rather than beside that GEMM (MFMA-bound like them): 2.30 vs 2.32 ms/step
SIDE_AFTER_DX = os.environ.get("SS_SIDE_AFTER_DX", "1") == "1"
USE_SPLIT_GRU = True
FUSE_GRU_DROPOUT = os.environ.get("SS_FUSE_GRU_DROPOUT", "1") != "0"  # 0: the inter-layer dropout as its own launch (ss_dropout)
# K slices of the d W_hh GEMMs (18 / 12 output tiles): fewer, longer slices than for d W_ih halve the scratch traffic of
# their reduce passes (measured 512: -0.3 % on the step against 768; 384 and 256: +0.4 %)
_HH_TARGET = int(os.environ.get("SS_SPLITK_TARGET_HH", "512"))
USE_SIDE_STREAM = os.environ.get("SS_NO_SIDE_STREAM", "0") != "1"


@dataclass
class Config:
    x_dim: int
    num_classes: int
    use_roi: bool
    roi_emb: int = 32
    hidden: int = 192
    gru_layers: int = 2
    roi_standardize: bool = True
    gru_dropout: float = 0.1
    head_dropout: float = 0.2
    head_mid: int = 128
    ln_eps: float = 1e-5
    cnn_channels: tuple = (8, 16, 24)
    precision: str = "f32"  # "bf16": the wide-model path of engine_bf16.py (BASELINE config 5)

    @property
    def in_dim(self) -> int:
        return self.x_dim + (self.roi_emb if self.use_roi else 0)


def gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, Cm, ldc, bias=None, accumulate=False, relu=False, splits=1,
         a_map=(INT_MAX, 0, 0), b_map=(INT_MAX, 0, 0), a_colsum=None, tag="gemm", batch=1, strides=(0, 0, 0, 0, 0),
         atomic=False, splitk_ws: Optional[torch.Tensor] = None, sum_batch=False):
    """C[M,N] (+)= opA * opB (+bias)(ReLU); A/B/Cm are ints (device addresses).  ``batch`` problems of one shape
    share a launch; ``strides`` = element strides of (A, B, C, bias, a_colsum) between them.  With ``splitk_ws`` (a float
    scratch tensor) the K slices are left there and folded into C by a second, tiny launch instead of float atomics."""
    if splitk_ws is not None and splits > 1 and accumulate and not relu and bias is None:
        need = L.gemm_splitk_ws_floats(M, N, K, splits, batch)
        if splitk_ws.numel() < need:
            raise RuntimeError(f"split-K workspace too small: {splitk_ws.numel()} < {need}")
        L.call("ss_gemm_f32_batched", int(a_kc), int(b_kc), M, N, K, A, lda, a_map[0], a_map[1], a_map[2], B, ldb,
               b_map[0], b_map[1], b_map[2], splitk_ws.data_ptr(), N, None, a_colsum, 8, splits, batch,
               strides[0], strides[1], 0, 0, strides[4], L.stream(), tag=tag)
        L.call("ss_gemm_splitk_reduce", splitk_ws.data_ptr(), M, N, K, splits, batch, Cm, ldc, strides[2], L.stream(),
               tag=tag + "_reduce")
        return
    flags = (1 if accumulate else 0) | (2 if relu else 0) | (4 if atomic else 0) | (16 if sum_batch else 0)
    L.call("ss_gemm_f32_batched", int(a_kc), int(b_kc), M, N, K, A, lda, a_map[0], a_map[1], a_map[2], B, ldb,
           b_map[0], b_map[1], b_map[2], Cm, ldc, bias, a_colsum, flags, splits, batch, *strides, L.stream(), tag=tag)


_SPLITK_TARGET = int(os.environ.get("SS_SPLITK_TARGET", "768"))
# Layer 0's weight-gradient group runs when no recurrence is left to share the chip with: wide 192 x 192 tiles, one workgroup per
# CU (ss_gemm_f32_splitk_group flags bit 0).  The upper layers' groups run beside the BPTT kernel of the layer below and keep the
# 128 x 64 form: measured in the step, the wide form there made the launch 12 % shorter and the step 1 % longer.
DW_WIDE_ALONE = int(os.environ.get("SS_DW_WIDE_ALONE", "1"))
# The fused ROI-CNN kernels walk only the frames inside their clips (rows t >= lengths[b] of a padded batch never reach the packed
# recurrence and carry no gradient): a batch that is 40 % padding costs 40 % less CNN time.  0 = every frame, as the reference does.
SKIP_PADDED_FRAMES = os.environ.get("SS_CNN_SKIP_PADDING", "1") != "0"


_DX_SPLIT_CAP = int(os.environ.get("SS_DX_SPLITS", "2"))  # K slices of a d layer_in GEMM with few output tiles (1 = off)


def split_k(M, N, K, batch=1, target_wgs=_SPLITK_TARGET):
    """K slices for a weight-gradient GEMM (tiny M x N, huge K): enough workgroups to fill the chip ~3x over."""
    tiles = -(-M // 128) * -(-N // 64) * batch
    return max(1, min(K // 128, target_wgs // tiles))


def dw_shapes(cfg, B, T, l):
    """(M, N, K, K slices) of the weight-gradient GEMMs of GRU layer ``l``: d W_ih, then the r|z and the n rows of d W_hh."""
    H, N = cfg.hidden, B * T
    K = cfg.in_dim if l == 0 else 2 * H
    out = [(3 * H, K, N, split_k(3 * H, K, N, 2))]
    if T > 1:
        Kh = B * (T - 1)
        out += [(2 * H, H, Kh, split_k(2 * H, H, Kh, 2, _HH_TARGET)), (H, H, Kh, split_k(H, H, Kh, 2, _HH_TARGET))]
    return out


def dw_problems(ws, G, cfg, l, lin, ld_in):
    """The weight-gradient GEMMs of GRU layer ``l`` as ss_gemm_problem records (one grouped launch, both directions each):
    d W_ih = dGi^T . layer_in;  d W_hh = dGh^T . h_prev in two pieces (rows r|z from columns [0, 2H) of dG, rows n from
    columns [3H, 4H) = d(W_hn h + b_hn)).  Rows (b,t) of dG pair with out rows (b,t-1) (forward) / (b,t+1) (reverse): with the
    row remap (group T-1 of stride T, A offset 1, B offset 0) the reverse direction is the same pairing seen from one row
    earlier in dG and one row later in out, i.e. two pointer shifts."""
    B, T, H = ws.B, ws.T, cfg.hidden
    N, K = B * T, (cfg.in_dim if l == 0 else 2 * H)
    dg = ws.dG[l].data_ptr()
    wi, wir = f"gru.weight_ih_l{l}", f"gru.weight_ih_l{l}_reverse"
    wh, whr = f"gru.weight_hh_l{l}", f"gru.weight_hh_l{l}_reverse"
    ident = (INT_MAX, 0, 0)

    def prob(M, Nn, Kk, A, B_, Cm, ldc, splits, a_map, b_map, sa, sb, sc):
        return L.GemmProblem(0, 0, M, Nn, Kk, A, 4 * H, a_map[0], a_map[1], a_map[2], B_, ld_in if b_map is ident else 2 * H,
                             b_map[0], b_map[1], b_map[2], Cm, ldc, splits, 2, sa, sb, sc)

    shapes = dw_shapes(cfg, B, T, l)
    out = [prob(*shapes[0][:3], dg, lin, G[wi].data_ptr(), K, shapes[0][3], ident, ident, N * 4 * H, 0, _pstride(G, wi, wir))]
    if T > 1:
        am, bm = (T - 1, T, 1), (T - 1, T, 0)
        sa, sb, sc = N * 4 * H - 4 * H, H + 2 * H, _pstride(G, wh, whr)
        gw, hp = G[wh], ws.out[l].data_ptr()
        out.append(prob(*shapes[1][:3], dg, hp, gw.data_ptr(), H, shapes[1][3], am, bm, sa, sb, sc))
        out.append(prob(*shapes[2][:3], dg + 3 * H * 4, hp, _addr(gw, 2 * H * H), H, shapes[2][3], am, bm, sa, sb, sc))
    return out


def _pstride(P, a: str, b: str) -> int:
    """Element stride between two tensors of the flat parameter (or gradient) bucket."""
    return (P[b].data_ptr() - P[a].data_ptr()) // 4


def _addr(t: torch.Tensor, offset_elems: int = 0) -> int:
    return t.data_ptr() + offset_elems * t.element_size()


def zero_buffers(bufs) -> None:
    """Clear float tensors on the current stream, two per launch (ss_zero_f32x2): no library fill on the path."""
    bufs = [b for b in bufs if b is not None and b.numel()]
    for k in range(0, len(bufs), 2):
        a, b = bufs[k], bufs[k + 1] if k + 1 < len(bufs) else None
        L.call("ss_zero_f32x2", a.data_ptr(), a.numel(), L.ptr(b), b.numel() if b is not None else 0, L.stream())


def check_gru_sync(ws) -> None:
    """Raise if a bounded wait of the multi-CU GRU recurrence ever gave up on this workspace (a partner workgroup was not
    co-resident: the kernels then poison their result with NaN and count the event in word 2 of the sync header).  Reads
    one int from the device: call it where the host synchronises anyway (end of an epoch, ``evaluate``)."""
    sync = getattr(ws, "gru_sync", None)
    if sync is not None and int(sync[2]) != 0:
        raise RuntimeError(f"multi-CU GRU recurrence: {int(sync[2])} exchange waits timed out (a partner workgroup never "
                           "arrived); results since then are poisoned with NaN")


_SIDE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def side_stream(device, slot=0) -> "torch.cuda.Stream":
    """The side stream of a training workspace: ONE per device and micro-batch slot, shared by every workspace of the process.
    A stream per workspace looked harmless until a process had built a handful of them (bench.py's default run: config 2, config 5,
    two shipped shapes, ...): HIP multiplexes streams onto a few hardware queues, the sixth workspace's side stream landed on the
    queue of the caller's stream, its weight-gradient GEMMs queued behind the critical path instead of running beside it, and the
    same step took 1.705 instead of 1.555 ms (gpurun_out/r4_b15.log against r4_b14_*.log).  Workspaces are used one after the
    other, so sharing costs nothing; slots that ARE in flight together (Trainer(micro_batches=M)) keep one stream each.
    Lowest priority: it must never delay the dispatch of the critical-path kernels on the caller's stream."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), slot if isinstance(slot, int) else 0)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=torch.cuda.Stream.priority_range()[0])
    return st


def make_workspace(cfg: Config, B: int, T: int, roi_hw, device, train: bool, slot=0):
    if cfg.precision == "bf16":
        from .engine_bf16 import WorkspaceBf16

        return WorkspaceBf16(cfg, B, T, roi_hw, device, train, slot)
    return Workspace(cfg, B, T, roi_hw, device, train, slot)


class Workspace:
    """Activation / gradient buffers for one (B, T, H, W) shape; reused across steps (hipGraph friendly)."""

    bf16 = False

    def __init__(self, cfg: Config, B: int, T: int, roi_hw, device, train: bool, slot=0):
        self.cfg, self.B, self.T, self.roi_hw, self.train = cfg, B, T, roi_hw, train
        self.stash_gen, self.stash_live = 0, False  # model._Fn: which autograd node the stashed activations belong to
        N, H = B * T, cfg.hidden
        f32 = dict(device=device, dtype=torch.float32)
        u8 = dict(device=device, dtype=torch.uint8)
        self.lengths = torch.empty(B, device=device, dtype=torch.int32)
        # step counters of the multi-CU recurrence (zeroed once; the kernels keep them consistent): only shapes small
        # enough to leave most of the chip idle under the one-CU-per-slice kernels get one
        nb = L.gru_sync_bytes(B, T, H) if USE_SPLIT_GRU else 0
        self.gru_sync = torch.zeros(nb // 4, device=device, dtype=torch.int32) if nb else None
        # weight-gradient GEMMs run on a side stream next to the (32-CU) recurrence of the layer below
        self.side = side_stream(device, slot) if train else None
        self.ev_fork = torch.cuda.Event() if train else None
        self.ev_join = torch.cuda.Event() if train else None
        self.ev_cnn_fwd = torch.cuda.Event() if train else None  # recorded after the ROI-CNN forward (micro-batch stagger)
        self.stagger = False  # set by the trainer when another micro-batch waits for ev_cnn_fwd
        # roi_hw None with use_roi: a workspace for forward(..., z_ready=True) -- rows that already hold the embeddings (serving.py)
        self.Z = torch.empty(N, cfg.in_dim, **f32) if cfg.use_roi and roi_hw is not None else None
        # ROI sizes the fused, LDS-resident CNN kernels are not built for run layer by layer (cnn_generic.py)
        self.cnn_generic = None
        if cfg.use_roi and roi_hw is not None:
            from . import cnn_generic

            if not cnn_generic.fused_supported(*roi_hw):
                self.cnn_generic = cnn_generic.GenericCnn(N, roi_hw[0], roi_hw[1], device, train)
        self.gi = [torch.empty(2, N, 3 * H, **f32) for _ in range(cfg.gru_layers)]
        self.out = [torch.empty(N, 2 * H, **f32) for _ in range(cfg.gru_layers)]
        self.attn = torch.empty(B, T, **f32)
        self.pooled = torch.empty(B, 2 * H, **f32)
        self.ln = torch.empty(B, 2 * H, **f32)
        self.mid = torch.empty(B, cfg.head_mid, **f32)
        self.logits = torch.empty(B, cfg.num_classes, **f32)
        self.out_drop = [torch.empty(N, 2 * H, **f32) for _ in range(cfg.gru_layers - 1)]
        self.mid_drop = torch.empty(B, cfg.head_mid, **f32)
        if train:
            self.save = [torch.empty(2, N, 4, H, **f32) for _ in range(cfg.gru_layers)]
            self.dG = [torch.empty(2, N, 4, H, **f32) for _ in range(cfg.gru_layers)]
            self.d_out = torch.empty(N, 2 * H, **f32)
            # gradient w.r.t. the output of layer l-1 (destination of layer l's d layer_in GEMM, l >= 1): zeroed on the side
            # stream while the top of the backward pass runs, summed into with atomics by both directions
            # scratch for the K slices of one layer's weight-gradient GEMMs (the layers follow each other on the side stream)
            need = 0
            for l in range(cfg.gru_layers):
                probs = [L.GemmProblem(0, 0, m_, n_, k_, None, 0, 1, 0, 0, None, 0, 1, 0, 0, None, 0, sp_, 2, 0, 0, 0)
                         for (m_, n_, k_, sp_) in dw_shapes(cfg, B, T, l)]
                need = max(need, L.gemm_group_ws_floats(probs))
            self.splitk_ws = torch.empty(need, **f32)
            self.d_lower = [None] + [torch.empty(N, 2 * H, **f32) for _ in range(1, cfg.gru_layers)]
            self.ev_zero = torch.cuda.Event()
            self.xhat = torch.empty(B, 2 * H, **f32)
            self.rstd = torch.empty(B, **f32)
            self.d_logits = torch.empty(B, cfg.num_classes, **f32)
            self.d_mid = torch.empty(B, cfg.head_mid, **f32)
            self.tail_part = torch.empty(B, 3, 2 * H, **f32)  # per-clip terms of d gamma | d beta | d w_score (tail_bwd)
            self.d_ln = torch.empty(B, 2 * H, **f32)
            self.d_pooled = torch.empty(B, 2 * H, **f32)
            self.dZ = torch.empty(N, cfg.in_dim, **f32) if cfg.use_roi else None
            if cfg.use_roi and self.cnn_generic is None:
                Hh, Ww = roi_hw
                # every size comes from the library (the kernels' own LDS images, kept as they are) and goes back to it with
                # each launch, where the forward and the backward object compare it with the layout they were compiled with
                self.cnn_sizes = L.cnn_stash_sizes(Hh, Ww)
                n_a1, n_a2, n_i1, n_i2, n_m3, n_feat = self.cnn_sizes
                self.st_a1 = torch.empty(N, n_a1, **f32)
                self.st_i1 = torch.empty(N, n_i1, **u8)
                self.st_a2 = torch.empty(N, n_a2, **f32)
                self.st_i2 = torch.empty(N, n_i2, **u8)     # (H/4, W/4, 16)
                self.st_m3 = torch.empty(N, n_m3, **u8)     # (H/4 * W/4, 32)
                self.st_feat = torch.empty(N, n_feat, **f32)  # 24 features, 24 counts, mean, std, pad
        # [0] = how many rows (b, t) lie inside their clip, [1 ...] = those rows: the fused CNN kernels walk only them
        self.frames = (torch.empty(1 + N, device=device, dtype=torch.int32)
                       if cfg.use_roi and roi_hw is not None and self.cnn_generic is None and SKIP_PADDED_FRAMES else None)
        # The backward kernel pays 1 % for the list (roi_cnn_bwd.hip: 48 instead of 38 spilled scalar registers), so a training
        # workspace keeps the count of the last batch it has SEEN FINISH in pinned memory -- an 8-byte copy on the side stream,
        # never waited for -- and a step that follows a batch of full clips launches the kernels without the list.  A wrong guess
        # walks the padding frames as the reference does: slower, never different.  walk_listed = this step's choice (the
        # forward makes it, the backward follows: the stash slots of frames the forward skipped are not there to read).
        self.frames_seen = (torch.full((1,), -1, dtype=torch.int32).pin_memory()
                            if self.frames is not None and train else None)
        self.walk_listed = self.frames is not None


def forward(P: Dict[str, torch.Tensor], cfg: Config, ws: Workspace, X: torch.Tensor, R: Optional[torch.Tensor], *,
            train: bool, stash: bool = False, seed: int = 0, ce=None, x_in_place: bool = False, z_ready: bool = False) -> torch.Tensor:
    """Runs the forward kernels; returns ws.logits (B,C).  ``ws.lengths`` must already hold the int32 lengths.
    ``z_ready``: X (B,T,in_dim) already holds torch.cat((landmark features, ROI embeddings)) of train_model_official.py:297 -- the
    embeddings were made when the frames arrived (sliding-window serving) -- and the ROI branch is skipped (inference only).
    ``train`` turns the two dropouts on (p from cfg); ``stash`` keeps what ``backward`` needs (needs a
    Workspace built with train=True).  ``ce = (y_ptr, label_smoothing, denom, loss_ptr, correct_ptr)`` makes the
    fused tail kernel also evaluate the loss and leave d(loss)/d(logits) in ``ws.d_logits``."""
    if ws.bf16:
        from . import engine_bf16

        return engine_bf16.forward(P, cfg, ws, X, R, train=train, stash=stash, seed=seed, ce=ce, x_in_place=x_in_place)
    B, T, H, N = ws.B, ws.T, cfg.hidden, ws.B * ws.T
    s = L.stream()
    if stash and not ws.train:
        raise RuntimeError("stash=True needs a training workspace")
    # ---- ROI branch: normalise + CNN -> columns [x_dim, x_dim+E) of Z; X -> columns [0, x_dim)
    if z_ready:
        if stash or train or not cfg.use_roi or X.shape[2] != cfg.in_dim:
            raise RuntimeError("z_ready: inference on rows of width in_dim = x_dim + roi_emb of a use_roi model")
        layer_in, ld_in = X.data_ptr(), cfg.in_dim
    elif cfg.use_roi:
        Hh, Ww = ws.roi_hw
        ws_Z = ws.Z
        if not x_in_place:  # (the trainer's prologue kernel has already put X there)
            L.call("ss_copy_rows_f32", X.data_ptr(), cfg.x_dim, ws_Z.data_ptr(), cfg.in_dim, N, cfg.x_dim, s)
        if ws.cnn_generic is not None:  # an ROI size outside the fused kernels' set: layer by layer (cnn_generic.py)
            ws.cnn_generic.forward(P, R, cfg.roi_standardize, cfg.roi_emb, _addr(ws_Z, cfg.x_dim), cfg.in_dim, stash)
        else:
            cw = [P[k].data_ptr() for k in ("roi_cnn.net.0.weight", "roi_cnn.net.0.bias", "roi_cnn.net.3.weight",
                                            "roi_cnn.net.3.bias", "roi_cnn.net.6.weight", "roi_cnn.net.6.bias",
                                            "roi_cnn.fc.weight", "roi_cnn.fc.bias")]
            st = ([ws.st_a1.data_ptr(), ws.st_i1.data_ptr(), ws.st_a2.data_ptr(), ws.st_i2.data_ptr(),
                   ws.st_m3.data_ptr(), ws.st_feat.data_ptr()] if stash else [None] * 6)
            if ws.frames is not None and not x_in_place:  # (the trainer's prologue kernel has listed them already)
                L.call("ss_roi_active_frames", ws.lengths.data_ptr(), B, T, ws.frames.data_ptr(), _addr(ws_Z, cfg.x_dim), cfg.in_dim,
                       cfg.roi_emb, s)
            ws.walk_listed = ws.frames is not None and not (stash and int(ws.frames_seen[0]) == N)
            L.call("ss_roi_cnn_fwd_frames", R.data_ptr(), N, Hh, Ww, int(cfg.roi_standardize), *cw, cfg.roi_emb,
                   _addr(ws_Z, cfg.x_dim), cfg.in_dim, *st, ws.cnn_sizes.ptr if stash else None,
                   L.ptr(ws.frames) if ws.walk_listed else None, s,
                   tag="ss_roi_cnn_fwd_stash")  # (the timing tag names the kernel, which is the same with or without a list)
        if ws.train and ws.stagger:  # an event record is a barrier packet on this stream (~6 us): only when somebody waits for it
            ws.ev_cnn_fwd.record()
        layer_in, ld_in = ws_Z.data_ptr(), cfg.in_dim
    else:
        layer_in, ld_in = X.data_ptr(), cfg.x_dim
    # ---- GRU layers
    for l in range(cfg.gru_layers):
        K = cfg.in_dim if l == 0 else 2 * H
        wf, wr = f"gru.weight_ih_l{l}", f"gru.weight_ih_l{l}_reverse"
        gemm(1, 1, N, 3 * H, K, layer_in, ld_in, P[wf].data_ptr(), K, ws.gi[l].data_ptr(), 3 * H,
             bias=P[f"gru.bias_ih_l{l}"].data_ptr(), tag="gemm_gru_ih", batch=2,
             strides=(0, _pstride(P, wf, wr), N * 3 * H, _pstride(P, f"gru.bias_ih_l{l}", f"gru.bias_ih_l{l}_reverse"), 0))
        drop = train and l < cfg.gru_layers - 1 and cfg.gru_dropout > 0.0
        # nn.GRU's inter-layer dropout: a by-product of the multi-CU recurrence kernel (its own launch + read of `out` otherwise)
        fused_drop = drop and ws.gru_sync is not None and FUSE_GRU_DROPOUT
        L.call("ss_gru_fwd_drop", ws.gi[l].data_ptr(), P[f"gru.weight_hh_l{l}"].data_ptr(),
               P[f"gru.weight_hh_l{l}_reverse"].data_ptr(), P[f"gru.bias_hh_l{l}"].data_ptr(),
               P[f"gru.bias_hh_l{l}_reverse"].data_ptr(), ws.lengths.data_ptr(), B, T, H, ws.out[l].data_ptr(),
               ws.save[l].data_ptr() if stash else None, ws.out_drop[l].data_ptr() if fused_drop else None,
               cfg.gru_dropout if fused_drop else 0.0, seed, (l + 1) << 40, L.ptr(ws.gru_sync), s, tag="ss_gru_fwd")
        layer_in, ld_in = ws.out[l].data_ptr(), 2 * H
        if drop:
            if not fused_drop:
                L.call("ss_dropout", ws.out[l].data_ptr(), ws.out_drop[l].data_ptr(), N * 2 * H, cfg.gru_dropout, seed,
                       (l + 1) << 40, None, s)
            layer_in = ws.out_drop[l].data_ptr()
    top = ws.out[cfg.gru_layers - 1]
    # ---- AttnPool + head (+ loss): one fused launch, a workgroup per clip
    p_drop = cfg.head_dropout if train else 0.0
    y_ptr, ls, denom, loss_ptr, correct_ptr = ce if ce is not None else (None, 0.0, 1.0, None, None)
    L.call("ss_tail_fwd", top.data_ptr(), ws.lengths.data_ptr(), P["pool.score.weight"].data_ptr(),
           P["pool.score.bias"].data_ptr(), P["head.0.weight"].data_ptr(), P["head.0.bias"].data_ptr(),
           P["head.1.weight"].data_ptr(), P["head.1.bias"].data_ptr(), P["head.4.weight"].data_ptr(),
           P["head.4.bias"].data_ptr(), y_ptr, B, T, 2 * H, cfg.head_mid, cfg.num_classes, cfg.ln_eps, p_drop, seed,
           7 << 40, ls, denom, ws.attn.data_ptr() if stash else None, ws.xhat.data_ptr() if stash else None,
           ws.rstd.data_ptr() if stash else None, ws.ln.data_ptr() if stash else None,
           ws.mid.data_ptr() if stash else None, ws.mid_drop.data_ptr() if stash else None, ws.logits.data_ptr(),
           ws.d_logits.data_ptr() if ce is not None else None, loss_ptr, correct_ptr, s)
    return ws.logits


def backward(P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor], cfg: Config, ws: Workspace, X: torch.Tensor,
             R: Optional[torch.Tensor], d_logits: torch.Tensor, *, train: bool, seed: int = 0,
             d_X: Optional[torch.Tensor] = None) -> None:
    """Accumulates d(loss)/d(param) into ``G`` (same keys as ``P``) given d(loss)/d(logits).
    Must follow a ``forward(..., train=<same>, seed=<same>)`` on the same workspace."""
    if ws.bf16:
        from . import engine_bf16

        return engine_bf16.backward(P, G, cfg, ws, X, R, d_logits, train=train, seed=seed, d_X=d_X)
    B, T, H, N = ws.B, ws.T, cfg.hidden, ws.B * ws.T
    s = L.stream()
    C, MID = cfg.num_classes, cfg.head_mid
    p_drop = cfg.head_dropout if train else 0.0
    top = ws.out[cfg.gru_layers - 1]
    # ---- tail: one fused launch down to d_out of the top GRU layer ...
    L.call("ss_tail_bwd", top.data_ptr(), ws.lengths.data_ptr(), P["pool.score.weight"].data_ptr(),
           P["head.0.weight"].data_ptr(), P["head.1.weight"].data_ptr(), P["head.4.weight"].data_ptr(),
           ws.attn.data_ptr(), ws.xhat.data_ptr(), ws.rstd.data_ptr(), ws.mid.data_ptr(), d_logits.data_ptr(), B, T,
           2 * H, MID, C, p_drop, seed, 7 << 40, ws.d_mid.data_ptr(), ws.d_out.data_ptr(),
           G["head.0.weight"].data_ptr(), G["head.0.bias"].data_ptr(), G["pool.score.weight"].data_ptr(),
           G["pool.score.bias"].data_ptr(), ws.tail_part.data_ptr(), s)
    # ... while the two Linear weight gradients (batched over the clips) go to the side stream.  The fork event is recorded here,
    # the side stream's launches are ENQUEUED behind the top layer's BPTT launch (head_side_work, called in the loop below): the
    # host needs ~30 us for these seven launches and the main queue sat empty meanwhile (15 us between tail_bwd and the BPTT
    # kernel in the kernel trace, tools/step_gaps.py; 6 us of it is the event's barrier packet and stays)
    side = ws.side if USE_SIDE_STREAM else torch.cuda.current_stream()
    ws.ev_fork.record()

    def head_side_work():
        with torch.cuda.stream(side):
            side.wait_event(ws.ev_fork)
            # the atomically summed destinations of the d layer_in GEMMs: cleared here, off the critical path, and first --
            # the top layer's d layer_in GEMM waits for them
            zero_buffers(ws.d_lower[1:] + ([ws.dZ] if cfg.use_roi else []))
            ws.ev_zero.record()
            if ws.frames_seen is not None:  # how many frames of this batch lay inside a clip: read by a later step's forward
                ws.frames_seen.copy_(ws.frames[:1], non_blocking=True)
            # LayerNorm gamma / beta and score-weight gradients: column sums of the rows the tail kernel left per clip
            for k_, name_ in enumerate(("head.0.weight", "head.0.bias", "pool.score.weight")):
                L.call("ss_colsum_f32", _addr(ws.tail_part, k_ * 2 * H), B, 2 * H, 3 * 2 * H, G[name_].data_ptr(), L.stream())
            gemm(0, 0, C, MID, B, d_logits.data_ptr(), C, ws.mid_drop.data_ptr(), MID, G["head.4.weight"].data_ptr(), MID,
                 accumulate=True, atomic=True, a_colsum=G["head.4.bias"].data_ptr())
            gemm(0, 0, MID, 2 * H, B, ws.d_mid.data_ptr(), MID, ws.ln.data_ptr(), 2 * H, G["head.1.weight"].data_ptr(),
                 2 * H, accumulate=True, atomic=True, a_colsum=G["head.1.bias"].data_ptr())

    # ---- GRU layers, top down
    use_drop = train and cfg.gru_dropout > 0.0
    zero_waited = False
    for l in range(cfg.gru_layers - 1, -1, -1):
        K = cfg.in_dim if l == 0 else 2 * H
        # layers below the top read the gradient w.r.t. their DROPPED-OUT output; the kernel re-draws the mask
        top_layer = l == cfg.gru_layers - 1
        g_in = ws.d_out if top_layer else ws.d_lower[l + 1]
        L.call("ss_gru_bwd", g_in.data_ptr(), ws.out[l].data_ptr(), ws.save[l].data_ptr(),
               P[f"gru.weight_hh_l{l}"].data_ptr(), P[f"gru.weight_hh_l{l}_reverse"].data_ptr(),
               ws.lengths.data_ptr(), B, T, H, ws.dG[l].data_ptr(),
               0.0 if (top_layer or not use_drop) else cfg.gru_dropout, seed, (l + 1) << 40,
               G[f"gru.bias_ih_l{l}"].data_ptr(), G[f"gru.bias_hh_l{l}"].data_ptr(),
               G[f"gru.bias_ih_l{l}_reverse"].data_ptr(), G[f"gru.bias_hh_l{l}_reverse"].data_ptr(), L.ptr(ws.gru_sync), s)
        if top_layer:
            head_side_work()
        if l == 0:
            if cfg.use_roi:
                lin, ld_in = ws.Z.data_ptr(), cfg.in_dim
            else:
                lin, ld_in = X.data_ptr(), cfg.x_dim
        else:
            use_drop = train and cfg.gru_dropout > 0.0
            lin, ld_in = (ws.out_drop[l - 1] if use_drop else ws.out[l - 1]).data_ptr(), 2 * H
        def side_work(l=l, K=K, lin=lin, ld_in=ld_in):
            # fork: everything that only feeds the parameter gradients of this layer goes to the side stream
            side = ws.side if USE_SIDE_STREAM else torch.cuda.current_stream()
            ws.ev_fork.record()
            with torch.cuda.stream(side):
                side.wait_event(ws.ev_fork)
                L.call("ss_gemm_f32_splitk_group", *L.gemm_group(dw_problems(ws, G, cfg, l, lin, ld_in)), ws.splitk_ws.data_ptr(),
                       ws.splitk_ws.numel(), DW_WIDE_ALONE if l == 0 else 0, L.stream(), tag="gemm_gru_dW")
        # the weight-gradient GEMMs of the upper layers start only when this layer's d layer_in GEMM is through: two
        # MFMA-bound GEMMs side by side gain nothing and the one on the critical path loses half its rate; beside the
        # latency-bound recurrence of the layer below they fill idle matrix pipes.  Layer 0 has no recurrence left to
        # hide behind, so its weight gradients run beside its (short) d layer_in GEMM.
        if not (SIDE_AFTER_DX and l > 0):
            side_work()
        # d layer_in = dGi_f . W_ih_f + dGi_r . W_ih_r
        need_dx = (l > 0) or cfg.use_roi or (d_X is not None)
        if need_dx:
            # both directions in ONE launch (twice the workgroups: the N=116 case alone leaves half the CUs idle),
            # summed with float atomics into a destination that was zeroed on the side stream.  (Measured alternative:
            # sum_batch=True -- one workgroup per tile running its K loop through both directions, plain stores, nothing
            # to clear -- is 0.8 % slower on the step: half the workgroups, and layer 0 has only 60 tiles.)
            if l > 0:
                dst, ld_dst = ws.d_lower[l].data_ptr(), 2 * H
            elif cfg.use_roi:
                dst, ld_dst = ws.dZ.data_ptr(), cfg.in_dim
            else:
                dst, ld_dst = d_X.data_ptr(), cfg.x_dim
                zero_buffers([d_X])
            if not zero_waited:  # once per backward pass: every cleared buffer is behind the same event
                torch.cuda.current_stream().wait_event(ws.ev_zero)
                zero_waited = True
            wi, wir = f"gru.weight_ih_l{l}", f"gru.weight_ih_l{l}_reverse"
            # nobody asked for d X: only the ROI-embedding columns of d Z feed the CNN backward
            c0 = cfg.x_dim if (l == 0 and cfg.use_roi and d_X is None) else 0
            # few output tiles (layer 0 with only the ROI columns wanted: 120 workgroups for 256 CUs): slice K as well --
            # the destination is summed atomically anyway
            tiles = -(-N // 128) * -(-(K - c0) // 64) * 2
            dx_splits = max(1, min(_DX_SPLIT_CAP, 768 // tiles, (3 * H) // 96))
            gemm(1, 0, N, K - c0, 3 * H, ws.dG[l].data_ptr(), 4 * H, _addr(P[wi], c0), K, dst + 4 * c0, ld_dst,
                 accumulate=True, atomic=True, tag="gemm_gru_dX", batch=2, splits=dx_splits,
                 strides=(N * 4 * H, _pstride(P, wi, wir), 0, 0, 0))
        if SIDE_AFTER_DX and l > 0:
            side_work()
    # ---- ROI CNN
    side_joined = False  # nothing is queued on the side stream after the join in front of the CNN backward
    if cfg.use_roi:
        Hh, Ww = ws.roi_hw
        if d_X is not None:
            L.call("ss_copy_rows_f32", ws.dZ.data_ptr(), cfg.in_dim, d_X.data_ptr(), cfg.x_dim, N, cfg.x_dim, s)
        if USE_SIDE_STREAM and JOIN_BEFORE_CNN_BWD:
            # the persistent ROI-CNN kernel takes every CU for ~1 ms: weight-gradient GEMMs still queued on the side
            # stream at that point would only finish after it (measured: a 180 us tail), so let them drain first --
            # they run beside the d layer_in GEMM above and cost the critical path a few tens of microseconds
            ws.ev_join.record(ws.side)
            torch.cuda.current_stream().wait_event(ws.ev_join)
            side_joined = True
        names = ("roi_cnn.net.0.weight", "roi_cnn.net.0.bias", "roi_cnn.net.3.weight", "roi_cnn.net.3.bias",
                 "roi_cnn.net.6.weight", "roi_cnn.net.6.bias", "roi_cnn.fc.weight", "roi_cnn.fc.bias")
        if ws.cnn_generic is not None:
            ws.cnn_generic.backward(P, G, cfg.roi_emb, _addr(ws.dZ, cfg.x_dim), cfg.in_dim)
        else:
            L.call("ss_roi_cnn_bwd_frames", R.data_ptr(), N, Hh, Ww, int(cfg.roi_standardize), *[P[k].data_ptr() for k in names],
                   cfg.roi_emb, ws.st_a1.data_ptr(), ws.st_i1.data_ptr(), ws.st_a2.data_ptr(), ws.st_i2.data_ptr(),
                   ws.st_m3.data_ptr(), ws.st_feat.data_ptr(), ws.cnn_sizes.ptr, _addr(ws.dZ, cfg.x_dim), cfg.in_dim,
                   *[G[k].data_ptr() for k in names], L.ptr(ws.frames) if ws.walk_listed else None, s, tag="ss_roi_cnn_bwd")
    # join the side stream: the caller's next kernels (all-reduce, clip, Adam) read every gradient
    if USE_SIDE_STREAM and not side_joined:
        ws.ev_join.record(ws.side)
        torch.cuda.current_stream().wait_event(ws.ev_join)
