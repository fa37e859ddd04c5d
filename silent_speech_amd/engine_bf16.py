"""Kernel orchestration of the bf16-MFMA path (BASELINE config 5: 100 words, 96x96 ROI, CNN 16/32/64/96, BiGRU H = 512).

Same contract as ``engine.py`` (explicit forward / backward over raw device buffers, no aten compute op, no CPU
fallback); what differs is the kernel set:

    ROI CNN      ss_c5_conv12_fwd (conv1 + conv2 fused) -> ss_c5_conv_fwd(3) -> ss_c5_conv_last_fwd   (cnn_bf16.hip)
                 pooled maps a2, a3 between the layers: NHWC bf16 in HBM + one argmax byte per element (the training stash);
                 the pooled conv1 map stays in LDS and is recomputed from the frame by the backward kernels that need it
    GRU layers   ss_gemm_bf16_batched (input projections, d layer_in, weight gradients) + ss_gru_bf16_fwd / _bwd
                 (one launch per time step, both directions; W_hh as bf16 copies made once per step by ss_gru_bf16_prep)
    tail         the f32 fused AttnPool / head / CE kernels of the f32 path (0.1 % of the FLOPs)

f32 master weights, f32 gradients, f32 Adam: only MFMA operands are bf16.  The reference defines no such model
(SURVEY.md 8d row 5); the module is train_model_official.py:209-310 with wider layers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib as L

INT_MAX = 2**31 - 1
USE_SIDE_STREAM = True  # bench.py turns it off for its per-kernel timing pass (HIP-event pairs need one stream)
FUSE_DGRAD2_WGRAD1 = os.environ.get("SS_C5_UNFUSED_DA1", "0") != "1"  # 0: conv2 dgrad and conv1 wgrad as two kernels (round-2 form)
USE_PERSISTENT_GRU = os.environ.get("SS_C5_STEP_GRU", "0") != "1"  # 0: one launch per time step (the round-2 form, kept as the fallback)
CNN_CHANNELS = (16, 32, 64, 96)
ROI_HW = (96, 96)
_IDENT = (INT_MAX, 0, 0)


def gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, Cm, ldc, bias=None, accumulate=False, atomic=False, splits=1, a_map=_IDENT,
         b_map=_IDENT, batch=1, strides=(0, 0, 0, 0), tag="gemm_bf16", src16=True):
    """``src16``: A and B are bf16 in HBM (the copies the recurrence kernels / ss_gru_bf16_prep / ss_cvt_bf16_rows leave);
    leading dimensions and strides then count bf16 elements."""
    flags = (1 if accumulate else 0) | (4 if atomic else 0) | (8 if src16 else 0)
    L.call("ss_gemm_bf16_batched", int(a_kc), int(b_kc), M, N, K, A, lda, *a_map, B, ldb, *b_map, Cm, ldc, bias, flags, splits,
           batch, *strides, L.stream(), tag=tag)


# K slices of the weight-gradient GEMMs aim at this many workgroups (measured 192 / 384 / 768 / 1536: 6.10 / 6.09 / 6.17 / 6.33 ms
# per step: the GEMMs run beside the BPTT steps on the side stream, fewer float atomics matter more than their own time)
_SPLITK_TARGET = int(os.environ.get("SS_C5_SPLITK_TARGET", "384"))


def split_k(M, N, K, batch, target_wgs=None):
    target_wgs = target_wgs or _SPLITK_TARGET
    tiles = -(-M // 128) * -(-N // 128) * batch
    return max(1, min(-(-K // 64), target_wgs // tiles))


# The weight-gradient GEMMs of a GRU layer as ONE launch of the LDS-DMA ring kernel, K dealt evenly over the CUs, + one reduce
# launch (ss_gemm_bf16_splitk_group) when every K is a whole number of 64-deep tiles (B a multiple of 64); otherwise three
# launches with K slices and float atomics.  SS_C5_DW_GROUP=0: always the latter (the form of the first half of round 3).
# conv1's pool winners (37 KB per frame) stashed by the forward kernel for the fused conv2-dgrad / conv1-wgrad kernel (0: recomputed there)
STASH_I1 = os.environ.get("SS_C5_STASH_I1", "1") != "0"
# Linear(96 -> roi_emb) behind the CNN and its gradients as GEMMs over all frames instead of per-frame loops inside the last layer's
# persistent kernels (0: inside the kernels, the form of the first half of round 3)
FC_AS_GEMM = os.environ.get("SS_C5_FC_GEMM", "1") != "0"
USE_DW_GROUP = os.environ.get("SS_C5_DW_GROUP", "1") != "0"
DW_ALL_LAYERS = os.environ.get("SS_C5_DW_ALL_LAYERS", "1") != "0"  # 0: one grouped launch per layer
# d layer_in = dGi_f W_ih_f + dGi_r W_ih_r as one product with K concatenated (plain stores, no cleared destination, no atomics)
USE_DX_KCAT = os.environ.get("SS_C5_DX_KCAT", "1") != "0"


def dw_problems(cfg, B, T, l, Kp, dg=16, lin=16, hp=16, g_ih=(16, 0), g_hh=(16, 0)):
    """ss_gemm_problem records of layer ``l``'s weight gradients (both directions each; bf16 operands): d W_ih = dGi^T . layer_in;
    d W_hh = dGh^T . h_prev in two pieces (rows r|z from columns [0, 2H) of dG, rows n from columns [3H, 4H)).  Rows (b,t) of dG
    pair with out rows (b,t-1) (forward) / (b,t+1) (reverse): the same pairing seen from one row earlier in dG and one row later
    in out (two pointer shifts = the batch strides).  Pointer defaults: shape-only records for the workspace-size query."""
    H, N = cfg.hidden, B * T
    K = cfg.in_dim if l == 0 else 2 * H
    out = [L.GemmProblem(0, 0, 3 * H, K, N, dg, 4 * H, INT_MAX, 0, 0, lin, Kp, INT_MAX, 0, 0, g_ih[0], K, 1, 2, N * 4 * H, 0, g_ih[1])]
    if T > 1:
        Kh = B * (T - 1)
        sa, sb = N * 4 * H - 4 * H, H + 2 * H
        out.append(L.GemmProblem(0, 0, 2 * H, H, Kh, dg, 4 * H, T - 1, T, 1, hp, 2 * H, T - 1, T, 0, g_hh[0], H, 1, 2, sa, sb, g_hh[1]))
        out.append(L.GemmProblem(0, 0, H, H, Kh, dg + 3 * H * 2, 4 * H, T - 1, T, 1, hp, 2 * H, T - 1, T, 0, g_hh[0] + 2 * H * H * 4, H,
                                 1, 2, sa, sb, g_hh[1]))
    return out


def dw_group_ok(problems) -> bool:
    return USE_DW_GROUP and all(q.K % 64 == 0 for q in problems)


DW_GROUP_MAX = 8  # problems per ss_gemm_bf16_splitk_group launch (csrc/gemm_bf16.hip ring::GROUP_MAX)


def dw_flush_after(cfg, l: int, pending: int) -> bool:
    """Does backward() launch the pending weight-gradient problems behind layer ``l`` (walking top-down), ``pending`` of them
    queued including this layer's?  One rule for the launches and for the scratch size (WorkspaceBf16 replays it): a group
    holds at most DW_GROUP_MAX problems and a layer adds up to three, so it is flushed as soon as another layer might not fit."""
    return l == 0 or not DW_ALL_LAYERS or pending + 3 > DW_GROUP_MAX


def dw_group_schedule(cfg, B, T, kp):
    """The groups of shape-only problem records backward() will launch, in launch order."""
    groups, pending = [], []
    for l in range(cfg.gru_layers - 1, -1, -1):
        pr = dw_problems(cfg, B, T, l, kp[l])
        if not dw_group_ok(pr):
            continue  # this layer goes through the three-launch form; the queue is untouched
        pending = pending + pr
        if dw_flush_after(cfg, l, len(pending)):
            groups.append(pending)
            pending = []
    if pending:
        groups.append(pending)
    return groups


def _pstride(P, a: str, b: str) -> int:
    return (P[b].data_ptr() - P[a].data_ptr()) // 4


def _addr(t: torch.Tensor, offset_elems: int = 0) -> int:
    return t.data_ptr() + offset_elems * t.element_size()


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def check_config(cfg, roi_hw) -> None:
    if cfg.hidden % 128 != 0:
        raise RuntimeError("precision='bf16' needs hidden % 128 == 0 (ss_gru_bf16_*); use the f32 path for small hidden sizes")
    if cfg.use_roi and (tuple(cfg.cnn_channels) != CNN_CHANNELS or tuple(roi_hw) != ROI_HW):
        raise RuntimeError(f"the bf16 ROI CNN is built for channels {CNN_CHANNELS} on {ROI_HW[0]}x{ROI_HW[1]} frames "
                           f"(got {tuple(cfg.cnn_channels)} on {tuple(roi_hw)})")
    if cfg.in_dim % 4 != 0 or (cfg.use_roi and cfg.x_dim % 4 != 0):
        raise RuntimeError("precision='bf16' needs x_dim % 4 == 0 and (x_dim + roi_emb) % 4 == 0 (vector loads of the operand "
                           "conversion)")


class WorkspaceBf16:
    """Activation / gradient buffers of the bf16 path for one (B, T) shape."""

    bf16 = True

    def __init__(self, cfg, B: int, T: int, roi_hw, device, train: bool, slot=0):
        check_config(cfg, roi_hw if cfg.use_roi else ROI_HW)
        self.cfg, self.B, self.T, self.roi_hw, self.train = cfg, B, T, roi_hw, train
        self.stash_gen, self.stash_live = 0, False
        self.stagger = False
        # weight-gradient GEMMs of a layer run on a side stream beside the BPTT steps of the layer below (a step is a tiny,
        # latency-bound launch that leaves the matrix pipes idle)
        # (one stream per device and micro-batch slot, shared by all workspaces: engine.side_stream)
        from .engine import side_stream

        self.side = side_stream(device, slot) if train else None
        self.ev_fork = torch.cuda.Event() if train else None
        self.ev_join = torch.cuda.Event() if train else None
        self.ev_cnn_fwd = torch.cuda.Event() if train else None  # recorded after the ROI-CNN forward (Trainer's micro-batch stagger)
        self.ev_zero = torch.cuda.Event() if train else None     # the cleared d layer_in destinations (side stream)
        N, H = B * T, cfg.hidden
        f32 = dict(device=device, dtype=torch.float32)
        u8 = dict(device=device, dtype=torch.uint8)
        i16 = dict(device=device, dtype=torch.int16)  # bf16 bit patterns
        self.lengths = torch.empty(B, device=device, dtype=torch.int32)
        self.Z = torch.empty(N, cfg.in_dim, **f32) if cfg.use_roi else None
        self.gi = [torch.empty(2, N, 3 * H, **f32) for _ in range(cfg.gru_layers)]
        self.out = [torch.empty(N, 2 * H, **f32) for _ in range(cfg.gru_layers)]
        # bf16 copies that feed MFMAs: the layer inputs (Z; the -- dropped-out -- output of the layer below, left by its recurrence
        # kernel), the layer outputs (h_prev operand of d W_hh), W_hh / W_hh^T / W_ih (once per step: ss_gru_bf16_prep)
        self.kin = [cfg.in_dim if l == 0 else 2 * H for l in range(cfg.gru_layers)]
        self.kp = [_pad8(k) for k in self.kin]
        self.lin_bf0 = torch.zeros(N, self.kp[0], **i16)
        self.out_bf = [torch.empty(N, 2 * H, **i16) for _ in range(cfg.gru_layers)]
        self.out_drop_bf = [torch.empty(N, 2 * H, **i16) for _ in range(cfg.gru_layers - 1)] if train else []
        self.whh = [torch.empty(2, 3 * H, H, **i16) for _ in range(cfg.gru_layers)]
        self.whht = [torch.empty(2, H, 3 * H, **i16) for _ in range(cfg.gru_layers)]
        self.wih = [torch.empty(2, 3 * H, self.kp[l], **i16) for l in range(cfg.gru_layers)]
        nb = C.c_long(0)
        if L.load().ss_gru_bf16_ws_bytes(B, H, C.byref(nb)) != 0:
            raise RuntimeError("ss_gru_bf16_ws_bytes failed")
        self.gru_ws = torch.empty(nb.value, **u8)
        # the persistent recurrence's exchange area (tagged granules + launch generation): zeroed ONCE, the kernels keep it consistent
        if L.load().ss_gru_bf16_sync_bytes(B, T, H, C.byref(nb)) != 0:
            raise RuntimeError("ss_gru_bf16_sync_bytes failed")
        self.gru_sync = torch.zeros(nb.value // 4, device=device, dtype=torch.int32) if (nb.value and USE_PERSISTENT_GRU) else None
        self.logits = torch.empty(B, cfg.num_classes, **f32)
        self.attn = torch.empty(B, T, **f32)
        self.mid_drop = torch.empty(B, cfg.head_mid, **f32)
        self.ln = torch.empty(B, 2 * H, **f32)
        self.mid = torch.empty(B, cfg.head_mid, **f32)
        if cfg.use_roi:
            c1, c2, c3, c4 = CNN_CHANNELS
            self.feat = torch.empty(N, c4, **f32)
            # (the pooled conv1 map never reaches HBM: conv1 lands in conv2's LDS image and is recomputed in the backward pass)
            self.a2 = torch.empty(N, 24, 24, c2, **i16)
            self.i2 = torch.empty(N, 24, 24, c2, **u8)
            self.a3 = torch.empty(N, 12, 12, c3, **i16)
            self.i3 = torch.empty(N, 12, 12, c3, **u8)
        if train:
            self.save = [torch.empty(2, N, 4, H, **f32) for _ in range(cfg.gru_layers)]
            # gate gradients: only the bf16 copy (the GEMMs' operand) when the persistent BPTT kernel makes the bias sums itself
            self.dG = [torch.empty(2, N, 4, H, **f32) if self.gru_sync is None else None for _ in range(cfg.gru_layers)]
            self.dG_bf = [torch.empty(2, N, 4, H, **i16) for _ in range(cfg.gru_layers)]
            self.d_out = torch.empty(N, 2 * H, **f32)
            # scratch of the grouped weight-gradient launches: the largest group of the schedule backward() really runs (round 3
            # sized it for one layer or all layers; with three GRU layers the launches are {2, 1} and {0}, and {2, 1} was larger
            # than either -- 18.9 MB written past the end)
            groups = dw_group_schedule(cfg, B, T, self.kp)
            self.dw_ws = torch.empty(max(4, max(L.gemm_group_ws_floats(g, bf16=True) for g in groups)), **f32) if groups else None
            self.d_lower = [None] + [torch.empty(N, 2 * H, **f32) for _ in range(1, cfg.gru_layers)]
            self.xhat = torch.empty(B, 2 * H, **f32)
            self.rstd = torch.empty(B, **f32)
            self.d_logits = torch.empty(B, cfg.num_classes, **f32)
            self.d_mid = torch.empty(B, cfg.head_mid, **f32)
            self.tail_part = torch.empty(B, 3, 2 * H, **f32)
            self.dZ = torch.empty(N, cfg.in_dim, **f32) if cfg.use_roi else None
            if cfg.use_roi:
                c1, c2, c3, c4 = CNN_CHANNELS
                self.st = torch.empty(N, 2, **f32)
                self.i1 = torch.empty(N, 48, 48, c1, **u8) if (STASH_I1 and FUSE_DGRAD2_WGRAD1) else None
                self.m4 = torch.empty(N, 144, c4, **u8)
                self.dfeat = torch.empty(N, c4, **f32)
                # partial weight-gradient sums of the conv layers' persistent workgroups (one per CU): plain stores + a reduce launch
                cus = torch.cuda.get_device_properties(device).multi_processor_count
                self.wg_part = torch.empty(min(N, cus) * c4 * c3 * 9, **f32)
                self.da1 = torch.empty(N, 48, 48, c1, **i16) if not FUSE_DGRAD2_WGRAD1 else None
                self.da2 = torch.empty(N, 24, 24, c2, **i16)
                self.da3 = torch.empty(N, 12, 12, c3, **i16)


_CONV = ("roi_cnn.net.0", "roi_cnn.net.3", "roi_cnn.net.6", "roi_cnn.net.9")


def forward(P: Dict[str, torch.Tensor], cfg, ws: WorkspaceBf16, X: torch.Tensor, R: Optional[torch.Tensor], *, train: bool,
            stash: bool = False, seed: int = 0, ce=None, x_in_place: bool = False) -> torch.Tensor:
    B, T, H, N = ws.B, ws.T, cfg.hidden, ws.B * ws.T
    s = L.stream()
    if stash and not ws.train:
        raise RuntimeError("stash=True needs a training workspace")
    if cfg.use_roi:
        if not x_in_place:
            L.call("ss_copy_rows_f32", X.data_ptr(), cfg.x_dim, ws.Z.data_ptr(), cfg.in_dim, N, cfg.x_dim, s)
        w = [P[k + ".weight"].data_ptr() for k in _CONV]
        b = [P[k + ".bias"].data_ptr() for k in _CONV]
        L.call("ss_c5_conv12_fwd_i1", R.data_ptr(), N, int(cfg.roi_standardize), w[0], b[0], w[1], b[1], ws.a2.data_ptr(), ws.i2.data_ptr(),
               ws.st.data_ptr() if stash else None, L.ptr(ws.i1) if stash else None, s, tag="ss_c5_conv12_fwd")
        L.call("ss_c5_conv_fwd", 3, ws.a2.data_ptr(), N, w[2], b[2], ws.a3.data_ptr(), ws.i3.data_ptr(), s, tag="ss_c5_conv3_fwd")
        if FC_AS_GEMM:
            from .engine import gemm as gemm_f32

            c4 = CNN_CHANNELS[3]
            L.call("ss_c5_conv_last_fwd_feat", ws.a3.data_ptr(), N, w[3], b[3], ws.m4.data_ptr() if stash else None, ws.feat.data_ptr(), s,
                   tag="ss_c5_conv_last_fwd")
            gemm_f32(1, 1, N, cfg.roi_emb, c4, ws.feat.data_ptr(), c4, P["roi_cnn.fc.weight"].data_ptr(), c4, _addr(ws.Z, cfg.x_dim),
                     cfg.in_dim, bias=P["roi_cnn.fc.bias"].data_ptr(), tag="gemm_fc")
        else:
            L.call("ss_c5_conv_last_fwd", ws.a3.data_ptr(), N, w[3], b[3], P["roi_cnn.fc.weight"].data_ptr(), P["roi_cnn.fc.bias"].data_ptr(),
                   cfg.roi_emb, _addr(ws.Z, cfg.x_dim), cfg.in_dim, ws.m4.data_ptr() if stash else None,
                   ws.feat.data_ptr() if stash else None, s)
        if ws.train and ws.stagger:  # only when another micro-batch waits for it (train.Trainer)
            ws.ev_cnn_fwd.record()
        layer_in, ld_in = ws.Z.data_ptr(), cfg.in_dim
    else:
        layer_in, ld_in = X.data_ptr(), cfg.x_dim
    X_src, ld_src = (ws.Z.data_ptr(), cfg.in_dim) if cfg.use_roi else (X.data_ptr(), cfg.x_dim)
    L.call("ss_cvt_bf16_rows", X_src, ld_src, ws.lin_bf0.data_ptr(), ws.kp[0], N, ws.kin[0], 0.0, 0, 0, s)
    lin = ws.lin_bf0
    for l in range(cfg.gru_layers):
        K, Kp = ws.kin[l], ws.kp[l]
        wf, wr = f"gru.weight_ih_l{l}", f"gru.weight_ih_l{l}_reverse"
        L.call("ss_gru_bf16_prep", P[f"gru.weight_hh_l{l}"].data_ptr(), P[f"gru.weight_hh_l{l}_reverse"].data_ptr(), H,
               ws.whh[l].data_ptr(), ws.whht[l].data_ptr(), P[wf].data_ptr(), P[wr].data_ptr(), K, ws.wih[l].data_ptr(), s)
        gemm(1, 1, N, 3 * H, K, lin.data_ptr(), Kp, ws.wih[l].data_ptr(), Kp, ws.gi[l].data_ptr(), 3 * H,
             bias=P[f"gru.bias_ih_l{l}"].data_ptr(), batch=2,
             strides=(0, 3 * H * Kp, N * 3 * H, _pstride(P, f"gru.bias_ih_l{l}", f"gru.bias_ih_l{l}_reverse")), tag="gemm_bf16_ih")
        # nn.GRU's inter-layer dropout is a by-product of the recurrence kernel: it leaves bf16(dropout(out)) as the next layer's operand
        drop = train and l < cfg.gru_layers - 1 and cfg.gru_dropout > 0.0 and ws.train
        L.call("ss_gru_bf16_fwd", ws.gi[l].data_ptr(), ws.whh[l].data_ptr(), P[f"gru.bias_hh_l{l}"].data_ptr(),
               P[f"gru.bias_hh_l{l}_reverse"].data_ptr(), ws.lengths.data_ptr(), B, T, H, ws.out[l].data_ptr(),
               ws.save[l].data_ptr() if stash else None, ws.out_bf[l].data_ptr(), ws.out_drop_bf[l].data_ptr() if drop else None,
               cfg.gru_dropout if drop else 0.0, seed, (l + 1) << 40, ws.gru_ws.data_ptr(), L.ptr(ws.gru_sync), L.nbytes(ws.gru_sync), s)
        lin = ws.out_drop_bf[l] if drop else ws.out_bf[l]
        ws.lin_bf = getattr(ws, "lin_bf", {})
        ws.lin_bf[l + 1] = lin
    top = ws.out[cfg.gru_layers - 1]
    p_drop = cfg.head_dropout if train else 0.0
    y_ptr, ls, denom, loss_ptr, correct_ptr = ce if ce is not None else (None, 0.0, 1.0, None, None)
    L.call("ss_tail_fwd", top.data_ptr(), ws.lengths.data_ptr(), P["pool.score.weight"].data_ptr(), P["pool.score.bias"].data_ptr(),
           P["head.0.weight"].data_ptr(), P["head.0.bias"].data_ptr(), P["head.1.weight"].data_ptr(), P["head.1.bias"].data_ptr(),
           P["head.4.weight"].data_ptr(), P["head.4.bias"].data_ptr(), y_ptr, B, T, 2 * H, cfg.head_mid, cfg.num_classes, cfg.ln_eps,
           p_drop, seed, 7 << 40, ls, denom, ws.attn.data_ptr() if stash else None, ws.xhat.data_ptr() if stash else None,
           ws.rstd.data_ptr() if stash else None, ws.ln.data_ptr() if stash else None, ws.mid.data_ptr() if stash else None,
           ws.mid_drop.data_ptr() if stash else None, ws.logits.data_ptr(), ws.d_logits.data_ptr() if ce is not None else None,
           loss_ptr, correct_ptr, s)
    return ws.logits


def backward(P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor], cfg, ws: WorkspaceBf16, X: torch.Tensor,
             R: Optional[torch.Tensor], d_logits: torch.Tensor, *, train: bool, seed: int = 0,
             d_X: Optional[torch.Tensor] = None) -> None:
    from .engine import gemm as gemm_f32, zero_buffers  # the head's two tiny weight-gradient GEMMs stay f32

    B, T, H, N = ws.B, ws.T, cfg.hidden, ws.B * ws.T
    s = L.stream()
    Cn, MID = cfg.num_classes, cfg.head_mid
    p_drop = cfg.head_dropout if train else 0.0
    top = ws.out[cfg.gru_layers - 1]
    L.call("ss_tail_bwd", top.data_ptr(), ws.lengths.data_ptr(), P["pool.score.weight"].data_ptr(), P["head.0.weight"].data_ptr(),
           P["head.1.weight"].data_ptr(), P["head.4.weight"].data_ptr(), ws.attn.data_ptr(), ws.xhat.data_ptr(), ws.rstd.data_ptr(),
           ws.mid.data_ptr(), d_logits.data_ptr(), B, T, 2 * H, MID, Cn, p_drop, seed, 7 << 40, ws.d_mid.data_ptr(),
           ws.d_out.data_ptr(), G["head.0.weight"].data_ptr(), G["head.0.bias"].data_ptr(), G["pool.score.weight"].data_ptr(),
           G["pool.score.bias"].data_ptr(), ws.tail_part.data_ptr(), s)
    def head_grads():
        # the atomically summed destinations of the d layer_in GEMMs first (the top layer's GEMM waits for them), then what only
        # feeds parameter gradients of the head: column sums of the rows the tail kernel left per clip, the two Linear weight
        # gradients (f32, K = B: latency-bound launches that leave the chip empty -- beside the top layer's BPTT kernel they are free)
        zero_buffers(ws.d_lower[1:] + ([ws.dZ] if cfg.use_roi else []))
        ws.ev_zero.record()
        for k_, name_ in enumerate(("head.0.weight", "head.0.bias", "pool.score.weight")):
            L.call("ss_colsum_f32", _addr(ws.tail_part, k_ * 2 * H), B, 2 * H, 3 * 2 * H, G[name_].data_ptr(), L.stream())
        gemm_f32(0, 0, Cn, MID, B, d_logits.data_ptr(), Cn, ws.mid_drop.data_ptr(), MID, G["head.4.weight"].data_ptr(), MID,
                 accumulate=True, atomic=True, a_colsum=G["head.4.bias"].data_ptr())
        gemm_f32(0, 0, MID, 2 * H, B, ws.d_mid.data_ptr(), MID, ws.ln.data_ptr(), 2 * H, G["head.1.weight"].data_ptr(), 2 * H,
                 accumulate=True, atomic=True, a_colsum=G["head.1.bias"].data_ptr())

    # the fork event is recorded here, the side stream's launches are enqueued behind the top layer's BPTT launch (below): the host
    # needs tens of microseconds for them and the main queue sat empty meanwhile (engine.backward, tools/step_gaps.py)
    if USE_SIDE_STREAM:
        ws.ev_fork.record()

    def head_side_work():
        if USE_SIDE_STREAM:
            with torch.cuda.stream(ws.side):
                ws.side.wait_event(ws.ev_fork)
                head_grads()
        else:
            head_grads()

    zero_waited = False
    dw_pending = []
    use_drop = train and cfg.gru_dropout > 0.0
    for l in range(cfg.gru_layers - 1, -1, -1):
        K = cfg.in_dim if l == 0 else 2 * H
        top_layer = l == cfg.gru_layers - 1
        g_in = ws.d_out if top_layer else ws.d_lower[l + 1]
        L.call("ss_gru_bf16_bwd", g_in.data_ptr(), ws.out[l].data_ptr(), ws.save[l].data_ptr(), ws.whht[l].data_ptr(),
               ws.lengths.data_ptr(), B, T, H, L.ptr(ws.dG[l]), ws.dG_bf[l].data_ptr(),
               0.0 if (top_layer or not use_drop) else cfg.gru_dropout, seed,
               (l + 1) << 40, G[f"gru.bias_ih_l{l}"].data_ptr(), G[f"gru.bias_hh_l{l}"].data_ptr(),
               G[f"gru.bias_ih_l{l}_reverse"].data_ptr(), G[f"gru.bias_hh_l{l}_reverse"].data_ptr(), ws.gru_ws.data_ptr(),
               L.ptr(ws.gru_sync), L.nbytes(ws.gru_sync), s)
        if top_layer:
            head_side_work()
        Kp = ws.kp[l]
        lin = (ws.lin_bf0 if l == 0 else ws.lin_bf[l]).data_ptr()  # what the forward pass multiplied W_ih with (dropped out or not)
        dg = ws.dG_bf[l].data_ptr()
        wi, wir = f"gru.weight_ih_l{l}", f"gru.weight_ih_l{l}_reverse"

        def param_grads(l=l, K=K, Kp=Kp, lin=lin, dg=dg, wi=wi, wir=wir):
            # (the bias gradients are by-products of the BPTT kernel)
            wh_, whr_ = f"gru.weight_hh_l{l}", f"gru.weight_hh_l{l}_reverse"
            pr = dw_problems(cfg, B, T, l, Kp, dg, lin, ws.out_bf[l].data_ptr(), (G[wi].data_ptr(), _pstride(G, wi, wir)),
                             (G[wh_].data_ptr(), _pstride(G, wh_, whr_)))
            if ws.dw_ws is not None and dw_group_ok(pr):
                # ALL layers' weight gradients in ONE launch, behind the bottom layer's BPTT kernel: 216 output tiles at the
                # config-5 shapes = one workgroup per tile and CU walking all of K (no K split, no slabs; neighbouring tiles share
                # operand panels in L2) -- per layer (144 / 72 tiles) the K tiles had to be dealt over the CUs, 154 + 80 us
                dw_pending.extend(pr)
                if dw_flush_after(cfg, l, len(dw_pending)):
                    arr, n = L.gemm_group(dw_pending)
                    L.call("ss_gemm_bf16_splitk_group", arr, n, ws.dw_ws.data_ptr(), ws.dw_ws.numel(), L.stream(), tag="gemm_bf16_dW")
                    dw_pending.clear()
                return
            # ---- weight gradients: d W_ih = dGi^T . layer_in;  d W_hh = dGh^T . h_prev (rows r|z from columns [0,2H), rows n from [3H,4H))
            gemm(0, 0, 3 * H, K, N, dg, 4 * H, lin, Kp, G[wi].data_ptr(), K, accumulate=True, atomic=True,
                 splits=split_k(3 * H, K, N, 2), batch=2, strides=(N * 4 * H, 0, _pstride(G, wi, wir), 0), tag="gemm_bf16_dW")
            if T > 1:
                wh, whr = f"gru.weight_hh_l{l}", f"gru.weight_hh_l{l}_reverse"
                Kh = B * (T - 1)
                am, bm = (T - 1, T, 1), (T - 1, T, 0)
                # forward direction pairs dG[b][t] with out[b][t-1]; the reverse direction dG[b][t] with out[b][t+1]: the same
                # pairing seen from one row earlier in dG and one row later in out (two pointer shifts = the batch strides)
                sa, sb, sc = N * 4 * H - 4 * H, H + 2 * H, _pstride(G, wh, whr)
                hp = ws.out_bf[l].data_ptr()
                gemm(0, 0, 2 * H, H, Kh, dg, 4 * H, hp, 2 * H, G[wh].data_ptr(), H, accumulate=True, atomic=True,
                     splits=split_k(2 * H, H, Kh, 2), a_map=am, b_map=bm, batch=2, strides=(sa, sb, sc, 0), tag="gemm_bf16_dW")
                gemm(0, 0, H, H, Kh, dg + 3 * H * 2, 4 * H, hp, 2 * H, _addr(G[wh], 2 * H * H), H, accumulate=True, atomic=True,
                     splits=split_k(H, H, Kh, 2), a_map=am, b_map=bm, batch=2, strides=(sa, sb, sc, 0), tag="gemm_bf16_dW")

        # ---- d layer_in = dGi_f . W_ih_f + dGi_r . W_ih_r (both directions in one launch, float atomics into a cleared buffer):
        # on the critical path (the layer below / the CNN backward waits for it), so it goes first
        need_dx = (l > 0) or cfg.use_roi or (d_X is not None)
        if need_dx:
            if l > 0:
                dst, ld_dst, c0 = ws.d_lower[l].data_ptr(), 2 * H, 0
            elif cfg.use_roi:
                # nobody asked for d X: only the ROI-embedding columns of d Z feed the CNN backward (from the 8-column chunk they start in)
                c0 = cfg.x_dim // 8 * 8 if d_X is None else 0
                dst, ld_dst = ws.dZ.data_ptr() + 4 * c0, cfg.in_dim
            else:
                zero_buffers([d_X])
                dst, ld_dst, c0 = d_X.data_ptr(), cfg.x_dim, 0
            if USE_SIDE_STREAM and not zero_waited:  # once per backward pass: every cleared buffer is behind the same event
                torch.cuda.current_stream().wait_event(ws.ev_zero)
                zero_waited = True
            # (layer 0 with the ROI branch: 68 output columns = 30 tiles of the ring kernel -- K slices and atomics fill more CUs)
            if USE_DX_KCAT and (3 * H) % 64 == 0 and K - c0 >= 128:
                L.call("ss_gemm_bf16_batched", 1, 0, N, K - c0, 3 * H, dg, 4 * H, *_IDENT, _addr(ws.wih[l], c0), Kp, *_IDENT, dst, ld_dst,
                       None, 8 | 16, 1, 2, N * 4 * H, 3 * H * Kp, 0, 0, L.stream(), tag="gemm_bf16_dX")
            else:
                gemm(1, 0, N, K - c0, 3 * H, dg, 4 * H, _addr(ws.wih[l], c0), Kp, dst, ld_dst, accumulate=True, atomic=True, batch=2,
                     splits=2 if K - c0 < 128 else 1, strides=(N * 4 * H, 3 * H * Kp, 0, 0), tag="gemm_bf16_dX")
        if USE_SIDE_STREAM:
            ws.ev_fork.record()
            with torch.cuda.stream(ws.side):
                ws.side.wait_event(ws.ev_fork)
                param_grads()
        else:
            param_grads()
    assert not dw_pending, "a weight-gradient group was queued and never launched"  # (dw_flush_after flushes at l == 0)
    if cfg.use_roi:
        if d_X is not None:
            L.call("ss_copy_rows_f32", ws.dZ.data_ptr(), cfg.in_dim, d_X.data_ptr(), cfg.x_dim, N, cfg.x_dim, s)
        w = [P[k + ".weight"].data_ptr() for k in _CONV]
        gw = [G[k + ".weight"].data_ptr() for k in _CONV]
        gb = [G[k + ".bias"].data_ptr() for k in _CONV]
        dz = _addr(ws.dZ, cfg.x_dim)
        wfc = P["roi_cnn.fc.weight"].data_ptr()
        if FC_AS_GEMM:
            from .engine import split_k as split_k_f32

            c4, E = CNN_CHANNELS[3], cfg.roi_emb
            gemm_f32(1, 0, N, c4, E, dz, cfg.in_dim, wfc, c4, ws.dfeat.data_ptr(), c4, tag="gemm_fc")  # d z . W_fc

            def fc_grads():  # d W_fc += d z^T . feat, d b_fc += column sums of d z
                gemm_f32(0, 0, E, c4, N, dz, cfg.in_dim, ws.feat.data_ptr(), c4, G["roi_cnn.fc.weight"].data_ptr(), c4, accumulate=True,
                         atomic=True, splits=split_k_f32(E, c4, N), a_colsum=G["roi_cnn.fc.bias"].data_ptr(), tag="gemm_fc")

            if USE_SIDE_STREAM:
                ws.ev_fork.record()
                with torch.cuda.stream(ws.side):
                    ws.side.wait_event(ws.ev_fork)
                    fc_grads()
            else:
                fc_grads()
            L.call("ss_c5_conv_last_wgrad_df", ws.a3.data_ptr(), ws.dfeat.data_ptr(), ws.m4.data_ptr(), N, gw[3], gb[3],
                   ws.wg_part.data_ptr(), ws.wg_part.numel(), s, tag="ss_c5_conv_last_wgrad")
            L.call("ss_c5_conv_last_dgrad_df", ws.dfeat.data_ptr(), ws.m4.data_ptr(), N, w[3], ws.da3.data_ptr(), s,
                   tag="ss_c5_conv_last_dgrad")
        else:
            L.call("ss_c5_conv_last_wgrad", ws.a3.data_ptr(), dz, cfg.in_dim, cfg.roi_emb, wfc, ws.m4.data_ptr(), ws.feat.data_ptr(), N,
                   gw[3], gb[3], G["roi_cnn.fc.weight"].data_ptr(), G["roi_cnn.fc.bias"].data_ptr(), s)
            L.call("ss_c5_conv_last_dgrad", dz, cfg.in_dim, cfg.roi_emb, wfc, ws.m4.data_ptr(), N, w[3], ws.da3.data_ptr(), s)
        # (partial sums through scratch + a reduce launch pay for the last layer's 55 k elements only: launch time at B = 256, T = 30,
        # tools/c5_fixed_cost.py / c5_last_bench.py with SS_NO_PART=1: layer 4 152 - 156 us against 172 - 174 with float atomics onto the
        # gradient, layer 3 equal, layer 2 (4.6 k elements) 376 against 362)
        L.call("ss_c5_conv_wgrad", 3, ws.a2.data_ptr(), ws.da3.data_ptr(), ws.i3.data_ptr(), N, gw[2], gb[2], s, tag="ss_c5_conv3_wgrad")
        L.call("ss_c5_conv_dgrad", 3, ws.da3.data_ptr(), ws.i3.data_ptr(), N, w[2], ws.da2.data_ptr(), s, tag="ss_c5_conv3_dgrad")
        bb = [P[k + ".bias"].data_ptr() for k in _CONV]
        L.call("ss_c5_conv2_wgrad_rc", R.data_ptr(), ws.st.data_ptr(), int(cfg.roi_standardize), w[0], bb[0], ws.da2.data_ptr(),
               ws.i2.data_ptr(), N, gw[1], gb[1], s)
        if FUSE_DGRAD2_WGRAD1:  # d a1 (the largest gradient map) is born and consumed in LDS: 1.13 GB per step less through HBM
            L.call("ss_c5_conv2_dgrad_conv1_wgrad_i1", ws.da2.data_ptr(), ws.i2.data_ptr(), N, w[1], R.data_ptr(), ws.st.data_ptr(),
                   int(cfg.roi_standardize), w[0], bb[0], None, gw[0], gb[0], L.ptr(ws.i1), s, tag="ss_c5_conv2_dgrad_conv1_wgrad")
        else:
            L.call("ss_c5_conv_dgrad", 2, ws.da2.data_ptr(), ws.i2.data_ptr(), N, w[1], ws.da1.data_ptr(), s, tag="ss_c5_conv2_dgrad")
            L.call("ss_c5_conv1_wgrad", R.data_ptr(), N, int(cfg.roi_standardize), ws.st.data_ptr(), ws.da1.data_ptr(), None, w[0],
                   bb[0], gw[0], gb[0], s)
    if USE_SIDE_STREAM:  # the caller's next kernels (all-reduce, clip, Adam) read every gradient
        ws.ev_join.record(ws.side)
        torch.cuda.current_stream().wait_event(ws.ev_join)
