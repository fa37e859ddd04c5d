"""Drop-in ``BiGRUClassifier`` for /root/reference/train_model_official.py:253-310 and
/root/reference/live_infer_official.py:101-138, running on the HIP kernels.

Kept from the reference: constructor arguments, ``forward(X, lengths, R)``,
the attributes ``use_roi / roi_cnn / gru / pool / head`` and every ``state_dict`` key name, shape and
dtype, so ``.pt`` checkpoints interchange in both directions.  The sub-modules below are parameter
holders only -- they have no aten forward; all arithmetic happens in ``libss_hotpath.so``.

All parameters are views into one flat fp32 buffer (``model.flat_params``), and their ``.grad``s are
views into ``model.flat_grads``: the optimiser step and the data-parallel all-reduce see a single
contiguous bucket.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import engine as E


def _uniform_(t: torch.Tensor, bound: float) -> torch.Tensor:
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the arithmetic of this layer runs inside the fused HIP kernels")


class ConvParams(_Holder):
    """Conv2d(cin, cout, 3, padding=1) parameters with torch's default init (kaiming_uniform a=sqrt(5))."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        bound = 1.0 / math.sqrt(cin * 9)
        self.weight = nn.Parameter(_uniform_(torch.empty(cout, cin, 3, 3), bound))
        self.bias = nn.Parameter(_uniform_(torch.empty(cout), bound))


class LinearParams(_Holder):
    def __init__(self, fin: int, fout: int):
        super().__init__()
        bound = 1.0 / math.sqrt(fin)
        self.weight = nn.Parameter(_uniform_(torch.empty(fout, fin), bound))
        self.bias = nn.Parameter(_uniform_(torch.empty(fout), bound))


class LayerNormParams(_Holder):
    def __init__(self, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Slot(_Holder):
    """Parameter-less placeholder keeping the reference's Sequential indices (ReLU / MaxPool / Dropout)."""


def _indexed(mods: Dict[int, nn.Module], n: int) -> nn.Module:
    box = _Holder()
    for i in range(n):
        box.add_module(str(i), mods.get(i, _Slot()))
    return box


class TinyROICNN(_Holder):
    """train_model_official.py:209-229: net.{0,3,6} convs, fc.  ``channels`` widens / deepens it the way the reference's
    Sequential would grow: [conv, ReLU, MaxPool] per block (conv at index 3 i), the last block without the pool, then the
    global average and ``fc`` -- BASELINE config 5 uses (16, 32, 64, 96)."""

    def __init__(self, out_dim: int = 32, channels=(8, 16, 24)):
        super().__init__()
        chans = (1,) + tuple(channels)
        self.net = _indexed({3 * i: ConvParams(chans[i], chans[i + 1]) for i in range(len(channels))}, 3 * len(channels) - 1)
        self.fc = LinearParams(chans[-1], out_dim)


class AttnPool(_Holder):
    """train_model_official.py:231-248."""

    def __init__(self, dim: int):
        super().__init__()
        self.score = LinearParams(dim, 1)


class GRUParams(_Holder):
    """nn.GRU(in_dim, hidden, num_layers, bidirectional=True) parameter names, shapes and init."""

    def __init__(self, in_dim: int, hidden: int, num_layers: int):
        super().__init__()
        bound = 1.0 / math.sqrt(hidden)
        for l in range(num_layers):
            lin = in_dim if l == 0 else 2 * hidden
            for suf in ("", "_reverse"):
                self.register_parameter(f"weight_ih_l{l}{suf}", nn.Parameter(_uniform_(torch.empty(3 * hidden, lin), bound)))
                self.register_parameter(f"weight_hh_l{l}{suf}", nn.Parameter(_uniform_(torch.empty(3 * hidden, hidden), bound)))
                self.register_parameter(f"bias_ih_l{l}{suf}", nn.Parameter(_uniform_(torch.empty(3 * hidden), bound)))
                self.register_parameter(f"bias_hh_l{l}{suf}", nn.Parameter(_uniform_(torch.empty(3 * hidden), bound)))


class _Fn(torch.autograd.Function):
    """Whole-model autograd node: forward = the fused forward kernels, backward = the fused backward kernels."""

    @staticmethod
    def forward(ctx, model, X, R, seed, ws, *params):
        P = model._param_dict()
        logits = E.forward(P, model.cfg, ws, X, R, train=model.training, stash=True, seed=seed)
        ctx.model, ctx.ws, ctx.X, ctx.R, ctx.seed, ctx.train = model, ws, X, R, seed, model.training
        ctx.need_dx = X.requires_grad
        # the stash of this forward lives in ``ws`` until its backward has run: stamp it, so that a later forward
        # that had to take the workspace over is noticed instead of silently corrupting this node's gradients
        model._stash_counter += 1  # one counter for all workspaces: the smallest live stamp is the oldest stash
        ws.stash_gen = model._stash_counter
        ws.stash_live = True
        ctx.gen = ws.stash_gen
        return logits.clone()

    @staticmethod
    def backward(ctx, d_logits):
        model, ws = ctx.model, ctx.ws
        if ws.stash_gen != ctx.gen or not ws.stash_live:
            raise RuntimeError(
                "silent_speech_amd: the activations stashed by this forward pass are gone -- more than "
                f"{model.max_live_stashes} grad-enabled forward passes of one shape were alive at once (or backward ran "
                "twice). Call backward() before further forwards, wrap evaluation in torch.no_grad(), or raise "
                "model.max_live_stashes.")
        ws.stash_live = False
        P = model._param_dict()
        names = list(P.keys())
        scratch = torch.zeros_like(model.flat_params)
        G = model._views_of(scratch)
        d_X = torch.empty_like(ctx.X) if ctx.need_dx else None
        E.backward(P, G, model.cfg, ws, ctx.X, ctx.R, d_logits.contiguous(), train=ctx.train, seed=ctx.seed, d_X=d_X)
        return (None, d_X, None, None, None, *[G[k] for k in names])


class BiGRUClassifier(nn.Module):
    """Same signature as train_model_official.py:254 plus live_infer_official.py:102's ``gru_layers`` and an
    explicit ``roi_standardize`` flag (True = training semantics, train_model_official.py:288-291;
    False = the live script's /255-only variant, live_infer_official.py:126)."""

    def __init__(self, x_dim, num_classes, use_roi=False, roi_emb=32, hidden=192, gru_layers=2,
                 roi_standardize=True, cnn_channels=(8, 16, 24), precision="f32"):
        super().__init__()
        if precision not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' (exact-f32 MFMA, the reference's numerics) or 'bf16' (bf16 MFMA operands, "
                             "f32 accumulation and master weights: BASELINE config 5)")
        if precision == "f32" and use_roi and tuple(cnn_channels) != (8, 16, 24):
            raise ValueError("the f32 ROI-CNN kernels are built for the reference's channels (8, 16, 24); wider CNNs run with "
                             "precision='bf16'")
        self.use_roi = use_roi
        self.roi_cnn = TinyROICNN(out_dim=roi_emb, channels=cnn_channels) if use_roi else None
        in_dim = x_dim + (roi_emb if use_roi else 0)
        self.gru = GRUParams(in_dim, hidden, gru_layers)
        self.pool = AttnPool(hidden * 2)
        self.head = _indexed({0: LayerNormParams(hidden * 2), 1: LinearParams(hidden * 2, 128),
                              4: LinearParams(128, num_classes)}, 5)
        self.cfg = E.Config(x_dim=x_dim, num_classes=num_classes, use_roi=bool(use_roi), roi_emb=roi_emb,
                            hidden=hidden, gru_layers=gru_layers, roi_standardize=roi_standardize,
                            gru_dropout=0.0 if gru_layers < 2 else 0.1, cnn_channels=tuple(cnn_channels), precision=precision)
        self.flat_params: Optional[torch.Tensor] = None
        self.flat_grads: Optional[torch.Tensor] = None
        self._ws_cache = {}
        self._step_seed = 0
        # grad-enabled forward passes of one shape whose stashes may be alive together (each owns a workspace)
        self.max_live_stashes = 2
        self._stash_counter = 0
        self._bucket_version = 0
        self._flatten()

    # ------------------------------------------------------------------ flat parameter bucket
    def _layout(self):
        off, lay = 0, []
        for name, p in self.named_parameters():
            lay.append((name, off, p.numel(), tuple(p.shape)))
            off += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned for the vector loads
        return lay, off

    def _views_of(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {name: flat[off:off + n].view(shape) for name, off, n, shape in self._layout()[0]}

    def _bucket_intact(self) -> bool:
        """True when every parameter still is the right view of ``flat_params`` (e.g. after a ``.to()`` that moved nothing)."""
        flat = self.flat_params
        if flat is None or self.flat_grads is None:
            return False
        params = dict(self.named_parameters())
        for name, off, n, shape in self._layout()[0]:
            p = params[name]
            if (p.dtype != torch.float32 or p.device != flat.device or tuple(p.shape) != shape or not p.is_contiguous()
                    or p.data_ptr() != flat.data_ptr() + 4 * off):
                return False
        return True

    def _flatten(self):
        if self._bucket_intact():
            return
        params = dict(self.named_parameters())
        lay, total = self._layout()
        dev = next(iter(params.values())).device
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        gflat = torch.zeros(total, device=dev, dtype=torch.float32)
        for name, off, n, shape in lay:
            p = params[name]
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(shape)
            p.grad = None
        self.flat_params, self.flat_grads = flat, gflat
        self._ws_cache = {}
        self._bucket_version += 1  # Trainer re-resolves its views of the bucket (and moves its Adam state) when this moves

    def attach_flat_grads(self):
        """Point every ``param.grad`` at its slice of ``flat_grads`` (used by the fused trainer)."""
        views = self._views_of(self.flat_grads)
        for name, p in self.named_parameters():
            p.grad = views[name]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def _param_dict(self) -> Dict[str, torch.Tensor]:
        return {k: p.data for k, p in self.named_parameters()}

    # ------------------------------------------------------------------ workspace
    def _workspace(self, X, R, train: bool, slot=0) -> E.Workspace:
        """Buffers for one (B, T, H, W) shape; ``slot`` separates micro-batches that are in flight together."""
        B, T, _ = X.shape
        hw = tuple(R.shape[2:]) if R is not None else None
        key = (B, T, hw, train, X.device, slot)
        ws = self._ws_cache.get(key)
        if ws is None:
            ws = E.make_workspace(self.cfg, B, T, hw, X.device, train, slot)
            self._ws_cache[key] = ws
        return ws

    def check_health(self) -> None:
        """Raises if any kernel of this module reported a failure it could not return as a status code (today: a lost
        partner in the multi-CU GRU recurrence).  One small device read per workspace -- for points that synchronise anyway."""
        for ws in self._ws_cache.values():
            E.check_gru_sync(ws)

    def _stash_workspace(self, X, R) -> E.Workspace:
        """Training workspace for an autograd forward: the first one of this shape whose stash is not waiting for a
        backward pass.  ``la = m(Xa); lb = m(Xb); (la + lb).backward()`` therefore gets two workspaces; when
        ``max_live_stashes`` are alive the oldest is taken over and ITS backward raises (see _Fn.backward)."""
        cands = []
        for i in range(max(1, self.max_live_stashes)):  # built on demand: a training workspace is hundreds of MB
            ws = self._workspace(X, R, train=True, slot=("autograd", i))
            if not ws.stash_live:
                return ws
            cands.append(ws)
        ws = min(cands, key=lambda w: w.stash_gen)
        ws.stash_live = False
        return ws

    def _check_inputs(self, X, lengths, R):
        if not X.is_cuda:
            raise RuntimeError("silent_speech_amd.BiGRUClassifier runs on an MI355X HIP device only; "
                               "move the module and its inputs to 'cuda' (there is no CPU path)")
        if self.flat_params.device != X.device:
            raise RuntimeError("module parameters and inputs are on different devices")
        if X.dim() != 3 or X.shape[2] != self.cfg.x_dim:
            raise RuntimeError(f"X must be (B,T,{self.cfg.x_dim}), got {tuple(X.shape)}")
        if lengths.shape[0] != X.shape[0]:
            raise RuntimeError("lengths must be (B,)")
        if self.use_roi:
            if R is None or R.dtype != torch.uint8 or R.dim() != 4 or R.shape[:2] != X.shape[:2]:
                raise RuntimeError("use_roi=True needs R: uint8 (B,T,H,W)")

    # ------------------------------------------------------------------ sliding-window serving: embeddings made once per frame
    def embed_rois(self, frames: torch.Tensor, out: Optional[torch.Tensor] = None, ld_out: Optional[int] = None) -> torch.Tensor:
        """TinyROICNN (with the per-frame normalisation of train_model_official.py:286-291) on n separate frames: (n,H,W) uint8 ->
        (n, roi_emb).  A frame's embedding depends on that frame alone, so a sliding window need not recompute the 58 frames it
        shares with the window before (serving.StreamServer caches them).  ``out`` / ``ld_out``: write into rows of a wider matrix."""
        if not self.use_roi or self.cfg.precision != "f32":
            raise RuntimeError("embed_rois needs an f32 use_roi model")
        if frames.dtype != torch.uint8 or frames.dim() != 3 or not frames.is_cuda:
            raise RuntimeError("frames must be uint8 (n,H,W) on the HIP device")
        n, Hh, Ww = frames.shape
        E_ = self.cfg.roi_emb
        if out is None:
            out, ld_out = torch.empty(n, E_, device=frames.device), E_
        P = self._param_dict()
        cw = [P[k].data_ptr() for k in ("roi_cnn.net.0.weight", "roi_cnn.net.0.bias", "roi_cnn.net.3.weight", "roi_cnn.net.3.bias",
                                        "roi_cnn.net.6.weight", "roi_cnn.net.6.bias", "roi_cnn.fc.weight", "roi_cnn.fc.bias")]
        E.L.call("ss_roi_cnn_fwd", frames.contiguous().data_ptr(), n, Hh, Ww, int(self.cfg.roi_standardize), *cw, E_, out.data_ptr(),
                 int(ld_out), E.L.stream())
        return out

    def forward_embedded(self, Z: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """Logits from rows that already hold torch.cat((X, roi_emb)) (train_model_official.py:297): everything behind the ROI branch.
        Inference only."""
        if not self.use_roi or self.cfg.precision != "f32" or Z.dim() != 3 or Z.shape[2] != self.cfg.in_dim or not Z.is_cuda:
            raise RuntimeError(f"Z must be (B,T,{self.cfg.in_dim}) on the HIP device of a use_roi model")
        Z = Z.contiguous().float()
        B, T, _ = Z.shape
        key = (B, T, "embedded", False, Z.device, 0)
        ws = self._ws_cache.get(key)
        if ws is None:
            ws = self._ws_cache[key] = E.make_workspace(self.cfg, B, T, None, Z.device, False)
        ws.lengths.copy_(lengths.to(torch.int32), non_blocking=True)
        with torch.no_grad():
            return E.forward(self._param_dict(), self.cfg, ws, Z, None, train=False, z_ready=True).clone()

    def forward(self, X, lengths, R=None):
        self._check_inputs(X, lengths, R)
        X = X.contiguous().float()
        if self.use_roi:
            R = R.contiguous()
            if not R.is_cuda:
                R = R.to(X.device)
        else:
            R = None
        need_grad = torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in self.parameters()))
        ws = self._stash_workspace(X, R) if need_grad else self._workspace(X, R, train=False)
        # lengths may live anywhere (the reference calls lengths.cpu()); the kernels want int32 on device
        ws.lengths.copy_(lengths.to(torch.int32), non_blocking=True)
        if need_grad:
            self._step_seed += 1
            return _Fn.apply(self, X, R, self._step_seed, ws, *self.parameters())
        logits = E.forward(self._param_dict(), self.cfg, ws, X, R, train=self.training, seed=self._step_seed)
        return logits.clone()
