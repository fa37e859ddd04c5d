"""Sliding-window serving of many streams on one GPU (BASELINE config 4's "4 096 concurrent streams").

Per stream the reference keeps ``deque(maxlen=max_t)`` of frame features and, every ``PRED_EVERY`` frames once
``WARMUP_MIN`` frames are in, zero-pads it to ``(max_t, D)`` and runs the model
(/root/reference/inactive/live_feed.py:155, :163-164, :201-213); important_landmarks.py:136-144 smooths the mouth openness
and applies an open/close hysteresis.  ``StreamServer`` holds the rings of S streams in HBM (``ss_ring_push``), builds the
windows of the streams that are due with one map kernel and the batch-assembly gathers, and runs ONE forward for all of
them -- the model sees ``lengths`` = frames held, so the padding never reaches the recurrence.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .model import BiGRUClassifier

PRED_EVERY, EMA_ALPHA, OPEN_THR, CLOSE_THR = 2, 0.25, 0.02, 0.02
# important_landmarks.py:49-54: landmark ids of the openness signal
MOUTH_TOP, MOUTH_BOTTOM, LEFT_EYE_CORNER, RIGHT_EYE_CORNER = 13, 14, 33, 263
# inactive/live_test_5.py:146-150 and :256 (clips shorter than 6 frames are not classified)
OPEN_THRESH, START_N, END_N, MAX_CLIP, MIN_CLIP = 0.18, 3, 5, 60, 6


def mouth_openness(landmarks: torch.Tensor, mode: str = "eye_span", idx=(MOUTH_TOP, MOUTH_BOTTOM, LEFT_EYE_CORNER, RIGHT_EYE_CORNER)):
    """(n, K, 2) f32 normalised landmarks on the device -> (n,) float64 openness.  ``eye_span``:
    important_landmarks.py:131-133 (lip gap over the eye-corner distance; ``idx`` = positions of landmarks 13, 14, 33, 263
    in the K given); ``y_range``: inactive/live_test_5.py:92-94 (max y - min y over the K landmarks); ``width_norm``:
    inactive/live_feed.py:69-78 in float32 (|lm[13] - lm[14]| over the mouth width |lm[291] - lm[61]| + 1e-6; ``idx`` =
    positions of landmarks 13, 14, 61, 291)."""
    if not landmarks.is_cuda:
        raise RuntimeError("mouth_openness runs on the HIP device (there is no CPU path)")
    lm = landmarks.to(torch.float32).contiguous()
    n, K, _ = lm.shape
    out = torch.empty(n, device=lm.device, dtype=torch.float64)
    L.call("ss_mouth_openness", lm.data_ptr(), n, K, {"eye_span": 0, "y_range": 1, "width_norm": 2}[mode], *[int(v) for v in idx], out.data_ptr(),
           L.stream())
    return out


class LiveFrontEnd:
    """The per-frame front of the live script for S streams as ONE chain of device launches
    (/root/reference/live_infer_official.py:264-296): distance gate on the mouth width -> ``extract_feature`` with the
    stream's velocity state (``prev_xy``, cleared whenever a frame of a recording stream falls outside the 60-150 px band,
    :295-296) -> ``crop_roi_gray`` box -> BGR2GRAY + INTER_AREA resize (zeros when the box is rejected, :292-293).
    State per stream lives in HBM: the previous kept frame's landmarks and a flag (``ss_feature_fuse_stream``)."""

    def __init__(self, n_streams: int, idxs: Sequence[int], frame_wh, roi_hw=None, variant: str = "live",
                 band=(None, None), device="cuda"):
        from . import features as F

        L.load()
        self.S, self.idxs, self.K = n_streams, tuple(int(i) for i in idxs), len(idxs)
        self.w, self.h = int(frame_wh[0]), int(frame_wh[1])
        self.roi_hw, self.variant = roi_hw, variant
        self.band = (float(F.MOUTH_W_MIN_PX if band[0] is None else band[0]), float(F.MOUTH_W_MAX_PX if band[1] is None else band[1]))
        self.anchors = F.anchor_positions(self.idxs)
        dev = self.device = torch.device(device)
        self.prev_lm = torch.zeros(n_streams, self.K, 2, device=dev)
        self.has_prev = torch.zeros(n_streams, device=dev, dtype=torch.uint8)

    def reset(self, stream_ids: Optional[Sequence[int]] = None):
        """``prev_xy = None`` for the given streams (all by default): the start of a recording, live...:333-336."""
        if stream_ids is None:
            self.has_prev.zero_()
        else:
            self.has_prev[torch.as_tensor(list(stream_ids), device=self.device, dtype=torch.long)] = 0

    def __call__(self, stream_ids: Sequence[int], lm: torch.Tensor, frames_bgr: Optional[torch.Tensor] = None,
                 recording: Optional[torch.Tensor] = None):
        """lm (n,K,2) f32 normalised landmarks of the K selected indices, frames_bgr (n,h,w,3) u8 (or None without ROI) ->
        kept (n,) bool on the HOST (the one read-back of a tick), X (n,2K+4) f32, rois (n,H,W) u8 or None; rows of dropped
        frames are zeros and must not be appended."""
        from . import features as F

        ids = np.asarray(stream_ids, np.int32)
        n = len(ids)
        if len(set(ids.tolist())) != n:
            raise ValueError("one frame per stream and tick")
        if n and (ids.min() < 0 or ids.max() >= self.S):
            raise ValueError(f"stream ids must be in [0, {self.S})")
        if not lm.is_cuda:
            raise RuntimeError("the live front end runs on the HIP device (there is no CPU path)")
        if tuple(lm.shape) != (n, self.K, 2) or lm.dtype != torch.float32:
            raise ValueError(f"lm must be float32 ({n}, {self.K}, 2), got {lm.dtype} {tuple(lm.shape)}")
        dev = self.device
        lm = lm.contiguous()
        ids_d = torch.from_numpy(ids).to(dev)
        rec = recording.to(dev, torch.uint8).contiguous() if recording is not None else None
        if rec is not None and rec.numel() != n:
            raise ValueError(f"recording must hold {n} values")
        D = 2 * self.K + 4
        X = torch.empty(n, D, device=dev)
        center = torch.empty(n, 2, device=dev)
        fourth = torch.empty(n, device=dev, dtype=torch.float64)
        kept = torch.empty(n, device=dev, dtype=torch.uint8)
        L.call("ss_feature_fuse_stream", lm.data_ptr(), ids_d.data_ptr(), L.ptr(rec), n, self.S, self.K, self.w, self.h, *self.anchors,
               F._VARIANTS[self.variant], self.band[0], self.band[1], self.prev_lm.data_ptr(), self.has_prev.data_ptr(),
               X.data_ptr(), D, center.data_ptr(), fourth.data_ptr(), kept.data_ptr(), L.stream())
        rois = None
        if self.roi_hw is not None:
            if frames_bgr is None or tuple(frames_bgr.shape) != (n, self.h, self.w, 3) or frames_bgr.dtype != torch.uint8:
                raise ValueError(f"frames_bgr must be uint8 ({n}, {self.h}, {self.w}, 3)")
            boxes = F.crop_boxes(center, fourth, self.w, self.h, self.variant)
            rois = F.crop_rois(frames_bgr.to(dev), boxes, self.roi_hw, self.variant)
        return kept.cpu().numpy().astype(bool), X, rois


class StreamServer:
    """``cache_embeddings`` (default: on for an f32 ROI model whose frame size the fused CNN kernels are built for): a frame's ROI
    embedding depends on that frame alone (the normalisation is per frame, train_model_official.py:286-291), and consecutive windows of
    a stream share all but ``pred_every`` frames -- so the CNN runs ONCE per frame, when it is pushed, the ring keeps the row
    torch.cat((features, embedding)) (:297) instead of the 4.6 KB of pixels, and a prediction is the recurrence, the pool and the
    head on rows that are already there.  Same logits, bit for bit, as recomputing every window from its pixels
    (``cache_embeddings=False``; test_stream_server_embedding_cache_changes_nothing)."""

    def __init__(self, model: BiGRUClassifier, n_streams: int, max_t: int, roi_hw=None, pred_every: int = PRED_EVERY,
                 warmup_min: Optional[int] = None, device="cuda", cache_embeddings: Optional[bool] = None):
        L.load()
        self.model, self.S, self.max_t, self.roi_hw = model.eval(), n_streams, max_t, roi_hw
        from .cnn_generic import fused_supported

        can_cache = bool(roi_hw) and model.use_roi and model.cfg.precision == "f32" and fused_supported(*roi_hw)
        if cache_embeddings and not can_cache:
            raise ValueError("cache_embeddings needs an f32 ROI model and a frame size of the fused CNN kernels")
        self.cache = can_cache if cache_embeddings is None else bool(cache_embeddings)
        self.Dx = model.cfg.x_dim                                   # landmark features per frame
        self.D = model.cfg.in_dim if self.cache else model.cfg.x_dim  # width of a ring row
        self.pred_every = pred_every
        self.warmup_min = min(10, max_t) if warmup_min is None else warmup_min
        dev = self.device = torch.device(device)
        self.ring_x = torch.zeros(n_streams, max_t, self.D, device=dev)
        self.ring_r = (torch.zeros(n_streams, max_t, roi_hw[0], roi_hw[1], device=dev, dtype=torch.uint8)
                       if roi_hw and not self.cache else None)
        self.head = torch.zeros(n_streams, device=dev, dtype=torch.int32)
        self.count = torch.zeros(n_streams, device=dev, dtype=torch.int32)
        self.frames_seen = torch.zeros(n_streams, device=dev, dtype=torch.int32)
        self.ema = torch.zeros(n_streams, device=dev, dtype=torch.float64)  # a Python float in the reference
        self.mouth_open = torch.zeros(n_streams, device=dev, dtype=torch.uint8)
        self.front: Optional[LiveFrontEnd] = None  # attach_front_end(): landmarks + camera frames instead of feature rows
        # host mirrors of the two counters decide who is due without reading the device back
        self._count = np.zeros(n_streams, np.int64)
        self._seen = np.zeros(n_streams, np.int64)

    def push(self, stream_ids: Sequence[int], feats: torch.Tensor, rois: Optional[torch.Tensor] = None,
             openness: Optional[torch.Tensor] = None):
        """One new frame for each of the (distinct) streams in ``stream_ids``.  Returns None, or
        ``(ids, logits, lengths)`` for the streams that are due for a prediction this tick."""
        ids = np.asarray(stream_ids, np.int32)
        n = len(ids)
        if len(set(ids.tolist())) != n:
            raise ValueError("one frame per stream and tick")
        dev = self.device
        ids_d = torch.from_numpy(ids).to(dev)
        feats = feats.to(dev, torch.float32).contiguous()
        rois = rois.to(dev).contiguous() if rois is not None else None
        if self.cache:  # the row the ring keeps: features | embedding of this frame (made now, once)
            if rois is None or tuple(rois.shape) != (n, *self.roi_hw) or tuple(feats.shape) != (n, self.Dx):
                raise ValueError(f"feats must be ({n}, {self.Dx}) and rois uint8 ({n}, {self.roi_hw[0]}, {self.roi_hw[1]})")
            rows = torch.empty(n, self.D, device=dev)
            L.call("ss_copy_rows_f32", feats.data_ptr(), self.Dx, rows.data_ptr(), self.D, n, self.Dx, L.stream())
            self.model.embed_rois(rois, out=rows[:, self.Dx:], ld_out=self.D)
            feats, rois = rows, None
        fb = self.roi_hw[0] * self.roi_hw[1] if (self.roi_hw and not self.cache) else 0
        L.call("ss_ring_push", self.ring_x.data_ptr(), L.ptr(self.ring_r), self.S, self.max_t, self.D, fb, ids_d.data_ptr(), n,
               feats.data_ptr(), L.ptr(rois) if self.ring_r is not None else None, self.head.data_ptr(), self.count.data_ptr(),
               self.frames_seen.data_ptr(), L.stream())
        if openness is not None:
            op = openness.to(dev, torch.float64).contiguous()
            L.call("ss_mouth_gate", ids_d.data_ptr(), n, op.data_ptr(), EMA_ALPHA, OPEN_THR, CLOSE_THR, self.ema.data_ptr(),
                   self.mouth_open.data_ptr(), L.stream())
        self._count[ids] = np.minimum(self._count[ids] + 1, self.max_t)
        self._seen[ids] += 1
        due = ids[(self._count[ids] >= self.warmup_min) & (self._seen[ids] % self.pred_every == 0)]
        if len(due) == 0:
            return None
        X, T, R = self.windows(due)
        with torch.no_grad():
            logits = self.model.forward_embedded(X, T) if self.cache else self.model(X, T, R if self.model.use_roi else None)
        return due, logits, T

    def skip(self, stream_ids: Sequence[int]) -> None:
        """A camera frame without a face for each of the given streams: their frame counters run on -- the reference's rule
        "every PRED_EVERY-th CAMERA frame" (inactive/live_feed.py:173, 201) counts those frames too --, nothing is buffered, nobody
        predicts (:179-185 ``continue``)."""
        ids = np.asarray(stream_ids, np.int32)
        if len(ids) == 0:
            return
        if len(set(ids.tolist())) != len(ids):
            raise ValueError("one frame per stream and tick")
        L.call("ss_ring_tick", torch.from_numpy(ids).to(self.device).data_ptr(), len(ids), self.frames_seen.data_ptr(), L.stream())
        self._seen[ids] += 1

    def attach_front_end(self, idxs: Sequence[int], frame_wh, variant: str = "live", band=(None, None)) -> "LiveFrontEnd":
        """Landmarks + camera frames in, instead of ready-made feature rows: see ``push_landmarks``."""
        if 2 * len(idxs) + 4 != self.Dx:
            raise ValueError(f"the model takes {self.Dx} features per frame, {len(idxs)} landmarks give {2 * len(idxs) + 4}")
        self.front = LiveFrontEnd(self.S, idxs, frame_wh, self.roi_hw, variant, band, self.device)
        return self.front

    def push_landmarks(self, stream_ids: Sequence[int], lm: torch.Tensor, frames_bgr: Optional[torch.Tensor] = None,
                       openness: Optional[torch.Tensor] = None):
        """The live chain as one device entry point (live_infer_official.py:264-296 per stream, then live_feed.py's window rule):
        distance gate -> feature fuse with per-stream velocity state -> crop box -> gray + resize -> ring push of the KEPT
        frames -> one forward over the streams that are due.  Returns ``(kept (n,) bool, result)`` with ``result`` as ``push``."""
        if self.front is None:
            raise RuntimeError("call attach_front_end(idxs, frame_wh) first")
        kept, X, rois = self.front(stream_ids, lm, frames_bgr)
        if not kept.any():
            return kept, None
        ids = np.asarray(stream_ids, np.int32)[kept]
        sel = torch.from_numpy(np.flatnonzero(kept)).to(self.device)
        op = openness.to(self.device).index_select(0, sel) if openness is not None else None
        res = self.push(ids, X.index_select(0, sel), rois.index_select(0, sel) if rois is not None else None, op)
        return kept, res

    def windows(self, ids: np.ndarray):
        """Zero-padded windows (oldest frame first) of the given streams: X (n,max_t,D), lengths (n,), R or None.  With cached
        embeddings a row of X is torch.cat((features, embedding)) and there is no R."""
        dev, n, mt = self.device, len(ids), self.max_t
        ids_d = torch.from_numpy(np.asarray(ids, np.int32)).to(dev)
        fmap = torch.empty(n, mt, device=dev, dtype=torch.int32)
        T = torch.empty(n, device=dev, dtype=torch.int64)
        L.call("ss_ring_window_map", ids_d.data_ptr(), n, mt, self.head.data_ptr(), self.count.data_ptr(), fmap.data_ptr(),
               T.data_ptr(), L.stream())
        X = torch.empty(n, mt, self.D, device=dev)
        L.call("ss_batch_gather_f32", self.ring_x.data_ptr(), self.D, fmap.data_ptr(), n * mt, None, None, 0.0, 0, X.data_ptr(),
               L.stream())
        R = None
        if self.ring_r is not None:
            H, W = self.roi_hw
            R = torch.empty(n, mt, H, W, device=dev, dtype=torch.uint8)
            L.call("ss_batch_gather_u8", self.ring_r.data_ptr(), H * W, fmap.data_ptr(), n * mt, R.data_ptr(), L.stream())
        X._ss_keep = (ids_d, fmap)
        return X, T, R


class ClipGateServer:
    """Openness-gated clip segmentation of S streams + batched classification of the clips that end
    (inactive/live_test_5.py:233-272, one state machine per camera there).  ``push`` takes one frame for each of the
    given streams -- feature rows, optional ROI frames, the openness value (``mouth_openness``) -- advances every
    stream's machine in one launch (``ss_clip_gate``), and runs ONE forward over the clips that ended in this tick with
    at least ``min_clip`` frames (lengths = clip lengths, so the padding never reaches the recurrence)."""

    def __init__(self, model: BiGRUClassifier, n_streams: int, roi_hw=None, open_thresh: float = OPEN_THRESH,
                 start_n: int = START_N, end_n: int = END_N, max_clip: int = MAX_CLIP, min_clip: int = MIN_CLIP, device="cuda"):
        L.load()
        self.model, self.S, self.roi_hw = model.eval(), n_streams, roi_hw
        self.open_thresh, self.start_n, self.end_n, self.max_clip, self.min_clip = open_thresh, start_n, end_n, max_clip, min_clip
        self.D = model.cfg.x_dim
        dev = self.device = torch.device(device)
        self.state = torch.zeros(n_streams, 4, device=dev, dtype=torch.int32)  # speaking, above_ct, below_ct, clip_len
        self.clip_x = torch.zeros(n_streams, max_clip, self.D, device=dev)
        self.clip_r = torch.zeros(n_streams, max_clip, roi_hw[0], roi_hw[1], device=dev, dtype=torch.uint8) if roi_hw else None

    def push(self, stream_ids: Sequence[int], feats: torch.Tensor, openness: torch.Tensor, rois: Optional[torch.Tensor] = None,
             face_present: Optional[torch.Tensor] = None):
        """-> (append_row (n,) int32, emit_len (n,) int32, result) with result None or (ids, logits, lengths) of the clips
        that ended in this tick."""
        ids = np.asarray(stream_ids, np.int32)
        n = len(ids)
        if len(set(ids.tolist())) != n:
            raise ValueError("one frame per stream and tick")
        if n and (ids.min() < 0 or ids.max() >= self.S):
            raise ValueError(f"stream ids must be in [0, {self.S})")
        # the kernel indexes feats[i*D + d], rois[i*H*W + c], openness[i], face_present[i] for i < n: a mis-shaped argument
        # would be an out-of-bounds device read or a garbage clip, not an error -- turn it away here
        if tuple(feats.shape) != (n, self.D):
            raise ValueError(f"feats must be ({n}, {self.D}), got {tuple(feats.shape)}")
        if openness.numel() != n:
            raise ValueError(f"openness must hold {n} values, got {openness.numel()}")
        if face_present is not None and face_present.numel() != n:
            raise ValueError(f"face_present must hold {n} values, got {face_present.numel()}")
        if rois is not None and self.clip_r is not None:
            if rois.dtype != torch.uint8 or tuple(rois.shape) != (n, *self.roi_hw):
                raise ValueError(f"rois must be uint8 ({n}, {self.roi_hw[0]}, {self.roi_hw[1]}), got {rois.dtype} {tuple(rois.shape)}")
        dev = self.device
        ids_d = torch.from_numpy(ids).to(dev)
        feats = feats.to(dev, torch.float32).contiguous()
        op = openness.to(dev, torch.float64).contiguous()
        rois = rois.to(dev).contiguous() if (rois is not None and self.clip_r is not None) else None
        face = face_present.to(dev, torch.uint8).contiguous() if face_present is not None else None
        fb = self.roi_hw[0] * self.roi_hw[1] if self.roi_hw else 0
        row = torch.empty(n, device=dev, dtype=torch.int32)
        emit = torch.empty(n, device=dev, dtype=torch.int32)
        L.call("ss_clip_gate", ids_d.data_ptr(), n, op.data_ptr(), L.ptr(face), float(self.open_thresh), self.start_n, self.end_n,
               self.max_clip, self.min_clip, self.state.data_ptr(), self.D, fb, feats.data_ptr(), L.ptr(rois),
               self.clip_x.data_ptr(), L.ptr(self.clip_r) if rois is not None else None, row.data_ptr(), emit.data_ptr(), L.stream())
        emit_h = emit.cpu().numpy()  # the one read-back of a tick: which clips ended
        done = np.flatnonzero(emit_h > 0)
        if len(done) == 0:
            return row, emit, None
        sel = torch.from_numpy(ids[done].astype(np.int64)).to(dev)
        X = self.clip_x.index_select(0, sel)
        T = torch.from_numpy(emit_h[done].astype(np.int64)).to(dev)
        # rows past a clip's length still hold older clips' frames: lengths keep them out of the recurrence and the pool
        R = self.clip_r.index_select(0, sel) if (self.clip_r is not None and self.model.use_roi) else None
        with torch.no_grad():
            logits = self.model(X, T, R)
        return row, emit, (ids[done], logits, T)
