"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's sliding-window loop and mouth gate.

Follows /root/reference/inactive/live_feed.py:155 (``deque(maxlen=max_t)``), :163-164 (``PRED_EVERY = 2``,
``WARMUP_MIN = min(10, max_t)``), :201-207 (predict when ``len(buf) >= WARMUP_MIN and frame_idx % PRED_EVERY == 0`` on the
buffer zero-padded to ``(max_t, D)``) and /root/reference/important_landmarks.py:57-61, :136-144 (EMA with
``EMA_ALPHA = 0.25``, open/close hysteresis at 0.02 / 0.02).  The reference runs this per camera inside its capture loops, which cannot be called.  PINNED all the same (round 4): the
statements of those loops were taken out of the parsed scripts and executed frame by frame on seeded traces
(tests/golden/make_golden.py:gen_serving_loops -> tests/golden/serving_loops.npz), and the three restatements here -- ``StreamRef``'s
EMA + hysteresis, ``ClipGateRef``, ``StreamRef``'s window rule incl. camera frames without a face (``tick``) -- are held to what the
reference's statements produced (tests/test_oracle_golden.py).  The SIGNALS they consume are pinned too -- ``openness_eye_span``,
``openness_y_range``, ``face_to_xvec`` and ``feat83_and_openness`` against the reference's own ``dist2d`` / ``compute_openness`` /
``face_to_xvec`` / ``extract_83_and_openness`` on seeded faces (tests/golden/serving.npz) -- and the model forward is pinned
through oracle/model_ref.py.

``openness_eye_span`` / ``openness_y_range`` restate important_landmarks.py:64-67, 131-133 and
inactive/live_test_5.py:92-94 literally (Python floats = float64, ``** 0.5``); ``ClipGateRef`` restates the clip-gating
state machine of inactive/live_test_5.py:146-152, 233-272 and its "NO FACE" reset (:293-301) statement by statement.
"""
from __future__ import annotations

from collections import deque
from typing import Dict, List, Optional

import numpy as np

PRED_EVERY, EMA_ALPHA, OPEN_THR, CLOSE_THR = 2, 0.25, 0.02, 0.02


class StreamRef:
    def __init__(self, max_t: int, D: int, roi_hw=None):
        self.max_t, self.D, self.roi_hw = max_t, D, roi_hw
        self.buf: deque = deque(maxlen=max_t)
        self.rbuf: deque = deque(maxlen=max_t)
        self.frame_idx = 0
        self.ema = 0.0  # ``mouth_ema = 0.0``: a Python float (float64), important_landmarks.py:109
        self.open = False

    def tick(self) -> None:
        """A camera frame without a face: live_feed.py:173 has counted it, :179-185 ``continue`` in front of the buffer."""
        self.frame_idx += 1

    def push(self, feat: np.ndarray, roi: Optional[np.ndarray] = None, openness: Optional[float] = None) -> Optional[Dict]:
        """One frame; returns the zero-padded window when this stream is due for a prediction."""
        self.frame_idx += 1
        self.buf.append(np.asarray(feat, np.float32))
        if roi is not None:
            self.rbuf.append(np.asarray(roi, np.uint8))
        if openness is not None:
            # important_landmarks.py:136-144, in Python floats exactly as written there
            self.ema = (1 - EMA_ALPHA) * self.ema + EMA_ALPHA * float(openness)
            if self.open:
                if self.ema < CLOSE_THR:
                    self.open = False
            else:
                if self.ema > OPEN_THR:
                    self.open = True
        if len(self.buf) >= min(10, self.max_t) and self.frame_idx % PRED_EVERY == 0:
            t = len(self.buf)
            X = np.zeros((self.max_t, self.D), np.float32)
            X[:t] = np.stack(list(self.buf), 0)
            out = {"X": X, "T": t}
            if self.roi_hw is not None:
                R = np.zeros((self.max_t,) + tuple(self.roi_hw), np.uint8)
                R[:t] = np.stack(list(self.rbuf), 0)
                out["R"] = R
            return out
        return None


# ------------------------------------------------------------------ openness signals
MOUTH_TOP, MOUTH_BOTTOM, LEFT_EYE_CORNER, RIGHT_EYE_CORNER = 13, 14, 33, 263  # important_landmarks.py:49-54


def openness_eye_span(face_xy, i_top=MOUTH_TOP, i_bot=MOUTH_BOTTOM, i_l=LEFT_EYE_CORNER, i_r=RIGHT_EYE_CORNER) -> float:
    """important_landmarks.py:131-133 with dist2d (:64-67); ``face_xy[i] = (x, y)`` as Python floats."""
    lip_gap = abs(float(face_xy[i_bot][1]) - float(face_xy[i_top][1]))
    dx = float(face_xy[i_l][0]) - float(face_xy[i_r][0])
    dy = float(face_xy[i_l][1]) - float(face_xy[i_r][1])
    eye_span = (dx * dx + dy * dy) ** 0.5 + 1e-6
    return lip_gap / eye_span


def openness_y_range(face_xy) -> float:
    """inactive/live_test_5.py:92-94 over the landmarks given."""
    ys = [float(p[1]) for p in face_xy]
    return float(max(ys) - min(ys))


def face_to_xvec(face_xy, idxs, with_openness: bool) -> np.ndarray:
    """inactive/live_test_5.py:96-112: float32 x / y of the clip's landmarks, each centred on its own mean, interleaved
    (x0, y0, x1, y1, ...); optionally one more entry, the y range of the CENTRED float32 values."""
    pts = np.asarray([[float(face_xy[i][0]), float(face_xy[i][1])] for i in idxs], np.float64)
    xs, ys = pts[:, 0].astype(np.float32), pts[:, 1].astype(np.float32)
    xs, ys = xs - xs.mean(), ys - ys.mean()
    v = np.stack([xs, ys], 1).reshape(-1).astype(np.float32)
    if with_openness:
        v = np.concatenate([v, np.asarray([float(ys.max() - ys.min())], np.float32)])
    return v


LIP_ORDER_LIVE_FEED = (185, 40, 39, 37, 0, 267, 269, 270, 409, 415, 310, 311, 312, 13, 82, 81, 42, 183, 78,
                       61, 146, 91, 181, 84, 17, 314, 405, 321, 375, 291, 308, 324, 318, 402, 317, 14, 87, 178, 88, 95)


def _norm2_f32(v) -> np.float32:
    """np.linalg.norm of a float32 2-vector: both squares, their sum and the root rounded to float32 one by one."""
    a, b = np.float32(v[0]), np.float32(v[1])
    return np.sqrt(np.float32(np.float32(a * a) + np.float32(b * b)))


def feat83_and_openness(lm: np.ndarray, lip_order=LIP_ORDER_LIVE_FEED):
    """inactive/live_feed.py:57-86 on a (478, 2) float32 landmark array: the 40 lip points centred on their centroid and
    divided by the mouth width (|lm[291] - lm[61]| + 1e-6), then openness = |lm[13] - lm[14]| / width,
    height = |lm[0] - lm[17]| / width and corner = |lm[61] - lm[291]| / width - 1 -- all in float32."""
    lm = np.asarray(lm, np.float32)
    pts = lm[list(lip_order)]
    width = _norm2_f32(lm[291] - lm[61]) + np.float32(1e-6)
    feat80 = ((pts - pts.mean(axis=0, keepdims=True)) / width).reshape(-1).astype(np.float32)
    openness = float(_norm2_f32(lm[13] - lm[14]) / width)
    height = float(_norm2_f32(lm[0] - lm[17]) / width)
    corner = float(_norm2_f32(lm[61] - lm[291]) / width) - 1.0
    return np.concatenate([feat80, np.asarray([openness, height, corner], np.float32)]), openness


# ------------------------------------------------------------------ clip gating (inactive/live_test_5.py:146-152, 233-272)
OPEN_THRESH, START_N, END_N, MAX_CLIP, MIN_CLIP = 0.18, 3, 5, 60, 6


class ClipGateRef:
    def __init__(self, open_thresh=OPEN_THRESH, start_n=START_N, end_n=END_N, max_clip=MAX_CLIP, min_clip=MIN_CLIP):
        self.open_thresh, self.start_n, self.end_n, self.max_clip, self.min_clip = open_thresh, start_n, end_n, max_clip, min_clip
        self.speaking = False
        self.above_ct = 0
        self.below_ct = 0
        self.clip_buf: List[np.ndarray] = []

    def push(self, openv: float, xvec: np.ndarray, face: bool = True):
        """One frame -> (appended, finished clip (t, D) or None)."""
        if not face:  # :293-301
            self.speaking = False
            self.above_ct = self.below_ct = 0
            self.clip_buf = []
            return False, None
        if openv > self.open_thresh:
            self.above_ct += 1
            self.below_ct = 0
        else:
            self.below_ct += 1
            self.above_ct = 0
        appended, done = False, None
        if not self.speaking:
            if self.above_ct >= self.start_n:
                self.speaking = True
                self.clip_buf = []
                self.above_ct = 0
                self.below_ct = 0
        else:
            self.clip_buf.append(np.asarray(xvec, np.float32))
            appended = True
            if self.below_ct >= self.end_n or len(self.clip_buf) >= self.max_clip:
                self.speaking = False
                self.above_ct = 0
                self.below_ct = 0
                if len(self.clip_buf) >= self.min_clip:
                    done = np.stack(self.clip_buf).astype(np.float32)
        return appended, done
