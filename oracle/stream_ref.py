"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's sliding-window loop and mouth gate.

Follows /root/reference/inactive/live_feed.py:155 (``deque(maxlen=max_t)``), :163-164 (``PRED_EVERY = 2``,
``WARMUP_MIN = min(10, max_t)``), :201-207 (predict when ``len(buf) >= WARMUP_MIN and frame_idx % PRED_EVERY == 0`` on the
buffer zero-padded to ``(max_t, D)``) and /root/reference/important_landmarks.py:57-61, :136-144 (EMA with
``EMA_ALPHA = 0.25``, open/close hysteresis at 0.02 / 0.02).  The reference runs this per camera inside its capture loop,
which cannot be called; parity of this file is therefore by restatement only (PARITY UNPINNED), the model forward it
feeds is pinned through oracle/model_ref.py.

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
