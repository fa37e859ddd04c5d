"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's sliding-window loop and mouth gate.

Follows /root/reference/inactive/live_feed.py:155 (``deque(maxlen=max_t)``), :163-164 (``PRED_EVERY = 2``,
``WARMUP_MIN = min(10, max_t)``), :201-207 (predict when ``len(buf) >= WARMUP_MIN and frame_idx % PRED_EVERY == 0`` on the
buffer zero-padded to ``(max_t, D)``) and /root/reference/important_landmarks.py:57-61, :136-144 (EMA with
``EMA_ALPHA = 0.25``, open/close hysteresis at 0.02 / 0.02).  The reference runs this per camera inside its capture loop,
which cannot be called; parity of this file is therefore by restatement only (unpinned), the model forward it feeds is
pinned through oracle/model_ref.py.
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
        self.ema = np.float32(0.0)
        self.open = False

    def push(self, feat: np.ndarray, roi: Optional[np.ndarray] = None, openness: Optional[float] = None) -> Optional[Dict]:
        """One frame; returns the zero-padded window when this stream is due for a prediction."""
        self.frame_idx += 1
        self.buf.append(np.asarray(feat, np.float32))
        if roi is not None:
            self.rbuf.append(np.asarray(roi, np.uint8))
        if openness is not None:
            a = np.float32(EMA_ALPHA)
            self.ema = np.float32(np.float32((np.float32(1.0) - a) * self.ema) + np.float32(a * np.float32(openness)))
            if self.open:
                if self.ema < np.float32(CLOSE_THR):
                    self.open = False
            elif self.ema > np.float32(OPEN_THR):
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
