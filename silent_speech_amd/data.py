"""The ``.npz`` clip format: its writer and the augmentation constants of /root/reference/train_model_official.py:29-45.
Schema: X (T,D) f32, ts, label, speaker, idxs, optional roi (T,H,W) u8 (train…:1-9).  Batches are assembled on the device
(``device_data.DeviceClipStore``); the CPU restatement of the reference's dataset / collate classes it is tested against
lives under ``oracle/dataset_ref.py`` (test infrastructure)."""
from __future__ import annotations

from typing import Optional

import numpy as np

MAX_T = 90
NOISE_STD = 0.01
DROP_FRAMES_PROB = 0.35
DROP_FRAMES_MAX = 2


def save_clip(path: str, X: np.ndarray, ts, label: str, speaker: str, idxs, roi: Optional[np.ndarray] = None) -> None:
    """Writer with the recorder's keys and its X/roi length alignment (record_landmarks_official.py:237-251)."""
    save = dict(X=np.asarray(X, np.float32), ts=np.asarray(ts), label=label, speaker=speaker, idxs=np.asarray(idxs))
    if roi is not None and len(roi):
        T = min(len(save["X"]), len(roi))
        save["X"] = save["X"][:T]
        save["roi"] = np.asarray(roi[:T], np.uint8)
    np.savez_compressed(path, **save)
