"""``.npz`` clip dataset and collate with the reference's padding / augmentation rules
(/root/reference/train_model_official.py:93-204).  Host side (NumPy): it feeds the device path and is not part of
the timed region.  Schema: X (T,D) f32, ts, label, speaker, idxs, optional roi (T,H,W) u8 (train…:1-9)."""
from __future__ import annotations

import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

MAX_T = 90
NOISE_STD = 0.01
DROP_FRAMES_PROB = 0.35
DROP_FRAMES_MAX = 2


def clip_pad_trim(X: np.ndarray, T: int, max_t: int) -> Tuple[np.ndarray, int]:
    """Trim to max_t or zero-pad to (max_t, D); returns the effective length (train…:93-103)."""
    if T >= max_t:
        return X[:max_t], max_t
    out = np.zeros((max_t, X.shape[1]), dtype=np.float32)
    out[:T] = X
    return out, T


def roi_pad_trim(R: np.ndarray, T: int, max_t: int) -> Tuple[np.ndarray, int]:
    if T >= max_t:
        return R[:max_t], max_t
    out = np.zeros((max_t,) + R.shape[1:], dtype=np.uint8)
    out[:T] = R
    return out, T


def save_clip(path: str, X: np.ndarray, ts, label: str, speaker: str, idxs, roi: Optional[np.ndarray] = None) -> None:
    """Writer with the recorder's keys and its X/roi length alignment (record_landmarks_official.py:237-251)."""
    save = dict(X=np.asarray(X, np.float32), ts=np.asarray(ts), label=label, speaker=speaker, idxs=np.asarray(idxs))
    if roi is not None and len(roi):
        T = min(len(save["X"]), len(roi))
        save["X"] = save["X"][:T]
        save["roi"] = np.asarray(roi[:T], np.uint8)
    np.savez_compressed(path, **save)


class NPZWordDataset(torch.utils.data.Dataset):
    def __init__(self, files: Sequence[str], label_to_id, max_t: int = MAX_T, augment: bool = False, use_roi: bool = True):
        self.files, self.label_to_id = list(files), label_to_id
        self.max_t, self.augment, self.use_roi = max_t, augment, use_roi

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        d = np.load(self.files[idx], allow_pickle=True)
        X = d["X"].astype(np.float32)
        T = int(X.shape[0])
        y = int(self.label_to_id[str(d["label"])])
        if self.augment:  # same draws, same order as train…:143-152 (noise p=.7; 1-2 interior frames dropped from X only)
            if random.random() < 0.7:
                X = X + np.random.normal(0, NOISE_STD, size=X.shape).astype(np.float32)
            if T > 12 and random.random() < DROP_FRAMES_PROB:
                k = random.randint(1, DROP_FRAMES_MAX)
                drop = np.random.choice(np.arange(1, T - 1), size=k, replace=False)
                keep = np.ones(T, dtype=bool)
                keep[drop] = False
                X = X[keep]
                T = int(X.shape[0])
        X, T_eff = clip_pad_trim(X, T, self.max_t)
        if ("roi" in d.files) and self.use_roi:
            R = d["roi"]
            T_use = min(T_eff, int(R.shape[0]), self.max_t)
            X_pad, T_use = clip_pad_trim(X[:T_use], T_use, self.max_t)
            R_pad, _ = roi_pad_trim(R[:T_use], T_use, self.max_t)
            return torch.from_numpy(X_pad), torch.tensor(T_use), torch.from_numpy(R_pad), torch.tensor(y)
        return torch.from_numpy(X), torch.tensor(T_eff), None, torch.tensor(y)


def collate_fn(batch: List, roi_hw: Tuple[int, int] = (48, 96)):
    """-> X (B,max_t,D) f32, T (B,) i64, R (B,max_t,H,W) u8 or None, y (B,) i64; clips lacking roi get zeros
    (train…:174-204; the reference hard-codes 48x96 there, here the size follows the clips that do have roi)."""
    Xs, Ts, Rs, ys = zip(*batch)
    X = torch.stack(Xs, 0)
    T = torch.stack(Ts, 0).long()
    y = torch.stack(ys, 0).long()
    if any(r is not None for r in Rs):
        hw = next(tuple(r.shape[1:]) for r in Rs if r is not None) or roi_hw
        R = torch.stack([r if r is not None else torch.zeros((X.shape[1],) + hw, dtype=torch.uint8) for r in Rs], 0)
    else:
        R = None
    return X, T, R, y
