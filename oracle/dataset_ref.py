"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's clip dataset and batch collation.

Restates ``clip_pad_trim`` / ``roi_pad_trim`` / ``NPZWordDataset.__getitem__`` / ``collate_fn`` of
/root/reference/train_model_official.py:93-204 in NumPy so that the device batch assembly
(``silent_speech_amd.device_data.DeviceClipStore``, csrc/batch.hip) can be compared with it bit for bit.  The random draws
are made with the same generators, in the same order and with the same arguments as the reference makes them (that order
is what makes batches reproducible under a seed); everything else is this repo's wording.  Pinned by
``tests/golden/dataset.npz`` (batches the reference's own classes produced, ``make_golden.py:gen_dataset``).
Only ``tests/`` may import this module; the product path is ``DeviceClipStore``."""
from __future__ import annotations

import random

import numpy as np
import torch

from silent_speech_amd.data import DROP_FRAMES_MAX, DROP_FRAMES_PROB, MAX_T, NOISE_STD


def fit_length(arr: np.ndarray, n: int, max_t: int, dtype):
    """First ``max_t`` frames, or the ``n`` frames followed by zero frames up to ``max_t``; -> (array, frames that count)
    (train...:93-120: one rule for features and ROI frames)."""
    if n >= max_t:
        return arr[:max_t], max_t
    padded = np.zeros((max_t,) + arr.shape[1:], dtype=dtype)
    padded[:n] = arr
    return padded, n


def augment_features(feats: np.ndarray) -> np.ndarray:
    """train...:143-152.  Draw order: python ``random`` decides (noise? p = 0.7), NumPy draws the noise; python ``random``
    decides (drop? p = DROP_FRAMES_PROB, only for clips longer than 12 frames) and how many, NumPy picks which interior
    frames.  Frames are dropped from the FEATURES only -- the ROI frames keep their count."""
    n = feats.shape[0]
    if random.random() < 0.7:
        feats = feats + np.random.normal(0, NOISE_STD, size=feats.shape).astype(np.float32)
    if n > 12 and random.random() < DROP_FRAMES_PROB:
        how_many = random.randint(1, DROP_FRAMES_MAX)
        gone = np.random.choice(np.arange(1, n - 1), size=how_many, replace=False)
        feats = np.delete(feats, gone, axis=0)
    return feats


class ClipDatasetRef(torch.utils.data.Dataset):
    """One item = (features (max_t, D) f32, frames that count, ROI (max_t, H, W) u8 or None, label id) -- train...:122-172."""

    def __init__(self, files, label_to_id, max_t: int = MAX_T, augment: bool = False, use_roi: bool = True):
        self.files, self.label_to_id = list(files), label_to_id
        self.max_t, self.augment, self.use_roi = max_t, augment, use_roi

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        clip = np.load(self.files[i], allow_pickle=True)
        feats = clip["X"].astype(np.float32)
        label = int(self.label_to_id[str(clip["label"])])
        if self.augment:
            feats = augment_features(feats)
        feats, n = fit_length(feats, int(feats.shape[0]), self.max_t, np.float32)
        if not (self.use_roi and "roi" in clip.files):
            return torch.from_numpy(feats), torch.tensor(n), None, torch.tensor(label)
        roi = clip["roi"]
        n = min(n, int(roi.shape[0]), self.max_t)  # both streams are cut to the shorter one, then padded again
        feats, _ = fit_length(feats[:n], n, self.max_t, np.float32)
        roi, _ = fit_length(roi[:n], n, self.max_t, np.uint8)
        return torch.from_numpy(feats), torch.tensor(n), torch.from_numpy(roi), torch.tensor(label)


def collate_ref(items, roi_hw=(48, 96)):
    """-> X (B, max_t, D) f32, T (B,) i64, R (B, max_t, H, W) u8 or None, y (B,) i64 (train...:174-204).  A clip without ROI
    frames gets zero frames when any clip of the batch has them (the reference hard-codes 48x96 for those zeros; here the size
    follows the clips that do have ROI frames, ``roi_hw`` only if none can tell)."""
    feats, lens, rois, labels = zip(*items)
    X = torch.stack(feats, 0)
    T = torch.stack(lens, 0).long()
    y = torch.stack(labels, 0).long()
    have = [r for r in rois if r is not None]
    if not have:
        return X, T, None, y
    hw = tuple(have[0].shape[1:]) or roi_hw
    blank = torch.zeros((X.shape[1],) + hw, dtype=torch.uint8)
    return X, T, torch.stack([blank if r is None else r for r in rois], 0), y
