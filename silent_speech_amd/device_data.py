"""Clips resident in HBM, batches assembled on the device.

Counterpart of ``NPZWordDataset`` + ``collate_fn`` (/root/reference/train_model_official.py:122-204) for the case the
reference cannot afford: every clip of the training set is uploaded ONCE (a ragged frame store: features
``(sum T, D)`` f32, ROI frames ``(sum Tr, H, W)`` u8) and a batch is two gather launches (``ss_batch_gather_f32`` /
``ss_batch_gather_u8``) driven by a ``(B, max_t)`` frame map.  The map encodes the reference's rules: noise on the
features with probability 0.7, one or two interior frames dropped from the FEATURES only (train...:146-152; the ROI
frames are not dropped), trim / zero-pad to ``max_t``, lengths aligned to ``min(T, Tr, max_t)``.

Two sources of randomness:
  * ``rng="reference"``: the host makes the draws with exactly the calls, order and distributions of the reference's
    ``__getitem__`` (``random.random``, ``np.random.normal``, ``random.randint``, ``np.random.choice``), so a batch is
    bit-identical to ``collate_fn([dataset[i] for i in indices])`` under the same seeds (tests/golden/dataset.npz);
  * ``rng="device"``: the decisions come from a ``numpy.random.Generator`` and the noise itself from the Philox stream
    inside the gather kernel -- nothing but two small index maps crosses PCIe.
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .data import DROP_FRAMES_MAX, DROP_FRAMES_PROB, MAX_T, NOISE_STD


class DeviceClipStore:
    def __init__(self, files: Sequence[str], label_to_id, max_t: int = MAX_T, use_roi: bool = True, device="cuda"):
        L.load()
        self.max_t, self.device = max_t, torch.device(device)
        xs, rs, self.x_off, self.x_len, self.r_off, self.r_len, ys = [], [], [], [], [], [], []
        xo = ro = 0
        self.roi_hw = None
        for f in files:
            d = np.load(f, allow_pickle=True)
            X = d["X"].astype(np.float32)
            xs.append(X)
            self.x_off.append(xo)
            self.x_len.append(len(X))
            xo += len(X)
            ys.append(int(label_to_id[str(d["label"])]))
            if use_roi and "roi" in d.files:
                R = np.asarray(d["roi"], np.uint8)
                self.roi_hw = self.roi_hw or tuple(R.shape[1:])
                rs.append(R)
                self.r_off.append(ro)
                self.r_len.append(len(R))
                ro += len(R)
            else:
                self.r_off.append(-1)
                self.r_len.append(0)
        self.D = xs[0].shape[1]
        self.X = torch.from_numpy(np.concatenate(xs, 0)).to(self.device)
        self.R = torch.from_numpy(np.concatenate(rs, 0)).to(self.device) if rs else None
        if self.R is not None and (self.roi_hw[0] * self.roi_hw[1]) % 16:
            raise ValueError("ROI frames must be a multiple of 16 bytes")
        self.y = torch.tensor(ys, dtype=torch.int64, device=self.device)

    def __len__(self):
        return len(self.x_len)

    # ------------------------------------------------------------------ decisions (host, a few integers per clip)
    def _plan(self, indices, augment, rng, gen):
        """Per clip: kept feature frames, whether noise is added (and, in reference mode, the noise itself)."""
        keeps, noises = [], []
        for i in indices:
            T = self.x_len[i]
            keep = np.arange(T)
            noise = None
            if augment:
                if rng == "reference":  # the reference's draws, in its order (train...:143-152)
                    if random.random() < 0.7:
                        noise = np.random.normal(0, NOISE_STD, size=(T, self.D)).astype(np.float32)
                    if T > 12 and random.random() < DROP_FRAMES_PROB:
                        k = random.randint(1, DROP_FRAMES_MAX)
                        drop = np.random.choice(np.arange(1, T - 1), size=k, replace=False)
                        m = np.ones(T, dtype=bool)
                        m[drop] = False
                        keep = keep[m]
                else:
                    if gen.random() < 0.7:
                        noise = True
                    if T > 12 and gen.random() < DROP_FRAMES_PROB:
                        k = int(gen.integers(1, DROP_FRAMES_MAX + 1))
                        drop = gen.choice(np.arange(1, T - 1), size=k, replace=False)
                        m = np.ones(T, dtype=bool)
                        m[drop] = False
                        keep = keep[m]
            keeps.append(keep)
            noises.append(noise)
        return keeps, noises

    def batch(self, indices: Sequence[int], augment: bool = False, rng: str = "device",
              generator: Optional[np.random.Generator] = None, seed: int = 0):
        """-> X (B,max_t,D) f32, T (B,) i64, R (B,max_t,H,W) u8 or None, y (B,) i64 -- all on the device."""
        indices = list(indices)
        B, mt = len(indices), self.max_t
        gen = generator or np.random.default_rng(seed)
        keeps, noises = self._plan(indices, augment, rng, gen)
        xmap = np.full((B, mt), -1, np.int32)
        nmap = np.full((B, mt), -1, np.int32)
        rmap = np.full((B, mt), -1, np.int32)
        lens = np.zeros(B, np.int64)
        host_noise: List[np.ndarray] = []
        n_rows = 0
        any_roi = self.R is not None and any(self.r_off[i] >= 0 for i in indices)
        for b, i in enumerate(indices):
            keep = keeps[b]
            t_eff = min(len(keep), mt)                                   # clip_pad_trim
            if any_roi and self.r_off[i] >= 0:
                t_eff = min(t_eff, self.r_len[i], mt)                     # T_use = min(T_eff, Tr, max_t)
                rmap[b, :t_eff] = self.r_off[i] + np.arange(t_eff)        # ROI frames are NOT dropped
            src = keep[:t_eff]
            xmap[b, :t_eff] = self.x_off[i] + src
            if noises[b] is not None:
                if rng == "reference":
                    nmap[b, :t_eff] = n_rows + src                       # noise was drawn for the undropped clip
                    host_noise.append(noises[b])
                    n_rows += self.x_len[i]
                else:
                    nmap[b, :t_eff] = 0
            lens[b] = t_eff
        dev = self.device
        xmap_d, nmap_d = torch.from_numpy(xmap).to(dev), torch.from_numpy(nmap).to(dev)
        X = torch.empty(B, mt, self.D, device=dev)
        noise_d = torch.from_numpy(np.concatenate(host_noise, 0)).to(dev) if host_noise else None
        use_noise = augment and (noise_d is not None or rng != "reference")
        L.call("ss_batch_gather_f32", self.X.data_ptr(), self.D, xmap_d.data_ptr(), B * mt, L.ptr(noise_d),
               nmap_d.data_ptr() if use_noise else None, float(NOISE_STD) if (use_noise and noise_d is None) else 0.0,
               int(gen.integers(0, 2 ** 62)) if rng != "reference" else 0, X.data_ptr(), L.stream())
        R = None
        if any_roi:
            H, W = self.roi_hw
            rmap_d = torch.from_numpy(rmap).to(dev)
            R = torch.empty(B, mt, H, W, device=dev, dtype=torch.uint8)
            L.call("ss_batch_gather_u8", self.R.data_ptr(), H * W, rmap_d.data_ptr(), B * mt, R.data_ptr(), L.stream())
            # keep the maps alive until the launches have consumed them
            R._ss_keep = (rmap_d,)
        X._ss_keep = (xmap_d, nmap_d, noise_d)
        return X, torch.from_numpy(lens).to(dev), R, self.y[torch.as_tensor(indices, device=dev)]
