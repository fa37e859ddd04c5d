"""Training harness around the fused step: what ``main()`` of the reference does around its inner loop.

Counterparts, by reference line (/root/reference/train_model_official.py):
  scan_clips        :315-373   inventory of a clip directory, majority feature width, label tables, ROI decision
  split_by_label    :52-77     per-label validation split, reproducible from a seed
  class_balanced_indices :382-397   the WeightedRandomSampler draw (1 / class count, with replacement)
  evaluate          :449-475   loss / accuracy / predictions over a validation set, forward only
  top_confusions    :79-91     "actual→predicted(count)" strings of the most frequent errors
  fit               :417-506   epochs, save-best checkpoint (:486-500), early stopping (:501-505)

The clips live in a ``DeviceClipStore`` (uploaded once), batches are assembled on the device, the step is
``Trainer.step``; nothing here touches the arithmetic of the hot path.
"""
from __future__ import annotations

import collections
import glob
import os
import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .checkpoint import save_checkpoint
from .device_data import DeviceClipStore
from .model import BiGRUClassifier
from .train import Trainer

VAL_FRAC, SEED, PATIENCE, EPOCHS, BATCH_SIZE = 0.15, 42, 12, 80, 16


def scan_clips(clip_dir: str):
    """-> dict(files, labels, x_dim, uniq, label_to_id, id_to_label, has_roi).  Clips whose feature width is not the most
    common one are left out, as the reference does (:347-358)."""
    files = sorted(glob.glob(os.path.join(clip_dir, "*.npz")))
    if not files:
        raise RuntimeError(f"No .npz files found in {clip_dir}")
    labels, dims, has_roi = [], [], 0
    for f in files:
        d = np.load(f, allow_pickle=True)
        labels.append(str(d["label"]))
        dims.append(int(d["X"].shape[1]))
        has_roi += int("roi" in d.files)
    x_dim = collections.Counter(dims).most_common(1)[0][0]
    keep = [k for k, dm in enumerate(dims) if dm == x_dim]
    files, labels = [files[k] for k in keep], [labels[k] for k in keep]
    uniq = sorted(set(labels))
    label_to_id = {lab: i for i, lab in enumerate(uniq)}
    return dict(files=files, labels=labels, x_dim=x_dim, uniq=uniq, label_to_id=label_to_id,
                id_to_label={i: lab for lab, i in label_to_id.items()}, has_roi=has_roi)


def split_by_label(files: Sequence[str], labels: Sequence[str], val_frac: float = VAL_FRAC, seed: int = SEED,
                   verbose: bool = False) -> Tuple[List[str], List[str]]:
    """Every label gives ``round(n * val_frac)`` clips (at least 1, at most n-1) to validation; one ``random.Random(seed)``
    drives, in this order, the shuffle inside each label (labels in first-seen order) and the two final shuffles."""
    rng = random.Random(seed)
    groups: Dict[str, List[str]] = {}
    for f, lab in zip(files, labels):
        groups.setdefault(lab, []).append(f)
    train: List[str] = []
    val: List[str] = []
    for lab, members in groups.items():
        rng.shuffle(members)
        n = len(members)
        n_val = min(max(1, int(round(n * val_frac))), n - 1)
        val += members[:n_val]
        train += members[n_val:]
        if verbose:
            print(f"{lab:>10}: total={n:4d}  train={n - n_val:4d}  val={n_val:4d}")
    rng.shuffle(train)
    rng.shuffle(val)
    return train, val


def top_confusions(y_true: Sequence[int], y_pred: Sequence[int], id_to_label: Dict[int, str], k: int = 8) -> List[str]:
    wrong = collections.Counter((t, p) for t, p in zip(y_true, y_pred) if t != p)
    return [f"{id_to_label[t]}→{id_to_label[p]}({n})" for (t, p), n in wrong.most_common(k)]


def class_balanced_indices(labels: Sequence[str], num_samples: Optional[int] = None,
                           generator: Optional[torch.Generator] = None) -> List[int]:
    """One epoch of the reference's WeightedRandomSampler: weight 1 / count(label), drawn with replacement."""
    counts = collections.Counter(labels)
    w = torch.tensor([1.0 / counts[lab] for lab in labels], dtype=torch.double)
    n = len(labels) if num_samples is None else num_samples
    return torch.multinomial(w, n, replacement=True, generator=generator).tolist()


@torch.no_grad()
def evaluate(model: BiGRUClassifier, store: DeviceClipStore, batch_size: int = BATCH_SIZE, label_smoothing: float = 0.05):
    """-> (mean loss, accuracy, y_true, y_pred) over every clip of ``store``, in order, eval mode, no augmentation."""
    from . import _lib as L
    from .checkpoint import softmax_topk

    was_training = model.training
    model.eval()
    y_true, y_pred = [], []
    dev = model.flat_params.device
    loss_sum = torch.zeros(1, device=dev, dtype=torch.float32)   # sum of the per-clip losses (denom = 1)
    correct = torch.zeros(1, device=dev, dtype=torch.int32)
    for lo in range(0, len(store), batch_size):
        idx = list(range(lo, min(len(store), lo + batch_size)))
        X, T, R, y = store.batch(idx, augment=False)
        logits = model(X, T, R if model.use_roi else None).contiguous()
        y = y.to(torch.int64).contiguous()
        # loss and hit count by the path's own cross-entropy kernel, predictions by its top-k kernel (no aten op)
        L.call("ss_ce_ls_fwd_bwd", logits.data_ptr(), y.data_ptr(), logits.shape[0], logits.shape[1], label_smoothing, 1.0,
               None, loss_sum.data_ptr(), correct.data_ptr(), L.stream())
        _, top = softmax_topk(logits, 1)
        y_true += y.cpu().tolist()
        y_pred += top[:, 0].cpu().tolist()
    model.train(was_training)
    model.check_health()  # the loop above has synchronised anyway
    n = max(1, len(store))
    return float(loss_sum) / n, int(correct) / n, y_true, y_pred


def fit(clip_dir: str, out_path: str, epochs: int = EPOCHS, batch_size: int = BATCH_SIZE, patience: int = PATIENCE,
        max_t: int = 90, lr: float = 3e-4, seed: int = SEED, use_roi_if_present: bool = True, device="cuda",
        log=print) -> float:
    """The reference's ``main()``: scan, split, train with class-balanced sampling and on-device augmentation, evaluate
    every epoch, keep the best checkpoint (reference schema), stop after ``patience`` epochs without improvement."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    info = scan_clips(clip_dir)
    train_files, val_files = split_by_label(info["files"], info["labels"], VAL_FRAC, seed=seed)
    use_roi = use_roi_if_present and info["has_roi"] > 0
    train_store = DeviceClipStore(train_files, info["label_to_id"], max_t=max_t, use_roi=use_roi, device=device)
    val_store = DeviceClipStore(val_files, info["label_to_id"], max_t=max_t, use_roi=use_roi, device=device)
    train_labels = [str(np.load(f, allow_pickle=True)["label"]) for f in train_files]
    model = BiGRUClassifier(info["x_dim"], len(info["uniq"]), use_roi=use_roi, roi_emb=32, hidden=192).to(device).train()
    trainer = Trainer(model, lr=lr)
    roi_hw = train_store.roi_hw or (48, 96)
    gen = np.random.default_rng(seed)
    best, bad = 0.0, 0
    for ep in range(1, epochs + 1):
        order = class_balanced_indices(train_labels)
        tr_loss = torch.zeros((), device=device)
        tr_ok = torch.zeros((), device=device, dtype=torch.int64)
        for lo in range(0, len(order), batch_size):
            idx = order[lo:lo + batch_size]
            X, T, R, y = train_store.batch(idx, augment=True, rng="device", generator=gen)
            loss, correct = trainer.step(X, T, R if use_roi else None, y)
            tr_loss += loss * len(idx)
            tr_ok += correct
        va_loss, va_acc, y_true, y_pred = evaluate(model, val_store, batch_size)
        confs = top_confusions(y_true, y_pred, info["id_to_label"], k=6)
        n = max(1, len(order))
        log(f"ep {ep:02d} | train loss {float(tr_loss) / n:.4f} acc {int(tr_ok) / n:.3f} | val loss {va_loss:.4f} acc {va_acc:.3f}"
            + ((" | top confusions: " + ", ".join(confs)) if confs else ""))
        if va_acc > best:
            best, bad = va_acc, 0
            save_checkpoint(out_path, model, info["uniq"], max_t=max_t, roi_w=roi_hw[1], roi_h=roi_hw[0], seed=seed)
            log(f"  saved {out_path} (best val acc {best:.3f})")
        else:
            bad += 1
            if bad >= patience:
                log(f"Early stopping. Best val acc: {best:.3f}")
                break
    return best
