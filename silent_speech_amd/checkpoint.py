"""``.pt`` checkpoint round trip with the reference scripts.

Writer: the dict of /root/reference/train_model_official.py:489-500.  Reader: ``load_classifier`` of
/root/reference/live_infer_official.py:198-221 (accepts the optional ``gru_layers`` key, hard-codes roi_emb=32 and
hidden=192 like the reference).  ``topk_from_logits``: live_infer_official.py:223-226.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

from .model import BiGRUClassifier


def save_checkpoint(path: str, model: BiGRUClassifier, labels: Sequence[str], max_t: int = 90, roi_w: int = 96,
                    roi_h: int = 48, seed: int = 42) -> None:
    labels = list(labels)
    label_to_id = {lab: i for i, lab in enumerate(labels)}
    torch.save({
        "model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "x_dim": model.cfg.x_dim, "max_t": max_t, "use_roi": bool(model.use_roi), "roi_w": roi_w, "roi_h": roi_h,
        "labels": labels, "label_to_id": label_to_id, "id_to_label": {i: lab for lab, i in label_to_id.items()},
        "seed": seed, "gru_layers": model.cfg.gru_layers,
    }, path)


def load_classifier(path: str, device="cuda", roi_standardize: bool = True):
    """-> (model.eval() on ``device``, id_to_label, max_t, use_roi).  ``roi_standardize=False`` reproduces the live
    script's own forward (which skips the per-frame standardisation, SURVEY.md note N2)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    x_dim, max_t = int(ckpt["x_dim"]), int(ckpt["max_t"])
    use_roi = bool(ckpt.get("use_roi", False))
    labels = ckpt["labels"]
    model = BiGRUClassifier(x_dim=x_dim, num_classes=len(labels), use_roi=use_roi, roi_emb=32, hidden=192,
                            gru_layers=int(ckpt.get("gru_layers", 2)), roi_standardize=roi_standardize)
    model.load_state_dict(ckpt["model"])
    model.to(device).eval()
    return model, ckpt["id_to_label"], max_t, use_roi


def softmax_topk(logits: torch.Tensor, k: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    """``ss_softmax_topk`` over a (B, C) batch of logits on the device: -> (probs (B,k) f32, class ids (B,k) int32),
    most probable first.  What the serving path returns for thousands of windows instead of raw logits."""
    from . import _lib as L

    if not logits.is_cuda:
        raise RuntimeError("softmax_topk runs on the HIP device (there is no CPU path)")
    lg = logits.detach().float().contiguous().reshape(-1, logits.shape[-1])
    B, C = lg.shape
    probs = torch.empty(B, k, device=lg.device, dtype=torch.float32)
    idx = torch.empty(B, k, device=lg.device, dtype=torch.int32)
    L.call("ss_softmax_topk", lg.data_ptr(), B, C, k, probs.data_ptr(), idx.data_ptr(), L.stream())
    return probs, idx


def topk_from_logits(logits: torch.Tensor, id_to_label: Dict[int, str], k: int = 3) -> List[Tuple[str, float]]:
    """live_infer_official.py:223-226 for one clip's logits (1, C): [(label, probability)] of the k most probable."""
    probs, idx = softmax_topk(logits.reshape(1, -1), min(k, logits.numel()))
    return [(id_to_label[int(i)], float(p)) for p, i in zip(probs[0].cpu().tolist(), idx[0].cpu().tolist())]
