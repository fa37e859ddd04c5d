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


def topk_from_logits(logits: torch.Tensor, id_to_label: Dict[int, str], k: int = 3) -> List[Tuple[str, float]]:
    probs = torch.softmax(logits.detach().float().reshape(-1), dim=-1)
    order = torch.argsort(probs, descending=True)[:k].cpu().tolist()
    probs = probs.cpu()
    return [(id_to_label[int(i)], float(probs[i])) for i in order]
