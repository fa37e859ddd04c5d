"""silent_speech_amd -- MI355X (gfx950) native hot path of the Silent-Speech lip-reading classifier.

Drop-in surface of the reference's per-clip path:
``BiGRUClassifier`` (same constructor, ``forward(X, lengths, R)``, state_dict keys), the fused
``Trainer`` step, the landmark feature fuse / crop-box helpers and the ``.pt`` / ``.npz`` formats.
All arithmetic runs in ``libss_hotpath.so`` (hand-written HIP, C ABI in ``include/ss_hotpath.h``);
there is no CPU or PyTorch-op fallback.
"""
from . import checkpoint, data, features, harness, serving
from .checkpoint import load_classifier, save_checkpoint, topk_from_logits
from .device_data import DeviceClipStore
from .engine import Config
from .features import crop_boxes, crop_rois, extract_features
from .infer import GraphedInference
from .serving import ClipGateServer, LiveFrontEnd, StreamServer, mouth_openness
from .model import AttnPool, BiGRUClassifier, TinyROICNN
from .train import Trainer, allreduce_flat_grads, shard_range

__all__ = ["Config", "BiGRUClassifier", "TinyROICNN", "AttnPool", "Trainer", "allreduce_flat_grads", "shard_range",
           "extract_features", "crop_boxes", "GraphedInference", "load_classifier", "save_checkpoint", "topk_from_logits", "features",
           "data", "checkpoint"]
