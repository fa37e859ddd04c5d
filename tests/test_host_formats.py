"""CPU: .npz clip dataset / collate and .pt checkpoint round trip (host logic, no GPU)."""
import os
import random

import numpy as np
import pytest
import torch

import silent_speech_amd as ss
from oracle import dataset_ref as DR
from silent_speech_amd import data as D


def _clip(tmp, name, T, Dm=180, roi=True, label="yes"):
    rng = np.random.default_rng(abs(hash(name)) % 2**32)
    X = rng.normal(size=(T, Dm)).astype(np.float32)
    R = rng.integers(0, 256, (T + 1, 48, 96), dtype=np.uint8) if roi else None
    p = os.path.join(tmp, name + ".npz")
    D.save_clip(p, X, np.arange(T), label, "me", np.arange(88), R)
    return p, X


def test_npz_schema_and_pad_trim(tmp_path):
    tmp = str(tmp_path)
    p1, X1 = _clip(tmp, "a", 20)
    p2, X2 = _clip(tmp, "b", 120, label="no")
    p3, X3 = _clip(tmp, "c", 7, roi=False)
    d = np.load(p1, allow_pickle=True)
    assert set(d.files) == {"X", "ts", "label", "speaker", "idxs", "roi"} and str(d["label"]) == "yes"
    assert d["roi"].shape == (20, 48, 96) and d["roi"].dtype == np.uint8  # trimmed to len(X), record…:245-248
    ds = DR.ClipDatasetRef([p1, p2, p3], {"yes": 0, "no": 1}, max_t=90, augment=False)
    X, T, R, y = ds[0]
    assert X.shape == (90, 180) and int(T) == 20 and R.shape == (90, 48, 96) and int(y) == 0
    assert torch.equal(X[:20], torch.from_numpy(X1)) and float(X[20:].abs().sum()) == 0 and int(R[20:].sum()) == 0
    X, T, R, y = ds[1]
    assert int(T) == 90 and torch.equal(X, torch.from_numpy(X2[:90])) and int(y) == 1
    Xb, Tb, Rb, yb = DR.collate_ref([ds[0], ds[1], ds[2]])
    assert Xb.shape == (3, 90, 180) and Tb.tolist() == [20, 90, 7] and Rb.shape == (3, 90, 48, 96) and Rb.dtype == torch.uint8
    assert int(Rb[2].sum()) == 0 and Tb.dtype == torch.int64 and yb.dtype == torch.int64
    Xn, Tn, Rn, yn = DR.collate_ref([ds[2]])
    assert Rn is None


def test_augmentation_follows_reference_rules(tmp_path):
    p, X0 = _clip(str(tmp_path), "a", 30)
    ds = DR.ClipDatasetRef([p], {"yes": 0}, augment=True)
    random.seed(0)
    np.random.seed(0)
    lens = set()
    for _ in range(40):
        X, T, R, y = ds[0]
        lens.add(int(T))
        assert 28 <= int(T) <= 30 and torch.equal(X[0, :0], X[0, :0])
    assert lens == {28, 29, 30}  # 0, 1 or 2 interior frames dropped


def test_checkpoint_round_trip(tmp_path):
    m = ss.BiGRUClassifier(180, 3, use_roi=True)
    path = str(tmp_path / "word_model_points_roi.pt")
    ss.save_checkpoint(path, m, ["aura", "no", "yes"])
    ck = torch.load(path, map_location="cpu", weights_only=False)
    for key in ("model", "x_dim", "max_t", "use_roi", "roi_w", "roi_h", "labels", "label_to_id", "id_to_label", "seed"):
        assert key in ck  # train_model_official.py:489-500
    assert ck["id_to_label"] == {0: "aura", 1: "no", 2: "yes"}
    m2, id_to_label, max_t, use_roi = ss.load_classifier(path, device="cpu")
    assert max_t == 90 and use_roi and not m2.training
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # softmax + top-k is a HIP kernel now (a12): like the rest of the path it refuses CPU tensors instead of falling back
    with pytest.raises(RuntimeError, match="HIP device"):
        ss.topk_from_logits(torch.tensor([[0.1, 2.0, -1.0]]), id_to_label, k=3)


def _golden_clips(tmp, golden_dir):
    """Re-creates the clip files of tests/golden/dataset.npz (made by the reference's own dataset code)."""
    d = np.load(os.path.join(golden_dir, "dataset.npz"), allow_pickle=True)
    files = []
    for k in range(int(d["n_clips"])):
        X = d[f"clip{k}::X"]
        roi = d[f"clip{k}::roi"] if f"clip{k}::roi" in d.files else None
        save = dict(X=X, ts=np.arange(len(X)), label=str(d[f"clip{k}::label"]), speaker="me", idxs=np.arange(4))
        if roi is not None:
            save["roi"] = roi
        p = os.path.join(tmp, f"c{k}.npz")
        np.savez_compressed(p, **save)
        files.append(p)
    return d, files


def test_dataset_and_collate_reproduce_the_reference_batches(tmp_path):
    """Same seeds, same visiting order -> the batches the reference's NPZWordDataset(augment=True) + collate_fn built."""
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d, files = _golden_clips(str(tmp_path), golden_dir)
    ds = DR.ClipDatasetRef(files, {"no": 0, "yes": 1}, max_t=int(d["max_t"]), augment=True)
    for b in range(int(d["n_batches"])):
        random.seed(1000 + b)
        np.random.seed(2000 + b)
        Xb, Tb, Rb, yb = DR.collate_ref([ds[i] for i in d[f"batch{b}::order"]], roi_hw=(16, 16))
        assert np.array_equal(Xb.numpy(), d[f"batch{b}::X"]) and np.array_equal(Tb.numpy(), d[f"batch{b}::T"])
        assert np.array_equal(Rb.numpy(), d[f"batch{b}::R"]) and np.array_equal(yb.numpy(), d[f"batch{b}::y"])


def test_harness_helpers_match_the_reference():
    """split_by_label / top_confusions against what the reference's own functions returned (tests/golden/harness.npz)."""
    from silent_speech_amd import harness as Hn

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness.npz"), allow_pickle=True)
    files, labels = g["files"].tolist(), g["labels"].tolist()
    for seed in (42, 7):
        tr, va = Hn.split_by_label(list(files), list(labels), float(g[f"split{seed}::frac"]), seed=seed)
        assert tr == g[f"split{seed}::train"].tolist() and va == g[f"split{seed}::val"].tolist()
        assert sorted(tr + va) == sorted(files) and not any(f.endswith("_solo.npz") for f in va)
    id_to_label = {0: "aura", 1: "go", 2: "no", 3: "stop", 4: "yes"}
    for k in (3, 6, 8):
        assert Hn.top_confusions(g["conf::y_true"].tolist(), g["conf::y_pred"].tolist(), id_to_label, k=k) == g[f"conf::top{k}"].tolist()
    # the sampler: inverse class frequency, with replacement -> rare classes are drawn about as often as common ones
    idx = Hn.class_balanced_indices(labels, 20000, generator=torch.Generator().manual_seed(0))
    drawn = np.bincount([sorted(set(labels)).index(labels[i]) for i in idx], minlength=6) / 20000.0
    assert np.all(np.abs(drawn - 1.0 / 6.0) < 0.02), drawn
