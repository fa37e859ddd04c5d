#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code in the authoring container.

    cd /root/repo && python tests/golden/make_golden.py

Needs /root/reference (read-only) and therefore never runs on the GPU box; the committed .npz
files are what travel.  Nothing of the reference's source is stored: fixtures hold inputs,
seeds and the numbers the reference computed.

* ``train_model_official`` imports cleanly (numpy/torch only).
* ``record_landmarks_official`` / ``live_infer_official`` import cv2 and mediapipe at module top;
  neither is installed here, so empty placeholder modules are registered for the *import*
  only.  The functions sampled (mouth_width_px, extract_feature, crop_roi*) are plain
  NumPy/Python arithmetic; crop_roi*'s two cv2 calls receive ``frame[y1:y2, x1:x2]`` of a frame
  whose pixels encode their own coordinates, which is how the integer crop box is read back
  without OpenCV.  OpenCV's gray/resize arithmetic itself is NOT sampled (parity unpinned).
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import weights as W  # noqa: E402

REF = "/root/reference"


def import_reference():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "inactive"))
    import train_model_official as tmo

    for name in ("cv2", "mediapipe", "mediapipe.tasks", "mediapipe.tasks.python", "mediapipe.tasks.python.vision"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mediapipe"].tasks = sys.modules["mediapipe.tasks"]
    sys.modules["mediapipe.tasks"].python = sys.modules["mediapipe.tasks.python"]
    sys.modules["mediapipe.tasks.python"].vision = sys.modules["mediapipe.tasks.python.vision"]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:  # recorder does os.makedirs("clips_npz") at import
        os.chdir(tmp)
        try:
            import record_landmarks_official as rec
            import live_infer_official as live
            import train_reduced as tred
        finally:
            os.chdir(cwd)
    return tmo, rec, live, tred


# ------------------------------------------------------------------ model cases
MODEL_CASES = [
    # name, cls ('train'|'live'), x_dim, C, roi_hw, B, T, lengths, layers
    ("model_lm_only", "train", 84, 5, None, 4, 12, [12, 7, 1, 12], 2),
    ("model_roi64", "train", 84, 5, (64, 64), 3, 8, [8, 5, 2], 2),
    ("model_shipped", "train", 180, 10, (48, 96), 2, 10, [10, 6], 2),
    ("model_live_l1", "live", 84, 5, (64, 64), 2, 6, [6, 3], 1),
    ("model_live_l2", "live", 84, 5, (64, 64), 2, 6, [4, 6], 2),
    # the reference's shipped shape at its shipped length: MAX_T = 90 frames, 88 landmarks -> D = 180, ROI 48 x 96, 10 words
    # (train_model_official.py:29-38); one full-length clip and one ragged one
    ("model_shipped_t90", "train", 180, 10, (48, 96), 2, 90, [90, 53], 2),
]


def gen_model_case(tmo, live, name, cls, x_dim, C, roi_hw, B, T, lengths, layers, seed):
    use_roi = roi_hw is not None
    sd = W.make_state_dict(seed, x_dim, C, use_roi, gru_layers=layers)
    X, L, R, y = W.make_inputs(seed, B, T, x_dim, C, roi_hw, lengths)
    if use_roi:
        # exercise the std clamp (train_model_official.py:290): constant and near-constant frames
        R[0, 0] = 0
        if T > 1:
            R[0, 1] = 200
        if T > 2:
            R[0, 2] = 17
            R[0, 2, 3, 5] = 18
    if cls == "train":
        m = tmo.BiGRUClassifier(x_dim, C, use_roi=use_roi, roi_emb=32, hidden=192)
    else:
        m = live.BiGRUClassifier(x_dim, C, use_roi=use_roi, roi_emb=32, hidden=192, gru_layers=layers)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), (list(ref_sd.keys()), list(sd.keys()))
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    m.eval()
    out = dict(seed=seed, x_dim=x_dim, num_classes=C, use_roi=int(use_roi), layers=layers, cls=cls,
               X=X.numpy(), lengths=L.numpy(), y=y.numpy(), sd_checksum=W.checksum(sd))
    if use_roi:
        out["R"] = R.numpy()
    hooks = {}

    def roi_hook(mod, i, o):
        hooks.setdefault("roi_e", o.detach().clone())

    def pool_hook(mod, i, o):
        hooks.setdefault("gru_out", i[0].detach().clone())
        hooks.setdefault("pooled", o.detach().clone())

    if use_roi:
        m.roi_cnn.register_forward_hook(roi_hook)
    m.pool.register_forward_hook(pool_hook)
    logits = m(X, L, R)
    out["logits"] = logits.detach().numpy()
    for k, v in hooks.items():
        out[k] = v.numpy()
    if cls == "train":
        # eval-mode gradients (both dropouts off) + one clip/Adam step, train_model_official.py:433-439
        loss_fn = torch.nn.CrossEntropyLoss(label_smoothing=0.05)
        opt = torch.optim.Adam(m.parameters(), lr=3e-4)
        loss = loss_fn(logits, y)
        opt.zero_grad()
        loss.backward()
        out["loss"] = float(loss)
        for k, p in m.named_parameters():
            out["grad::" + k] = W.reduce_tensor(p.grad)
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        out["total_norm"] = float(tn)
        opt.step()
        for k, p in m.named_parameters():
            out["step1::" + k] = W.reduce_tensor(p.detach())
        # second step on the same batch pins the Adam state recursion
        logits2 = m(X, L, R)
        loss2 = loss_fn(logits2, y)
        opt.zero_grad()
        loss2.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        out["loss2"] = float(loss2)
        out["logits_after2"] = m(X, L, R).detach().numpy()
        _check_against_the_loop_statements(tmo, sd, x_dim, C, use_roi, X, L, R, y, m, float(loss2))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "logits", out["logits"][0][:3])


# ------------------------------------------------------------------ real-weights KAT
def gen_kat(tred):
    ck = torch.load(os.path.join(REF, "inactive", "word_model_5.pt"), map_location="cpu", weights_only=True)
    sd = ck["model"] if "model" in ck else ck
    in_dim = sd["gru.weight_ih_l0"].shape[1]
    hid = sd["gru.weight_hh_l0"].shape[1]
    C = sd["head.0.weight"].shape[0]
    m = tred.GRUClassifier(in_dim, C, hidden=hid)
    m.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 20, in_dim, generator=g) * 0.5
    out, _ = m.gru(x)
    logits = m(x)
    save = {"w::" + k: v.numpy() for k, v in sd.items()}
    save.update(x=x.numpy(), gru_out=out.detach().numpy(), logits=logits.detach().numpy())
    np.savez_compressed(os.path.join(HERE, "kat_word_model_5.npz"), **save)
    print("wrote kat_word_model_5", in_dim, hid, C)


# ------------------------------------------------------------------ features
class _P:
    __slots__ = ("x", "y")

    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)


def synth_faces(seed, T, w=640, h=480):
    """(T,478,2) float32 normalised landmarks; anchors placed so the mouth width is 60-150 px."""
    rng = np.random.default_rng(seed)
    base = np.stack([rng.uniform(0.3, 0.7, 478), rng.uniform(0.4, 0.8, 478)], 1)
    lm = base[None] + rng.normal(0, 0.004, (T, 478, 2))
    lm[:, 61] = [0.42, 0.61] + rng.normal(0, 0.003, (T, 2))
    lm[:, 291] = [0.58, 0.61] + rng.normal(0, 0.003, (T, 2))
    lm[:, 13] = [0.50, 0.59] + rng.normal(0, 0.003, (T, 2))
    lm[:, 14] = [0.50, 0.63] + rng.normal(0, 0.006, (T, 2))
    return lm.astype(np.float32)


def gen_features(rec, live):
    w, h = 640, 480
    T = 14
    lm = synth_faces(11, T, w, h)
    idx88 = list(rec.FIXED_IDXS)
    lip40 = sorted(set(rec.MOUTH_LOWER + rec.MOUTH_UPPER))
    out = dict(lm=lm, w=w, h=h, idx88=np.array(idx88), idx40=np.array(lip40))
    reset = np.zeros(T, bool)
    reset[0] = True
    reset[6] = True
    out["reset"] = reset
    for tag, idxs in (("88", idx88), ("40", lip40)):
        for vname, mod in (("record", rec), ("live", live)):
            feats, cens, fourth, mws = [], [], [], []
            prev = None
            for t in range(T):
                face = [_P(*lm[t, i]) for i in range(478)]
                if reset[t]:
                    prev = None
                feat, prev, c, s = mod.extract_feature(face, w, h, idxs, prev)
                feats.append(feat)
                cens.append(np.asarray(c))
                fourth.append(float(s))
                mws.append(mod.mouth_width_px(face, w, h))
            out[f"feat_{vname}_{tag}"] = np.stack(feats)
            out[f"center_{vname}_{tag}"] = np.stack(cens)
            out[f"fourth_{vname}_{tag}"] = np.array(fourth, np.float64)
            out[f"mw_{vname}_{tag}"] = np.array(mws, np.float64)
            assert feats[0].dtype == np.float32 and feats[0].shape == (2 * len(idxs) + 4,)
    np.savez_compressed(os.path.join(HERE, "features.npz"), **out)
    print("wrote features", out["feat_record_88"].shape)


# ------------------------------------------------------------------ crop boxes
def gen_crop(rec, live):
    cv2 = sys.modules["cv2"]
    seen = {}

    def cvt(roi, code=None):
        seen["x1"], seen["y1"] = int(roi[0, 0, 0]), int(roi[0, 0, 1])
        seen["hh"], seen["ww"] = roi.shape[:2]
        return roi

    cv2.cvtColor = cvt
    cv2.resize = lambda roi, size, interpolation=None: roi
    cv2.COLOR_BGR2GRAY = 6
    cv2.INTER_AREA = 3

    rng = np.random.default_rng(3)
    rows = []
    sizes = [(640, 480), (1280, 720), (320, 240)]
    for (w, h) in sizes:
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        frame = np.stack([xx, yy, xx * 0], -1).astype(np.int32)
        cases = []
        for _ in range(400):
            cases.append((rng.uniform(-40, w + 40), rng.uniform(-40, h + 40), rng.uniform(0.0, 160.0)))
        # borders, exact-integer edges, degenerate boxes
        for s in (0.0, 1e-6, 0.4, 0.5, 1.0, 60.000001, 100.0, 150.000001, 400.0, 2000.0):
            for cx, cy in ((0, 0), (w, h), (w / 2, h / 2), (1.0, h - 1.0), (w - 0.5, 0.5), (-5, -5), (w + 5, h + 5)):
                cases.append((cx, cy, s))
        for cx, cy, s in cases:
            c32 = np.array([cx, cy], np.float32)
            scale = float(s)
            for variant, fn in (("record", rec.crop_roi), ("live", live.crop_roi_gray)):
                seen.clear()
                r = fn(frame, c32, scale)
                if r is None:
                    rows.append((w, h, c32[0], c32[1], scale, 0 if variant == "record" else 1, 0, -1, -1, -1, -1))
                else:
                    rows.append((w, h, c32[0], c32[1], scale, 0 if variant == "record" else 1, 1,
                                 seen["x1"], seen["x1"] + seen["ww"], seen["y1"], seen["y1"] + seen["hh"]))
    rows = np.array(rows, np.float64)
    np.savez_compressed(os.path.join(HERE, "crop.npz"),
                        wh=rows[:, :2].astype(np.int32), center=rows[:, 2:4].astype(np.float32),
                        scale=rows[:, 4], variant=rows[:, 5].astype(np.int32), valid=rows[:, 6].astype(np.int32),
                        box=rows[:, 7:11].astype(np.int32))
    print("wrote crop", rows.shape, "valid frac", rows[:, 6].mean())


# ------------------------------------------------------------------ dataset / augmentation / collate
def gen_dataset(tmo):
    """Runs the reference's NPZWordDataset(augment=True) + collate_fn on synthetic clips under fixed seeds and stores the
    clips, the seeds, the visiting order and the batches it produced."""
    import random

    rng = np.random.default_rng(77)
    D, H, W, max_t = 12, 16, 16, 24
    specs = [(30, 31, "yes"), (13, 13, "no"), (40, 38, "yes"), (9, 9, "no"), (24, None, "yes"), (26, 20, "no")]  # (T, Tr, label)
    out = {"max_t": max_t, "n_clips": len(specs)}
    label_to_id = {"no": 0, "yes": 1}
    with tempfile.TemporaryDirectory() as tmp:
        files = []
        for k, (T, Tr, lab) in enumerate(specs):
            X = rng.normal(size=(T, D)).astype(np.float32)
            save = dict(X=X, ts=np.arange(T), label=lab, speaker="me", idxs=np.arange(4))
            if Tr is not None:
                save["roi"] = rng.integers(0, 256, (Tr, H, W), dtype=np.uint8)
                out[f"clip{k}::roi"] = save["roi"]
            f = os.path.join(tmp, f"c{k}.npz")
            np.savez_compressed(f, **save)
            files.append(f)
            out[f"clip{k}::X"], out[f"clip{k}::label"] = X, lab
        old_hw = (tmo.ROI_H, tmo.ROI_W)
        tmo.ROI_H, tmo.ROI_W = H, W  # collate_fn pads roi-less clips with zeros of the module-level size
        try:
            ds = tmo.NPZWordDataset(files, label_to_id, max_t=max_t, augment=True, use_roi=True)
            orders = [[0, 1, 2, 3, 4, 5], [5, 0, 2], [4, 4, 1, 3]]
            for b, order in enumerate(orders):
                random.seed(1000 + b)
                np.random.seed(2000 + b)
                Xb, Tb, Rb, yb = tmo.collate_fn([ds[i] for i in order])
                out[f"batch{b}::order"] = np.asarray(order)
                out[f"batch{b}::X"], out[f"batch{b}::T"] = Xb.numpy(), Tb.numpy()
                out[f"batch{b}::R"], out[f"batch{b}::y"] = Rb.numpy(), yb.numpy()
            ds.augment = False
            Xb, Tb, Rb, yb = tmo.collate_fn([ds[i] for i in range(len(files))])
            out["plain::X"], out["plain::T"], out["plain::R"], out["plain::y"] = Xb.numpy(), Tb.numpy(), Rb.numpy(), yb.numpy()
        finally:
            tmo.ROI_H, tmo.ROI_W = old_hw
    out["n_batches"] = 3
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("wrote dataset: lengths", [out[f"batch{b}::T"].tolist() for b in range(3)])


# ------------------------------------------------------------------ harness helpers
def gen_harness(tmo):
    import contextlib
    import io

    rng = np.random.default_rng(5)
    labels = [str(x) for x in rng.choice(["aura", "no", "yes", "stop", "go"], size=61, p=[0.4, 0.25, 0.2, 0.1, 0.05])]
    labels += ["solo"]  # a label with a single clip: max(1, ..) then min(.., n - 1) = 0 validation clips
    files = [f"clips_npz/{k:03d}_{lab}.npz" for k, lab in enumerate(labels)]
    out = {"files": np.asarray(files), "labels": np.asarray(labels)}
    for seed, frac in ((42, 0.15), (7, 0.3)):
        with contextlib.redirect_stdout(io.StringIO()):
            tr, va = tmo.split_by_label(list(files), list(labels), frac, seed=seed)
        out[f"split{seed}::train"], out[f"split{seed}::val"], out[f"split{seed}::frac"] = np.asarray(tr), np.asarray(va), frac
    y_true = rng.integers(0, 5, 200)
    y_pred = np.where(rng.random(200) < 0.6, y_true, rng.integers(0, 5, 200))
    id_to_label = {0: "aura", 1: "go", 2: "no", 3: "stop", 4: "yes"}
    out["conf::y_true"], out["conf::y_pred"] = y_true, y_pred
    for k in (3, 6, 8):
        out[f"conf::top{k}"] = np.asarray(tmo.top_confusions(y_true.tolist(), y_pred.tolist(), id_to_label, k=k))
    np.savez_compressed(os.path.join(HERE, "harness.npz"), **out)
    print("wrote harness: val sizes", len(out["split42::val"]), len(out["split7::val"]), "top3", out["conf::top3"])


# ------------------------------------------------------------------ checkpoint written by the build -> reference loader
def gen_loader(live):
    """A ``.pt`` written by the BUILD's ``save_checkpoint`` goes through the reference's own ``load_classifier``
    (live_infer_official.py:198-221, torch.load with its defaults) and ``topk_from_logits`` (:223-226): stores the logits
    the reference computes from that file on a fixed clip and its top-3, plus top-3 of a few free logit rows."""
    repo = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, repo)
    import silent_speech_amd as ss  # CPU instance: parameter container + checkpoint writer only

    seed, x_dim, C, hw, B, T = 31, 84, 7, (48, 96), 2, 9
    labels = ["aura", "go", "help", "no", "stop", "water", "yes"]
    out = dict(seed=seed, x_dim=x_dim, num_classes=C, B=B, T=T, labels=np.asarray(labels))
    for layers in (1, 2):
        sd = W.make_state_dict(seed + layers, x_dim, C, True, gru_layers=layers)
        m = ss.BiGRUClassifier(x_dim, C, use_roi=True, gru_layers=layers)
        m.load_state_dict(sd)
        X, L, R, _ = W.make_inputs(seed + layers, B, T, x_dim, C, hw, lengths=[T, 5])
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "word_model_points_roi.pt")
            ss.save_checkpoint(path, m, labels, max_t=90, roi_w=hw[1], roi_h=hw[0], seed=42)
            ref_model, id_to_label, max_t, use_roi = live.load_classifier(path)
        assert max_t == 90 and use_roi and [id_to_label[i] for i in range(C)] == labels
        with torch.no_grad():
            logits = ref_model(X, L, R)
        out[f"l{layers}::logits"] = logits.numpy()
        for b in range(B):
            top = live.topk_from_logits(logits[b:b + 1], id_to_label, k=3)
            out[f"l{layers}::top_labels{b}"] = np.asarray([t[0] for t in top])
            out[f"l{layers}::top_probs{b}"] = np.asarray([t[1] for t in top], np.float64)
    g = torch.Generator().manual_seed(9)
    free = torch.randn(16, C, generator=g) * 3.0
    id_to_label = {i: lab for i, lab in enumerate(labels)}
    out["free::logits"] = free.numpy()
    out["free::top_idx"] = np.asarray([[labels.index(t[0]) for t in live.topk_from_logits(free[b:b + 1], id_to_label, k=3)]
                                       for b in range(16)])
    out["free::top_probs"] = np.asarray([[t[1] for t in live.topk_from_logits(free[b:b + 1], id_to_label, k=3)]
                                         for b in range(16)], np.float64)
    np.savez_compressed(os.path.join(HERE, "loader.npz"), **out)
    print("wrote loader: top-3 of clip 0", out["l2::top_labels0"], out["l2::top_probs0"])


def _functions_of(path, wanted, constants=()):
    """The named top-level functions (and constants) of a reference script whose MODULE BODY cannot run here
    (inactive/live_feed.py loads a checkpoint, builds a MediaPipe landmarker and opens the camera at import).  The script is
    parsed, every top-level statement except imports, the named constant assignments and the named function definitions is
    dropped, and what is left -- the reference's own, unmodified code objects -- is executed in a fresh namespace with the
    same empty cv2 / mediapipe placeholders the other imports use.  Nothing of the source is kept: the fixture holds numbers."""
    import ast

    tree = ast.parse(open(path).read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            keep.append(node)
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in wanted:
            keep.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and t.id in constants for t in node.targets):
            keep.append(node)
    tree.body = keep
    ns = {"__name__": "reference_functions"}
    exec(compile(tree, path, "exec"), ns)
    missing = [w for w in list(wanted) + list(constants) if w not in ns]
    assert not missing, missing
    return ns


def gen_serving():
    """The serving SIGNALS the reference computes per frame, from its own functions on seeded synthetic faces:

    * ``important_landmarks.dist2d`` (:64-67) on the eye corners and the openness formed exactly as :131-133
      (``abs(face[MOUTH_BOTTOM].y - face[MOUTH_TOP].y) / (dist2d(...) + 1e-6)``, Python floats);
    * ``inactive/live_test_5.compute_openness`` (:92-94) and ``face_to_xvec`` (:96-112) over a clip's landmark indices;
    * ``inactive/live_feed.extract_83_and_openness`` (:57-86) on (478, 2) float32 landmark arrays.

    The EMA / hysteresis / clip-gating / sliding-window LOOPS live inline in the scripts' capture loops and cannot be called:
    oracle/stream_ref.py restates them (stated there)."""
    import important_landmarks as il
    import live_test_5 as lt5

    lf = _functions_of(os.path.join(REF, "inactive", "live_feed.py"), ["extract_83_and_openness"],
                       ["LOWER_LIPS", "UPPER_LIPS", "LIP_ORDER"])
    rs = np.random.RandomState(77)
    n_faces, n_lm = 96, 478
    lm = rs.uniform(0.2, 0.8, size=(n_faces, n_lm, 2)).astype(np.float32)
    # a talking mouth: inner lips a small, varying gap apart; eye corners ~0.25 apart; values within a few ulps of each other too
    gap = rs.uniform(0.0, 0.06, size=n_faces).astype(np.float32)
    lm[:, il.MOUTH_TOP, 1] = 0.60
    lm[:, il.MOUTH_BOTTOM, 1] = (np.float32(0.60) + gap).astype(np.float32)
    lm[:, il.LEFT_EYE_CORNER] = (np.array([0.38, 0.40], np.float32) + rs.normal(0, 0.01, (n_faces, 2))).astype(np.float32)
    lm[:, il.RIGHT_EYE_CORNER] = (np.array([0.63, 0.41], np.float32) + rs.normal(0, 0.01, (n_faces, 2))).astype(np.float32)
    lm[0, il.MOUTH_BOTTOM, 1] = lm[0, il.MOUTH_TOP, 1]            # closed mouth: openness 0
    lm[1, il.RIGHT_EYE_CORNER] = lm[1, il.LEFT_EYE_CORNER]        # degenerate span: the 1e-6 carries the division
    lm[2, il.MOUTH_BOTTOM, 1] = np.nextafter(lm[2, il.MOUTH_TOP, 1], np.float32(1))  # one float32 ulp of gap
    lm[3, 291] = lm[3, 61]                                        # live_feed: zero mouth width -> 1e-6

    class P:  # a landmark as MediaPipe hands it over: attributes x, y holding Python floats
        __slots__ = ("x", "y")

        def __init__(self, x, y):
            self.x, self.y = float(x), float(y)

    idxs = [int(i) for i in lt5_landmark_idxs()]
    out = dict(lm=lm, idxs=np.asarray(idxs, np.int32), mouth_top=il.MOUTH_TOP, mouth_bottom=il.MOUTH_BOTTOM,
               eye_l=il.LEFT_EYE_CORNER, eye_r=il.RIGHT_EYE_CORNER, lip_order=np.asarray(lf["LIP_ORDER"], np.int32))
    eye_span, open_eye, open_yr, xv41, xv40, f83, open83 = [], [], [], [], [], [], []
    for f in range(n_faces):
        face = [P(x, y) for x, y in lm[f]]
        d = il.dist2d(face[il.LEFT_EYE_CORNER], face[il.RIGHT_EYE_CORNER])
        eye_span.append(d)
        lip_gap = abs(face[il.MOUTH_BOTTOM].y - face[il.MOUTH_TOP].y)   # important_landmarks.py:131
        open_eye.append(lip_gap / (d + 1e-6))                           # :132-133
        open_yr.append(lt5.compute_openness(face, idxs))
        xv41.append(lt5.face_to_xvec(face, idxs, 2 * len(idxs) + 1))
        xv40.append(lt5.face_to_xvec(face, idxs, 2 * len(idxs)))
        a, b = lf["extract_83_and_openness"](lm[f])
        f83.append(a)
        open83.append(b)
    out.update(eye_span=np.asarray(eye_span, np.float64), openness_eye=np.asarray(open_eye, np.float64),
               openness_yrange=np.asarray(open_yr, np.float64), xvec_with_open=np.stack(xv41), xvec=np.stack(xv40),
               feat83=np.stack(f83), openness83=np.asarray(open83, np.float64))
    assert out["feat83"].dtype == np.float32 and out["xvec"].dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "serving.npz"), **out)
    print("wrote serving: openness_eye[:4]", out["openness_eye"][:4], "openness83[:4]", out["openness83"][:4])


def _statements_between(path, first, last, drop=None, list_has=None):
    """A run of CONSECUTIVE statements of one statement list anywhere in a reference script -- from the first statement for which
    ``first(node)`` holds through the first one at or behind it for which ``last(node)`` holds -- compiled as they stand.  This is how
    the state machines that live inline in the scripts' capture loops (``while True: ok, frame = cap.read() ...``) are run here:
    the loop cannot be called, its statements can be executed in a namespace that holds the loop's variables.  ``drop(node)``
    removes statements from the run (the overlay drawing calls between them); ``list_has`` picks the statement list when ``first``
    matches in several.  Nothing of the source is kept: the fixture holds the numbers the statements produced."""
    import ast

    tree = ast.parse(open(path).read(), filename=path)
    for node in ast.walk(tree):
        for field in ("body", "orelse", "finalbody"):
            stmts = getattr(node, field, None)
            if not isinstance(stmts, list) or (list_has is not None and not any(list_has(st) for st in stmts)):
                continue
            for i, st in enumerate(stmts):
                if first(st):
                    for j in range(i, len(stmts)):
                        if last(stmts[j]):
                            run = [x for x in stmts[i:j + 1] if drop is None or not drop(x)]
                            return compile(ast.Module(body=run, type_ignores=[]), path, "exec"), len(run)
    raise AssertionError("statements not found in " + path)


def _ast_preds():
    import ast

    def names(t):
        return {n.id for n in ast.walk(t) if isinstance(n, ast.Name)}

    def assigns(*ids):  # an assignment whose targets are exactly these names (``a = b = 0`` has two, ``a, b = f()`` one tuple)
        return lambda n: isinstance(n, ast.Assign) and set().union(*[names(t) for t in n.targets]) == set(ids)

    def aug(name):
        return lambda n: isinstance(n, ast.AugAssign) and isinstance(n.target, ast.Name) and n.target.id == name

    def if_on(*ids):  # an ``if`` whose test mentions all of these names
        return lambda n: isinstance(n, ast.If) and set(ids) <= names(n.test)

    def cv2_call(n):
        return isinstance(n, ast.Expr) and isinstance(n.value, ast.Call) and "cv2" in names(n.value.func)

    return assigns, aug, if_on, cv2_call


def gen_serving_loops():
    """The serving STATE MACHINES, run from the reference's own statements (round 4 pinned the signals, these are the
    loops around them).  Each lives inline in a capture loop, so the statements are taken out of the parsed script
    (``_statements_between``) and executed frame by frame in a namespace holding the loop's variables:

    * important_landmarks.py:130-144 -- lip gap / eye span, the EMA ``mouth_ema`` (a Python float) and the open / close
      hysteresis; trace with values a few ulps on either side of both thresholds;
    * inactive/live_test_5.py:229-272 (+ the NO FACE branch :293-301) -- openness-gated clip segmentation: above / below
      counters, START_N / END_N / MAX_CLIP, clips shorter than 6 frames dropped; the clip the prediction is made from;
    * inactive/live_feed.py:173, 197-207 -- the sliding window: frame counter, ``deque(maxlen=max_t)``, WARMUP_MIN / PRED_EVERY,
      the zero-padded window handed to the model (stored as the source frame of every row)."""
    import collections

    import important_landmarks as il
    import live_test_5 as lt5

    assigns, aug, if_on, cv2_call = _ast_preds()
    rs = np.random.RandomState(2024)
    out = {}

    class P:
        __slots__ = ("x", "y")

        def __init__(self, x, y):
            self.x, self.y = float(x), float(y)

    # ---- (1) EMA + hysteresis
    code, n1 = _statements_between(os.path.join(REF, "important_landmarks.py"), assigns("lip_gap"), if_on("mouth_state_open"))
    assert n1 == 5, n1  # lip_gap, eye_span, openness, mouth_ema, the hysteresis ``if``
    F = 360
    pts = np.zeros((F, 4, 2), np.float32)  # top, bottom, left eye corner, right eye corner
    pts[:, 0] = (0.5, 0.60)
    pts[:, 2] = np.array([0.38, 0.40], np.float32) + rs.normal(0, 0.004, (F, 2)).astype(np.float32)
    pts[:, 3] = np.array([0.63, 0.41], np.float32) + rs.normal(0, 0.004, (F, 2)).astype(np.float32)
    gap = np.abs(np.cumsum(rs.normal(0, 0.0016, F))).astype(np.float32) % np.float32(0.012)  # wanders through both thresholds
    ns = dict(vars(il))
    ns.update(mouth_ema=0.0, mouth_state_open=False)
    ema, state, opn = [], [], []
    for f in range(F):
        if f % 9 == 4:  # aim the EMA at a threshold: the float32 landmark nearest the solution leaves it ulps to one side or the other
            thr = il.OPEN_THR if not ns["mouth_state_open"] else il.CLOSE_THR
            span = float(np.hypot(float(pts[f, 2, 0]) - float(pts[f, 3, 0]), float(pts[f, 2, 1]) - float(pts[f, 3, 1]))) + 1e-6
            want = (thr - (1 - il.EMA_ALPHA) * ns["mouth_ema"]) / il.EMA_ALPHA * span
            if 0.0 <= want < 0.2:
                gap[f] = np.float32(want) if f % 2 else np.nextafter(np.float32(want), np.float32(1))
        pts[f, 1] = (0.5, np.float32(0.60) + gap[f])
        face = {il.MOUTH_TOP: P(*pts[f, 0]), il.MOUTH_BOTTOM: P(*pts[f, 1]), il.LEFT_EYE_CORNER: P(*pts[f, 2]),
                il.RIGHT_EYE_CORNER: P(*pts[f, 3])}
        ns["face"] = face
        exec(code, ns)
        ema.append(ns["mouth_ema"]); state.append(bool(ns["mouth_state_open"])); opn.append(ns["openness"])
    out.update(gate_pts=pts, gate_idx=np.asarray([il.MOUTH_TOP, il.MOUTH_BOTTOM, il.LEFT_EYE_CORNER, il.RIGHT_EYE_CORNER], np.int32),
               gate_ema=np.asarray(ema, np.float64), gate_open=np.asarray(state, np.uint8), gate_openness=np.asarray(opn, np.float64),
               gate_consts=np.asarray([il.EMA_ALPHA, il.OPEN_THR, il.CLOSE_THR], np.float64))
    near = np.abs(np.asarray(ema) - il.OPEN_THR) < 1e-12
    print("gate: %d frames, %d flips, %d EMA values within 1e-12 of a threshold" % (F, int(np.abs(np.diff(out["gate_open"].astype(int))).sum()), int(near.sum())))

    # ---- (2) openness-gated clips
    path = os.path.join(REF, "inactive", "live_test_5.py")
    code, n2 = _statements_between(path, assigns("openv"), if_on("speaking"), drop=cv2_call)
    assert n2 == 3, n2  # openv, the counters' ``if``, the ``if not speaking`` state machine (with the prediction inside)
    code_noface, n2b = _statements_between(path, assigns("speaking"), assigns("hold"), drop=cv2_call, list_has=assigns("above_ct", "below_ct"))
    assert n2b == 6, n2b
    idxs = [int(i) for i in lt5_landmark_idxs()]
    K = len(idxs)
    torch.manual_seed(5)
    ns = dict(vars(lt5))
    D_NPZ = 2 * K + 1
    ns.update(idxs=idxs, D_NPZ=D_NPZ, D_IN=2 * D_NPZ, dev="cpu", labels=["hello", "yes", "no", "please", "thanks"],
              model=lt5.MLP(in_dim=2 * D_NPZ, num_classes=5).eval(),
              speaking=False, above_ct=0, below_ct=0, clip_buf=[], last_pred=None, last_conf=0.0, hold=0)
    F = 420
    lm = rs.uniform(0.3, 0.7, size=(F, K, 2)).astype(np.float32)
    # the openness is the y range of the K landmarks: squeeze them into a band whose height follows a script of talk / silence
    plan, f = [], 0
    for talk, quiet in ((2, 4), (3, 6), (9, 3), (4, 5), (12, 7), (70, 8), (6, 2), (3, 1), (3, 9), (25, 4), (8, 6), (5, 5), (30, 5)):
        plan += [1] * talk + [0] * quiet
    plan = (plan * 3)[:F]
    noface = np.zeros(F, bool)
    noface[[57, 58, 150, 301]] = True  # one of them in the middle of a clip
    height = np.where(np.asarray(plan) > 0, rs.uniform(0.19, 0.3, F), rs.uniform(0.05, 0.17, F))
    height[[20, 90]] = lt5.OPEN_THRESH  # exactly on the threshold: ``>`` is strict
    ylo = rs.uniform(0.3, 0.4, F)
    lm[:, :, 1] = (ylo[:, None] + rs.uniform(0, 1, (F, K)) * height[:, None]).astype(np.float32)
    lm[:, 0, 1] = ylo.astype(np.float32)
    lm[:, 1, 1] = (ylo + height).astype(np.float32)
    rec = collections.defaultdict(list)
    for f in range(F):
        ns["hold"] = 0
        if noface[f]:
            exec(code_noface, ns)
            rec["openv"].append(0.0); rec["emit_len"].append(0); rec["emit_sum"].append(0.0)
        else:
            face = [None] * 478
            for k, i in enumerate(idxs):
                face[i] = P(*lm[f, k])
            ns["face"] = face
            before = ns["clip_buf"]
            exec(code, ns)
            rec["openv"].append(ns["openv"])
            done = ns["hold"] == lt5.HOLD_FRAMES  # the prediction ran: the clip had >= 6 frames
            clip = before if ns["clip_buf"] is before else ns["clip_buf"]
            rec["emit_len"].append(len(ns["Xclip"]) if done else 0)
            rec["emit_sum"].append(float(np.asarray(ns["Xclip"], np.float64).sum()) if done else 0.0)
            del clip
        rec["speaking"].append(bool(ns["speaking"])); rec["above"].append(ns["above_ct"]); rec["below"].append(ns["below_ct"])
        rec["buf_len"].append(len(ns["clip_buf"]))
    out.update(clip_lm=lm, clip_idxs=np.asarray(idxs, np.int32), clip_noface=noface, clip_openv=np.asarray(rec["openv"], np.float64),
               clip_speaking=np.asarray(rec["speaking"], np.uint8), clip_above=np.asarray(rec["above"], np.int32),
               clip_below=np.asarray(rec["below"], np.int32), clip_buf_len=np.asarray(rec["buf_len"], np.int32),
               clip_emit_len=np.asarray(rec["emit_len"], np.int32), clip_emit_sum=np.asarray(rec["emit_sum"], np.float64),
               clip_consts=np.asarray([lt5.OPEN_THRESH, lt5.START_N, lt5.END_N, lt5.MAX_CLIP], np.float64))
    el = out["clip_emit_len"]
    print("clips: %d frames, %d clips emitted (lengths %s), %d at MAX_CLIP" % (F, int((el > 0).sum()), sorted(set(el[el > 0].tolist())), int((el == lt5.MAX_CLIP).sum())))

    # ---- (3) the sliding window of live_feed.py
    path = os.path.join(REF, "inactive", "live_feed.py")
    lf = _functions_of(path, ["extract_83_and_openness", "softmax_np", "GRUWordClassifier"],
                       ["LOWER_LIPS", "UPPER_LIPS", "LIP_ORDER", "PRED_EVERY"])
    max_t = 24
    ns = dict(lf)
    torch.manual_seed(6)
    ns.update(max_t=max_t, input_dim=83, DEVICE="cpu", id_to_label={i: "w%d" % i for i in range(7)}, frame_idx=0,
              buf=collections.deque(maxlen=max_t), model=lf["GRUWordClassifier"](input_dim=83, hidden=128, num_classes=7).eval(),
              last_label="...", last_conf=0.0)
    exec(_statements_between(path, assigns("WARMUP_MIN"), assigns("WARMUP_MIN"))[0], ns)
    tick = _statements_between(path, aug("frame_idx"), aug("frame_idx"))[0]
    code, n3 = _statements_between(path, assigns("feat", "open_val"), if_on("WARMUP_MIN", "PRED_EVERY"))
    assert n3 == 3, n3  # feat / open_val, buf.append, the prediction ``if``
    used = sorted(set(lf["LIP_ORDER"]) | {0, 17, 13, 14, 61, 291})
    F = 90
    pts = rs.uniform(0.3, 0.7, size=(F, len(used), 2)).astype(np.float32)
    noface = np.zeros(F, bool)
    noface[[3, 4, 17, 40, 41, 42, 77]] = True
    src = -np.ones((F, max_t), np.int32)  # for a frame that predicted: which frame every window row came from (-1: zero padding)
    feats, logits = np.zeros((F, 83), np.float32), np.zeros((F, 7), np.float32)
    with torch.no_grad():
        for f in range(F):
            exec(tick, ns)
            if noface[f]:  # :179-185: ``continue`` in front of the buffer
                continue
            full = np.zeros((478, 2), np.float32)
            full[used] = pts[f]
            ns["landmarks_xy"] = full
            ns.pop("X", None)
            exec(code, ns)
            feats[f] = ns["feat"]
            if "X" in ns:
                for r in range(max_t):
                    hit = np.flatnonzero((feats[: f + 1] == ns["X"][r]).all(1) & ~noface[: f + 1])
                    src[f, r] = hit[-1] if len(hit) and ns["X"][r].any() else -1
                logits[f] = ns["logits"]
    out.update(win_pts=pts, win_used=np.asarray(used, np.int32), win_noface=noface, win_src=src, win_feats=feats, win_logits=logits,
               win_consts=np.asarray([max_t, ns["WARMUP_MIN"], ns["PRED_EVERY"]], np.int32))
    print("window: %d frames, %d predictions, first at frame %d" % (F, int((src[:, 0] >= 0).sum()), int(np.argmax(src[:, 0] >= 0))))
    np.savez_compressed(os.path.join(HERE, "serving_loops.npz"), **out)
    print("wrote serving_loops")


def gen_live_loop(live, script="live_infer_official.py", roi_flag="use_roi", out_name="live_loop.npz"):
    """The per-frame body of the OFFICIAL live script (live_infer_official.py:272-296), run from its own statements: mouth width,
    the 60-150 px distance gate, ``extract_feature`` with the ``prev_xy`` it carries, the buffer append, and ``prev_xy = None`` when
    a frame of a recording falls outside the band.  ``use_roi`` is off in the namespace (``crop_roi_gray`` needs OpenCV; the crop
    BOX is pinned by crop.npz); pressing "r" is restated by its two assignments (:334-336: buffers emptied, ``prev_xy = None``).
    Called a second time for the RECORDER's loop (record_landmarks_official.py:182-201: the same shape, float64 mouth width, and
    ``prev_xy = None`` on every frame that is not appended, recording or not; ``SAVE_ROI`` off for the same reason)."""
    import ast

    assigns, aug, if_on, cv2_call = _ast_preds()
    draws = lambda n: cv2_call(n) or (isinstance(n, ast.If) and "DRAW_POINTS" in {x.id for x in ast.walk(n.test) if isinstance(x, ast.Name)})
    code, n = _statements_between(os.path.join(REF, script), assigns("mw"), if_on("recording", "in_range"), drop=draws)
    assert n == 3, n  # mw, in_range, ``if recording and in_range: ... else: ...``

    class P:
        __slots__ = ("x", "y")

        def __init__(self, x, y):
            self.x, self.y = float(x), float(y)

    rs = np.random.RandomState(31)
    idxs = [int(i) for i in live.FIXED_IDXS]
    K, w, h, F = len(idxs), 640, 480, 260
    pos = {i: k for k, i in enumerate(idxs)}
    base = rs.uniform(0.35, 0.65, size=(K, 2)).astype(np.float32)
    lm = (base[None] + rs.normal(0, 0.004, (F, K, 2))).astype(np.float32)
    # mouth corners: a horizontal distance that wanders in and out of the band, sits just inside / outside both edges, and leaves
    # for single frames
    width_px = 105 + 60 * np.sin(np.arange(F) / 9.0) + rs.normal(0, 2, F)
    width_px[[40, 41]] = [59.99, 60.01]
    width_px[[90, 91]] = [150.01, 149.99]
    width_px[[120, 160, 161, 200]] = [30.0, 400.0, 20.0, 155.0]
    for f in range(F):
        lm[f, pos[live.LEFT_CORNER]] = (0.5 - width_px[f] / (2 * w), 0.61)
        lm[f, pos[live.RIGHT_CORNER]] = (0.5 + width_px[f] / (2 * w), 0.61)
    recording = np.ones(F, bool)
    recording[:12] = False
    recording[130:150] = False
    ns = dict(vars(live))
    ns.update({roi_flag: False}, recording=False, bufX=[], bufR=[], bufT=[], prev_xy=None, w=w, h=h, ts=0)
    mw, in_range, appended, has_prev = np.zeros(F), np.zeros(F, bool), np.zeros(F, bool), np.zeros(F, bool)
    feats = np.zeros((F, 2 * K + 4), np.float32)
    for f in range(F):
        if recording[f] and not ns["recording"]:  # "r" pressed: :334-336
            ns["bufX"], ns["bufR"] = [], []
            ns["prev_xy"] = None
        ns["recording"], ns["ts"] = bool(recording[f]), 33 * f
        face = [None] * 478
        for i, k in pos.items():
            face[i] = P(*lm[f, k])
        ns["face"] = face
        n_before = len(ns["bufX"])
        exec(code, ns)
        mw[f], in_range[f] = ns["mw"], ns["in_range"]
        appended[f] = len(ns["bufX"]) == n_before + 1
        if appended[f]:
            feats[f] = ns["bufX"][-1]
        has_prev[f] = ns["prev_xy"] is not None
    assert feats.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, out_name), lm=lm, idxs=np.asarray(idxs, np.int32), wh=np.asarray([w, h], np.int32),
                        recording=recording, mouth_w=mw, in_range=in_range, appended=appended, has_prev=has_prev, feats=feats,
                        band=np.asarray([live.MOUTH_W_MIN_PX, live.MOUTH_W_MAX_PX], np.float64))
    print("wrote " + out_name + ": %d frames, %d appended, %d out of band while recording, %d resets of prev_xy" % (
        F, int(appended.sum()), int((recording & ~in_range).sum()), int((np.diff(has_prev.astype(int)) < 0).sum())))


def _check_against_the_loop_statements(tmo, sd, x_dim, C, use_roi, X, L, R, y, m_after2, loss2):
    """The two steps above are written out here; this runs the SAME two steps from the reference's own statements -- the optimiser
    and loss construction (train_model_official.py:403-406) and the body of its batch loop (:427-440), taken out of ``main()`` --
    on a second copy of the model and requires bit-equal parameters and loss: what the fixtures call "the reference's step" is the
    reference's loop, not a paraphrase of it."""
    import ast

    assigns, aug, if_on, cv2_call = _ast_preds()
    path = os.path.join(REF, "train_model_official.py")
    is_step = lambda n: isinstance(n, ast.Expr) and isinstance(n.value, ast.Call) and ast.unparse(n.value.func) == "opt.step"
    setup, n0 = _statements_between(path, assigns("opt"), assigns("loss_fn"))
    body, n1 = _statements_between(path, assigns("X"), is_step, list_has=is_step)
    assert n0 == 2 and n1 == 10, (n0, n1)  # X / lengths / y to the device, ``if use_roi``, logits, loss, zero_grad, backward, clip, step
    model = tmo.BiGRUClassifier(x_dim, C, use_roi=use_roi, roi_emb=32, hidden=192)
    model.load_state_dict(sd)
    model.eval()  # as above: both dropouts off
    ns = dict(vars(tmo))
    ns.update(model=model, use_roi=use_roi, DEVICE="cpu")
    exec(setup, ns)
    for _ in range(2):
        ns.update(X=X, lengths=L, R=R, y=y)
        exec(body, ns)
    assert float(ns["loss"]) == loss2, (float(ns["loss"]), loss2)
    for (k, a), (_, b) in zip(model.state_dict().items(), m_after2.state_dict().items()):
        assert torch.equal(a, b), k
    print("   two steps of the reference's own loop statements: identical parameters and loss")


def lt5_landmark_idxs():
    """inactive/live_test_5.py reads its landmark indices from a recorded clip's ``idxs`` (:78-84); the synthetic clip here uses
    the 40 lip landmarks of the official recorder (record_landmarks_official.py:30-44 order does not matter to either function)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), ".."))
    from silent_speech_amd.features import LIP_IDXS_40

    return LIP_IDXS_40


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    tmo, rec, live, tred = import_reference()
    only = set(sys.argv[1:])  # python make_golden.py [case names]: regenerate only those model cases
    if only == {"serving"}:
        gen_serving()
        gen_serving_loops()
        gen_live_loop(live)
        gen_live_loop(rec, "record_landmarks_official.py", "SAVE_ROI", "record_loop.npz")
        return
    for i, case in enumerate(MODEL_CASES):
        if not only or case[0] in only:
            gen_model_case(tmo, live, *case, seed=100 + i)
    if only:
        return
    gen_kat(tred)
    gen_features(rec, live)
    gen_crop(rec, live)
    gen_dataset(tmo)
    gen_harness(tmo)
    gen_loader(live)
    gen_serving()
    gen_serving_loops()
    gen_live_loop(live)
    gen_live_loop(rec, "record_landmarks_official.py", "SAVE_ROI", "record_loop.npz")


if __name__ == "__main__":
    main()
