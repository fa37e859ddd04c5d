"""GPU: the drop-in module, autograd path and fused trainer against the golden vectors (which the
reference produced) and against the CPU oracle on fresh seeded inputs.

Bar (BASELINE.json north_star): logits within 1e-3 fp32 of the CPU reference.  The checks below
use the much tighter tolerances the fp32 kernels actually reach.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import weights as W  # noqa: E402
from oracle import model_ref as MR  # noqa: E402

LOGIT_TOL = 1e-3  # the contract
TIGHT = 5e-5      # what we hold ourselves to


@pytest.fixture(scope="module")
def ss():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import silent_speech_amd as ss_

    return ss_


def load_case(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    use_roi = bool(int(d["use_roi"]))
    sd = W.make_state_dict(int(d["seed"]), int(d["x_dim"]), int(d["num_classes"]), use_roi, gru_layers=int(d["layers"]))
    X, Lh, y = torch.from_numpy(d["X"]), torch.from_numpy(d["lengths"]), torch.from_numpy(d["y"])
    R = torch.from_numpy(d["R"]) if use_roi else None
    return d, sd, X, Lh, R, y


def build(ss, d, sd, standardize=True):
    m = ss.BiGRUClassifier(int(d["x_dim"]), int(d["num_classes"]), use_roi=bool(int(d["use_roi"])),
                           gru_layers=int(d["layers"]), roi_standardize=standardize)
    m.load_state_dict(sd)  # strict: key names and shapes are the reference's
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["model_lm_only", "model_roi64", "model_shipped", "model_live_l1", "model_live_l2", "model_shipped_t90"])
def test_logits_match_reference_golden(ss, golden_dir, name):
    d, sd, X, Lh, R, y = load_case(golden_dir, name)
    m = build(ss, d, sd, standardize=(str(d["cls"]) == "train"))
    with torch.no_grad():
        logits = m(X.cuda(), Lh, R.cuda() if R is not None else None)  # lengths stay on the host, as in the reference
    err = float((logits.cpu() - torch.from_numpy(d["logits"])).abs().max())
    assert err < LOGIT_TOL, err
    assert err < TIGHT, f"{name}: logit max-abs-err {err:.3e}"


def test_gru_lost_partner_reaches_the_host(ss):
    """The failure channel of the multi-CU GRU recurrence (gru_split.h): a workgroup that never publishes must not hang the
    GPU -- its partners' bounded sweeps give up, poison what they own with NaN from then on, count the event in the sync header
    and ``model.check_health()`` raises.  Forced through the header's fault-injection word (sync[5] = 1 + workgroup index):
    the owner of the workspace sets it, the kernels only read it.  Afterwards the same workspace serves correct results again
    (the generation counter moved on; nothing has to be cleared)."""
    sd = W.make_state_dict(2, 84, 5, False)
    B, T = 32, 6
    X, Lh, _, y = W.make_inputs(2, B, T, 84, 5, None)
    m = ss.BiGRUClassifier(84, 5, use_roi=False)
    m.load_state_dict(sd)
    m.cuda().eval()
    Xd = X.cuda()
    with torch.no_grad():
        good = m(Xd, Lh).cpu()
    ws = m._workspace(Xd, None, train=False)
    if ws.gru_sync is None:
        pytest.skip("this shape does not take the multi-CU recurrence")
    m.check_health()
    assert int(ws.gru_sync[2]) == 0
    # workgroup 9 of the layer-0 forward launch (and of every later launch) plays dead: 4 pairs are launched as 8 (the pad that
    # puts the partners of a pair on one XCD), part-major: workgroup b is pair b % 8, part b // 8
    ws.gru_sync[5] = 1 + 9
    with torch.no_grad():
        bad = m(Xd, Lh).cpu()
    torch.cuda.synchronize()
    assert int(ws.gru_sync[2]) > 0, "no bounded wait gave up"
    # workgroup 9 = (slice 0, reverse direction), part 1: its five partners run out of patience and emit NaN for everything they
    # own from then on, so the 16 clips of slice 0 are poisoned and the other slice is untouched
    assert torch.isnan(bad[:16]).all(), "the poisoned result did not reach the logits"
    assert torch.equal(bad[16:], good[16:])
    with pytest.raises(RuntimeError, match="GRU recurrence"):
        m.check_health()
    ws.gru_sync[5] = 0
    ws.gru_sync[2] = 0  # acknowledge: the counter is never reset by the kernels
    with torch.no_grad():
        again = m(Xd, Lh).cpu()
    assert torch.equal(again, good)
    m.check_health()


def test_state_dict_surface(ss):
    m = ss.BiGRUClassifier(180, 10, use_roi=True)
    ref = W.param_shapes(180, 10, True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, shp in ref.items():
        assert tuple(sd[k].shape) == shp and sd[k].dtype == torch.float32
    assert sum(v.numel() for v in sd.values()) == 1190819
    m2 = ss.BiGRUClassifier(84, 5, use_roi=False)
    assert not any(k.startswith("roi_cnn") for k in m2.state_dict())
    with pytest.raises(RuntimeError):
        m2(torch.zeros(1, 3, 84), torch.tensor([3]))  # CPU tensors: no CPU path, must fail loudly


@pytest.mark.parametrize("name", ["model_lm_only", "model_roi64", "model_shipped", "model_shipped_t90"])
def test_autograd_grads_match_oracle_and_golden(ss, golden_dir, name):
    d, sd, X, Lh, R, y = load_case(golden_dir, name)
    m = build(ss, d, sd)
    logits = m(X.cuda(), Lh.cuda(), R.cuda() if R is not None else None)
    loss = torch.nn.functional.cross_entropy(logits, y.cuda(), label_smoothing=0.05)
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 2e-5
    _, _, grads = MR.loss_and_grads(sd, X, Lh, R, y)
    for k, p in m.named_parameters():
        ref = grads[k]
        got = p.grad.cpu()
        if k == "pool.score.bias":
            # softmax is shift-invariant: the true gradient is exactly 0 and both sides hold rounding noise of the sum over the
            # attention weights (~1e-8; its size moves with the summation order of the kernels in front)
            assert float(got.abs().max()) < 1e-6 and float(ref.abs().max()) < 1e-6
            continue
        scale = max(float(ref.abs().max()), 1e-4)
        bad = (got - ref).abs() > 2e-4 * scale + 2e-3 * ref.abs()
        assert not bad.any(), f"{k}: max err {float((got - ref).abs().max()):.3e} vs scale {scale:.3e}"
        # and the reference's own numbers (reduced form)
        np.testing.assert_allclose(W.reduce_tensor(got), d["grad::" + k], atol=3e-4 * scale * max(1.0, np.sqrt(got.numel() / 53)),
                                   rtol=5e-3, err_msg=k)


@pytest.mark.parametrize("name", ["model_lm_only", "model_roi64", "model_shipped_t90"])
def test_fused_trainer_two_steps_match_reference(ss, golden_dir, name):
    d, sd, X, Lh, R, y = load_case(golden_dir, name)
    m = build(ss, d, sd)
    tr = ss.Trainer(m, dropout=False)
    Xd, Ld, yd = X.cuda(), Lh.cuda(), y.cuda()
    Rd = R.cuda() if R is not None else None
    loss1, correct1 = tr.step(Xd, Ld, Rd, yd)
    assert abs(float(loss1) - float(d["loss"])) < 2e-5
    assert abs(float(tr.grad_norm()) - float(d["total_norm"])) < 1e-3 * float(d["total_norm"])
    sd1 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k, v in sd1.items():
        atol = 6.1e-4 if k == "pool.score.bias" else 3e-6  # see tests/test_oracle_golden.py
        np.testing.assert_allclose(W.reduce_tensor(v), d["step1::" + k], atol=atol, rtol=2e-5, err_msg=k)
    loss2, _ = tr.step(Xd, Ld, Rd, yd)
    assert abs(float(loss2) - float(d["loss2"])) < 5e-5
    with torch.no_grad():
        after = m(Xd, Ld, Rd)
    err = float((after.cpu() - torch.from_numpy(d["logits_after2"])).abs().max())
    assert err < LOGIT_TOL and err < 2e-4, err


def test_minimal_change_loop_matches_reference_step(ss, golden_dir):
    """the reference's own loop body (train_model_official.py:433-439) with only the class swapped:
    forward -> CrossEntropyLoss(label_smoothing) -> zero_grad -> backward -> clip_grad_norm_ -> torch.optim.Adam.step,
    against the parameters the reference had after its first step (``step1::*``) and its second loss."""
    d, sd, X, Lh, R, y = load_case(golden_dir, "model_roi64")
    m = build(ss, d, sd)  # eval(): both dropouts off, as when the golden step was made
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    loss_fn = torch.nn.CrossEntropyLoss(label_smoothing=0.05)
    Xd, Rd, yd = X.cuda(), R.cuda(), y.cuda()
    losses = []
    for _ in range(2):
        logits = m(Xd, Lh, Rd)
        loss = loss_fn(logits, yd)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
        if len(losses) == 1:
            assert abs(float(total) - float(d["total_norm"])) < 1e-3 * float(d["total_norm"])
            for k, v in m.state_dict().items():
                atol = 6.1e-4 if k == "pool.score.bias" else 3e-6
                np.testing.assert_allclose(W.reduce_tensor(v.detach().cpu()), d["step1::" + k], atol=atol, rtol=2e-5, err_msg=k)
    assert abs(losses[0] - float(d["loss"])) < 2e-5 and abs(losses[1] - float(d["loss2"])) < 5e-5


def test_two_forwards_one_backward_and_stale_stash_raises(ss, golden_dir):
    """two grad-enabled forwards of ONE shape before a backward must each keep their own stash (the first
    one's gradients used to be silently wrong); with more live stashes than workspaces the oldest must fail loudly."""
    d, sd, X, Lh, R, y = load_case(golden_dir, "model_roi64")
    m = build(ss, d, sd)
    Xa, Ra = X.cuda(), R.cuda()
    g = torch.Generator().manual_seed(5)
    Xb = (X + 0.3 * torch.randn(X.shape, generator=g)).cuda()
    Rb = torch.randint(0, 256, R.shape, generator=g, dtype=torch.uint8).cuda()
    la = m(Xa, Lh, Ra)
    lb = m(Xb, Lh, Rb)
    loss = torch.nn.functional.cross_entropy(la, y.cuda(), label_smoothing=0.05) + \
        torch.nn.functional.cross_entropy(lb, y.cuda(), label_smoothing=0.05)
    loss.backward()
    _, _, ga = MR.loss_and_grads(sd, X, Lh, R, y)
    _, _, gb = MR.loss_and_grads(sd, Xb.cpu(), Lh, Rb.cpu(), y)
    for k, p in m.named_parameters():
        ref = ga[k] + gb[k]
        scale = max(float(ref.abs().max()), 1e-4)
        bad = (p.grad.cpu() - ref).abs() > 4e-4 * scale + 2e-3 * ref.abs()
        assert not bad.any(), f"{k}: {float((p.grad.cpu() - ref).abs().max()):.3e} vs {scale:.3e}"
    # three alive at once with two workspaces: the oldest stash is taken over and its backward must say so
    l1, l2, l3 = m(Xa, Lh, Ra), m(Xb, Lh, Rb), m(Xa, Lh, Ra)
    l3.sum().backward()
    l2.sum().backward()
    with pytest.raises(RuntimeError, match="stashed"):
        l1.sum().backward()


def test_trainer_survives_module_moves(ss, golden_dir):
    """``model.to(dev)`` after the Trainer was built used to re-allocate the flat buckets, leaving the Trainer
    to accumulate into an orphan: training silently stopped.  A no-op move must keep the bucket; a real move must be
    followed (gradient views re-taken, Adam moments moved)."""
    d, sd, X, Lh, R, y = load_case(golden_dir, "model_roi64")
    m = build(ss, d, sd)
    tr = ss.Trainer(m, dropout=False)
    flat0, ver0 = m.flat_params.data_ptr(), m._bucket_version
    m.to("cuda")
    m.float()
    m.cuda()
    assert m.flat_params.data_ptr() == flat0 and m._bucket_version == ver0
    Xd, Ld, Rd, yd = X.cuda(), Lh.cuda(), R.cuda(), y.cuda()
    loss1, _ = tr.step(Xd, Ld, Rd, yd)
    assert abs(float(loss1) - float(d["loss"])) < 2e-5
    m.cpu()
    m.cuda()  # a real round trip: new buckets
    assert m._bucket_version != ver0
    loss2, _ = tr.step(Xd, Ld, Rd, yd)
    assert abs(float(loss2) - float(d["loss2"])) < 5e-5
    with torch.no_grad():
        after = m(Xd, Ld, Rd)
    assert float((after.cpu() - torch.from_numpy(d["logits_after2"])).abs().max()) < 2e-4


def test_one_rank_rccl_trainer_equals_single_process(ss, tmp_path):
    """The data-parallel leg on the hardware at hand: a FRESH child process (torch.distributed.run, one rank) joins an
    "nccl" (= RCCL) group and pushes the flat gradient bucket of two training steps through dist.all_reduce; the result
    must equal the single-process trainer (a sum over one rank is the identity; dropout on, so the rank-folded seed is
    covered too), to the run-to-run noise of the float atomics (the loss itself is summed with one atomic per workgroup)."""
    import socket
    import subprocess
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "child.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "_nccl_child.py"), out],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out, map_location="cpu", weights_only=False)
    assert got["backend"] == "nccl" and got["world"] == 1
    B, T = 16, 9
    sd = W.make_state_dict(21, 84, 5, True)
    X, Lh, R, y = W.make_inputs(21, B, T, 84, 5, (64, 64))
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().train()
    tr = ss.Trainer(m, dropout=True)
    losses = [float(tr.step(X.cuda(), Lh.cuda(), R.cuda(), y.cuda())[0]) for _ in range(2)]
    # same kernels, same seeds; what may differ between two runs is the order of the float atomics that sum the CNN and
    # GRU-bias gradients over workgroups (rounding noise that Adam turns into at most a fraction of lr = 3e-4)
    assert abs(losses[0] - got["losses"][0]) < 1e-6 and abs(losses[1] - got["losses"][1]) < 1e-5, (losses, got["losses"])
    for k, v in m.state_dict().items():
        atol = 6.1e-4 if k == "pool.score.bias" else 2e-5
        assert float((v.cpu() - got["sd"][k]).abs().max()) <= atol, k


def test_bench_multi_rank_branch_runs_under_torchrun(ss):
    """bench.py's OWN data-parallel branch -- init_process_group("nccl", device_id=), broadcast of the flat parameters, the
    gradient all-reduce in every step, dist.barrier, all_reduce(MAX) of the wall time -- exactly as the driver starts it for
    N > 1 (python -m torch.distributed.run ... bench.py --gpus N), here with the one rank this box has (--force-dist), in a
    fresh child that has not touched the GPU before it joins the group.  The line must carry the all-reduce time."""
    import json
    import socket
    import subprocess
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
           "--batch", "32", "--no-cpu-baseline", "--no-config4", "--no-config5", "--no-shipped"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert d["allreduce_ms"] > 0 and d["allreduce_bytes"] == 4 * 1079588
    assert d["final_loss"] == d["final_loss"]


@pytest.mark.parametrize("layers", [1, 2])
def test_checkpoint_and_topk_against_the_reference_loader(ss, golden_dir, layers, tmp_path):
    """a11 + a12 pinned: tests/golden/loader.npz holds what the REFERENCE's load_classifier / forward / topk_from_logits
    (live_infer_official.py:198-226) made of a ``.pt`` this build wrote.  Same file, this build's loader and kernels."""
    d = np.load(os.path.join(golden_dir, "loader.npz"), allow_pickle=False)
    seed, x_dim, C, B, T = int(d["seed"]), int(d["x_dim"]), int(d["num_classes"]), int(d["B"]), int(d["T"])
    labels = [str(x) for x in d["labels"]]
    sd = W.make_state_dict(seed + layers, x_dim, C, True, gru_layers=layers)
    X, Lh, R, _ = W.make_inputs(seed + layers, B, T, x_dim, C, (48, 96), lengths=[T, 5])
    m0 = ss.BiGRUClassifier(x_dim, C, use_roi=True, gru_layers=layers)
    m0.load_state_dict(sd)
    path = str(tmp_path / "word_model_points_roi.pt")
    ss.save_checkpoint(path, m0, labels, max_t=90, roi_w=96, roi_h=48, seed=42)
    model, id_to_label, max_t, use_roi = ss.load_classifier(path, roi_standardize=False)  # the live script's forward
    assert max_t == 90 and use_roi and model.cfg.gru_layers == layers
    with torch.no_grad():
        logits = model(X.cuda(), Lh, R.cuda())
    assert float((logits.cpu() - torch.from_numpy(d[f"l{layers}::logits"])).abs().max()) < TIGHT
    for b in range(B):
        top = ss.topk_from_logits(logits[b:b + 1], id_to_label, k=3)
        assert [t[0] for t in top] == [str(x) for x in d[f"l{layers}::top_labels{b}"]]
        np.testing.assert_allclose([t[1] for t in top], d[f"l{layers}::top_probs{b}"], rtol=2e-4)
    # free logit rows: the kernel against the reference's softmax / argsort
    from silent_speech_amd.checkpoint import softmax_topk

    probs, idx = softmax_topk(torch.from_numpy(d["free::logits"]).cuda(), 3)
    assert np.array_equal(idx.cpu().numpy(), d["free::top_idx"])
    np.testing.assert_allclose(probs.cpu().numpy(), d["free::top_probs"], rtol=3e-6)
    # k > C and wide rows (C > 64: more than one class per lane)
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(5, 300, generator=g)
    p2, i2 = softmax_topk(wide.cuda(), 5)
    ref = torch.softmax(wide, -1)
    order = torch.argsort(ref, dim=1, descending=True)[:, :5]
    assert torch.equal(i2.cpu().long(), order) and float((p2.cpu() - ref.gather(1, order)).abs().max()) < 1e-7


def test_padding_never_leaks_and_batch_permutes(ss, golden_dir):
    d, sd, X, Lh, R, y = load_case(golden_dir, "model_roi64")
    m = build(ss, d, sd)
    with torch.no_grad():
        a = m(X.cuda(), Lh, R.cuda())
        X2, R2 = X.clone(), R.clone()
        for b, n in enumerate(Lh.tolist()):
            X2[b, n:] = 123.0
            R2[b, n:] = 77
        b_ = m(X2.cuda(), Lh, R2.cuda())
        perm = torch.tensor([2, 0, 1])
        c = m(X[perm].cuda(), Lh[perm], R[perm].cuda())
    assert torch.equal(a, b_), "values past a clip's length changed its logits"
    assert torch.equal(a[perm], c), "a clip's logits depend on its position in the batch"


def test_config2_shapes_vs_oracle(ss):
    """BASELINE config 2 geometry (D=84, 64x64 ROI, T=30) at a batch the CPU oracle finishes in seconds."""
    B, T = 24, 30
    sd = W.make_state_dict(7, 84, 5, True)
    X, Lh, R, y = W.make_inputs(7, B, T, 84, 5, (64, 64))
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    with torch.no_grad():
        logits = m(X.cuda(), Lh, R.cuda())
    ref = MR.forward(sd, X, Lh, R, impl="aten")
    err = float((logits.cpu() - ref).abs().max())
    assert err < TIGHT, err
    # one fused step vs the oracle's step
    tr = ss.Trainer(m, dropout=False)
    loss, _ = tr.step(X.cuda(), Lh.cuda(), R.cuda(), y.cuda())
    sd2 = {k: v.clone() for k, v in sd.items()}
    loss_ref, _, _, total = MR.train_step(sd2, {}, X, Lh, R, y, impl="aten")
    assert abs(float(loss) - float(loss_ref)) < 2e-5
    assert abs(float(tr.grad_norm()) - total) < 1e-3 * total
    with torch.no_grad():
        after = m(X.cuda(), Lh, R.cuda())
    ref_after = MR.forward(sd2, X, Lh, R, impl="aten")
    assert float((after.cpu() - ref_after).abs().max()) < 2e-4


@pytest.mark.parametrize("roi,B,T", [((48, 96), 9, 11), ((32, 32), 20, 6), ((64, 64), 130, 5)])
def test_train_step_other_geometries_vs_oracle(ss, roi, B, T):
    """The reference's shipped ROI size (48x96), the smallest built one, and a batch whose (slice, direction) count is
    not a multiple of 8 (write-through granules) next to one that is: one fused step against the oracle's."""
    sd = W.make_state_dict(11, 84, 5, True)
    X, Lh, R, y = W.make_inputs(11, B, T, 84, 5, roi)
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    tr = ss.Trainer(m, dropout=False)
    loss, _ = tr.step(X.cuda(), Lh.cuda(), R.cuda(), y.cuda())
    sd2 = {k: v.clone() for k, v in sd.items()}
    loss_ref, _, _, total = MR.train_step(sd2, {}, X, Lh, R, y, impl="aten")
    assert abs(float(loss) - float(loss_ref)) < 2e-5
    assert abs(float(tr.grad_norm()) - total) < 1e-3 * total
    with torch.no_grad():
        after = m(X.cuda(), Lh, R.cuda())
    ref_after = MR.forward(sd2, X, Lh, R, impl="aten")
    assert float((after.cpu() - ref_after).abs().max()) < 2e-4


def test_micro_batch_streams_match_oracle(ss):
    """Two micro-batches on two HIP streams (Trainer(micro_batches=2)) add up to the same step as the oracle's."""
    B, T = 32, 12
    sd = W.make_state_dict(11, 84, 5, True)
    X, Lh, R, y = W.make_inputs(11, B, T, 84, 5, (64, 64))
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    tr = ss.Trainer(m, dropout=False, micro_batches=2)
    Xd, Ld, Rd, yd = X.cuda(), Lh.cuda(), R.cuda(), y.cuda()
    loss, correct = tr.step(Xd, Ld, Rd, yd)
    sd2 = {k: v.clone() for k, v in sd.items()}
    loss_ref, logits_ref, _, total = MR.train_step(sd2, {}, X, Lh, R, y, impl="aten")
    assert abs(float(loss) - float(loss_ref)) < 2e-5
    assert abs(float(tr.grad_norm()) - total) < 1e-3 * total
    assert int(correct) == int((logits_ref.argmax(1) == y).sum())
    with torch.no_grad():
        after = m(Xd, Lh, Rd)
    assert float((after.cpu() - MR.forward(sd2, X, Lh, R, impl="aten")).abs().max()) < 2e-4
    # and it is the same step as the single-stream trainer takes
    m1 = ss.BiGRUClassifier(84, 5, use_roi=True)
    m1.load_state_dict(sd)
    m1.cuda().eval()
    ss.Trainer(m1, dropout=False, micro_batches=1).step(Xd, Ld, Rd, yd)
    for (k, a), (_, b) in zip(m.state_dict().items(), m1.state_dict().items()):
        # float atomics add the two slices' gradients in a different order than one slice does; where a gradient element
        # is ~0 Adam's normalisation turns that rounding noise into a visible fraction of lr (3e-4)
        atol = 6.1e-4 if k == "pool.score.bias" else 2e-5
        assert float((a - b).abs().max()) <= atol, k


@pytest.mark.parametrize("roi_hw", [(64, 64), (48, 96)])
def test_padding_frames_are_skipped_without_changing_a_number(ss, roi_hw, monkeypatch):
    """A ragged batch with the CNN walking only the frames inside their clips (the default) against the same batch with every
    padded frame computed as the reference does (SS_CNN_SKIP_PADDING=0): logits bit-equal -- a clip's rows are the same numbers,
    the padding rows never reach the packed recurrence --, gradients equal up to the order of the sums, and a fused train step
    sees the same loss and gradient norm.  Autograd path (its own list launch) and trainer path (the list comes from the prologue)."""
    from silent_speech_amd import engine as E

    B, T = 19, 11
    X, Lh, R, y = W.make_inputs(41, B, T, 84, 5, roi_hw, lengths=[T, 1, 2, 7] + [3 + (5 * b) % 8 for b in range(B - 4)])
    sd = W.make_state_dict(9, 84, 5, True, gru_layers=2)
    Xd, Ld, Rd, yd = X.cuda(), Lh.cuda(), R.cuda(), y.cuda()
    res = {}
    for skip in (True, False):
        monkeypatch.setattr(E, "SKIP_PADDED_FRAMES", skip)
        m = ss.BiGRUClassifier(84, 5, use_roi=True, gru_layers=2)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        logits = m(Xd, Ld, Rd)
        torch.nn.functional.cross_entropy(logits, yd, label_smoothing=0.05).backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        m.zero_grad()
        tr = ss.Trainer(m, dropout=False)
        loss, _ = tr.step(Xd, Ld, Rd, yd)
        for slot in (("autograd", 0), 0):  # the autograd forward lists the frames with its own launch, the trainer's prologue does it
            ws = m._workspace(Xd, Rd, train=True, slot=slot)
            assert (ws.frames is not None) == skip
            if skip:
                assert int(ws.frames[0]) == int(Lh.sum()) and int(ws.frames[1:1 + int(Lh.sum())].diff().min()) > 0
        res[skip] = (logits.detach().clone(), grads, float(loss), float(tr.grad_norm()))
    assert torch.equal(res[True][0], res[False][0])
    assert abs(res[True][2] - res[False][2]) < 1e-6
    for k in res[True][1]:
        a, b = res[True][1][k], res[False][1][k]
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6), k
    assert abs(res[True][3] - res[False][3]) < 1e-5 * res[False][3]  # (Adam's first step is sign-like: parameters are not compared)
    # and against the oracle, which computes every frame
    _, _, ref = MR.loss_and_grads(sd, X, Lh, R, y)
    for k, gg in res[True][1].items():
        if k == "pool.score.bias":
            continue
        scale = max(float(ref[k].abs().max()), 1e-4)
        assert float((gg.cpu() - ref[k]).abs().max()) < 2e-4 * scale + 2e-3 * float(ref[k].abs().max()), k


def test_a_batch_of_full_clips_is_followed_by_kernels_without_the_frame_list(ss, monkeypatch):
    """The backward kernel pays 1 % for walking a list, so a training workspace remembers (pinned memory, an 8-byte copy on the
    side stream, never waited for) how many frames of the last batch it saw finish lay inside a clip; a step that follows a batch
    of full clips launches forward and backward without the list.  The guess can be stale by a step in either direction --
    then the padding frames are computed as the reference computes them.  Same losses as a model that never uses the list."""
    from silent_speech_amd import engine as E

    B, T, hw = 17, 9, (64, 64)
    Xf, Lf, Rf, yf = W.make_inputs(3, B, T, 84, 5, hw, lengths=[T] * B)
    Xr, Lr, Rr, yr = W.make_inputs(4, B, T, 84, 5, hw, lengths=[T] + [1 + (3 * b) % T for b in range(B - 1)])
    sd = W.make_state_dict(2, 84, 5, True, gru_layers=2)
    seq = [(Xf, Lf, Rf, yf), (Xf, Lf, Rf, yf), (Xr, Lr, Rr, yr), (Xr, Lr, Rr, yr), (Xf, Lf, Rf, yf), (Xf, Lf, Rf, yf)]
    want_listed = [True, False, False, True, True, False]  # (every step is synchronised here, so the guess is exactly one step old)
    losses = {}
    for skip in (True, False):
        monkeypatch.setattr(E, "SKIP_PADDED_FRAMES", skip)
        m = ss.BiGRUClassifier(84, 5, use_roi=True, gru_layers=2)
        m.load_state_dict(sd)
        tr = ss.Trainer(m.cuda().train(), dropout=False)
        out = []
        for k, (X, Lh, R, y) in enumerate(seq):
            loss, _ = tr.step(X.cuda(), Lh.cuda(), R.cuda(), y.cuda())
            torch.cuda.synchronize()
            ws = m._workspace(X.cuda(), R.cuda(), train=True, slot=0)
            if skip:
                assert ws.walk_listed == want_listed[k], (k, ws.walk_listed)
                assert int(ws.frames_seen[0]) == int(Lh.sum()) == int(ws.frames[0])
            else:
                assert ws.frames is None and not ws.walk_listed
            out.append(float(loss))
        losses[skip] = out
    for a, b in zip(losses[True], losses[False]):
        assert abs(a - b) < 2e-3 * max(1.0, abs(b)), (losses[True], losses[False])
    assert abs(losses[True][0] - losses[False][0]) < 1e-6


@pytest.mark.parametrize("roi_hw,chunk_rows", [((96, 96), None), ((40, 56), None), ((40, 56), 5000), ((20, 12), None)])
def test_any_roi_size_runs_layer_by_layer(ss, roi_hw, chunk_rows, monkeypatch):
    """ROI sizes outside the fused kernels' set (64x64, 48x96, 32x32) -- the reference takes whatever ROI_H x ROI_W its constants
    say, SURVEY section 4 lists 96x96 among the parity variants -- run layer by layer (cnn_generic.py: im2col + GEMM, csrc/
    roi_cnn_generic.hip): logits, every gradient and one fused train step against the oracle; ``chunk_rows`` forces several frame
    chunks (incl. a ragged last one)."""
    from silent_speech_amd import cnn_generic

    if chunk_rows:
        monkeypatch.setattr(cnn_generic, "MAX_GEMM_ROWS", chunk_rows)
    B, T = 3, 5
    sd = W.make_state_dict(12, 84, 5, True)
    X, Lh, R, y = W.make_inputs(12, B, T, 84, 5, roi_hw, lengths=[5, 3, 1])
    # a constant frame takes the std clamp.  Only the all-zero one is a parity target at these sizes: for a constant u > 0 the
    # reference's float32 mean of H W copies of fl(u/255) is exact when H W is a power of two (64 x 64) and off by rounding
    # otherwise, and the clamp divides that rounding noise by 1e-6 -- the kernels (integer sums: exactly 0) are then the more
    # accurate side of a comparison that means nothing
    R[0, 0] = 0
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    logits = m(X.cuda(), Lh.cuda(), R.cuda())
    ref = MR.forward(sd, X, Lh, R)
    assert float((logits.detach().cpu() - ref).abs().max()) < TIGHT
    assert any(w.cnn_generic is not None for w in m._ws_cache.values()), "the fused kernels took a shape they are not built for"
    loss = torch.nn.functional.cross_entropy(logits, y.cuda(), label_smoothing=0.05)
    loss.backward()
    l_ref, _, grads = MR.loss_and_grads(sd, X, Lh, R, y)
    assert abs(float(loss) - float(l_ref)) < 2e-5
    for k, p in m.named_parameters():
        got, want = p.grad.cpu(), grads[k]
        if k == "pool.score.bias":
            assert float(got.abs().max()) < 1e-6
            continue
        scale = max(float(want.abs().max()), 1e-4)
        bad = (got - want).abs() > 2e-4 * scale + 2e-3 * want.abs()
        assert not bad.any(), f"{k}: max err {float((got - want).abs().max()):.3e} vs scale {scale:.3e}"
    # the fused trainer takes the same path
    m2 = ss.BiGRUClassifier(84, 5, use_roi=True)
    m2.load_state_dict(sd)
    m2.cuda().train()
    tr = ss.Trainer(m2, dropout=False)
    l2, _ = tr.step(X.cuda(), Lh.cuda(), R.cuda(), y.cuda())
    sd2 = {k: v.clone() for k, v in sd.items()}
    l2_ref, _, _, _ = MR.train_step(sd2, {}, X, Lh, R, y, impl="explicit")
    assert abs(float(l2) - float(l2_ref)) < 2e-5
    with torch.no_grad():
        after = m2.eval()(X.cuda(), Lh.cuda(), R.cuda()).cpu()
    assert float((after - MR.forward(sd2, X, Lh, R)).abs().max()) < 1e-3
    with pytest.raises(RuntimeError, match="multiples of 4"):
        ss.BiGRUClassifier(84, 5, use_roi=True).cuda()(X.cuda(), Lh.cuda(), torch.zeros(B, T, 30, 44, dtype=torch.uint8).cuda())


def test_full_size_properties(ss):
    """BASELINE config 2 at full size (B=256): properties that need no oracle run."""
    B, T = 256, 30
    sd = W.make_state_dict(9, 84, 5, True)
    X, Lh, R, y = W.make_inputs(9, B, T, 84, 5, (64, 64))
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    Xd, Rd = X.cuda(), R.cuda()
    with torch.no_grad():
        a = m(Xd, Lh, Rd)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
        b = m(Xd[perm.cuda()], Lh[perm], Rd[perm.cuda()])
        sub = m(Xd[:160].contiguous(), Lh[:160], Rd[:160].contiguous())
        sub16 = m(Xd[:16].contiguous(), Lh[:16], Rd[:16].contiguous())
        sub64 = m(Xd[:64].contiguous(), Lh[:64], Rd[:64].contiguous())
        sub128 = m(Xd[:128].contiguous(), Lh[:128], Rd[:128].contiguous())
    assert torch.isfinite(a).all()
    assert torch.equal(a[perm.cuda()], b)
    assert torch.equal(a[:160], sub), "a clip's logits depend on the batch size"
    # up to 128 clips the forward recurrence runs over twelve parts instead of six (gru_split_fwd_parts): its contraction is cut
    # into four k slices instead of two, so the sums associate differently -- the last bits of h, not more
    assert float((a[:16] - sub16).abs().max()) < 2e-5, "a small batch's logits drift from the large batch's"
    # ... and INSIDE each regime the bits do not depend on the batch size ("Batch size and the last bits")
    assert torch.equal(sub64[:16], sub16) and torch.equal(sub128[:64], sub64), "logits differ between two batches of <= 128 clips"
    # training decreases the loss on a fixed batch, and stays finite
    tr = ss.Trainer(m, dropout=True)
    m.train()
    losses = [float(tr.step(Xd, Lh.cuda(), Rd, y.cuda())[0]) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_train_mode_dropout_is_active_and_seeded(ss, golden_dir):
    d, sd, X, Lh, R, y = load_case(golden_dir, "model_lm_only")
    m = build(ss, d, sd).train()
    Xd = X.cuda()
    with torch.no_grad():
        a = m(Xd, Lh)
        b = m(Xd, Lh)
        m.eval()
        c = m(Xd, Lh)
    assert not torch.equal(a, c)          # dropout changes the logits
    assert torch.equal(a, b)              # same step seed -> same mask (no_grad calls do not advance it)
    assert float((a - c).abs().max()) < 1.0


def test_hipgraph_inference_equals_eager(ss):
    """BASELINE config 4 flavour (sliding T=60 windows, forward-only, hipGraph-captured) at a test-sized batch."""
    B, T = 96, 60
    sd = W.make_state_dict(13, 84, 5, True)
    X, Lh, R, y = W.make_inputs(13, B, T, 84, 5, (64, 64), lengths=[T] * B)
    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    Xd, Rd = X.cuda(), R.cuda()
    with torch.no_grad():
        eager = m(Xd, Lh, Rd)
    g = ss.GraphedInference(m, B, T, (64, 64), topk=3)
    out1 = g(Xd, Lh, Rd).clone()
    assert torch.equal(out1, eager)
    # softmax + top-3 of every window ride in the same graph (live_infer_official.py:223-226 per window)
    pr = torch.softmax(eager, -1)
    order = torch.argsort(pr, dim=1, descending=True)[:, :3]
    assert torch.equal(g.top_idx.long(), order) and float((g.top_probs - pr.gather(1, order)).abs().max()) < 1e-6
    # replay on new window contents and ragged lengths without re-capturing
    X2, L2, R2, _ = W.make_inputs(14, B, T, 84, 5, (64, 64))
    with torch.no_grad():
        eager2 = m(X2.cuda(), L2, R2.cuda())
    out2 = g(X2.cuda(), L2, R2.cuda()).clone()
    assert torch.equal(out2, eager2) and not torch.equal(out2, out1)
    ref = MR.forward(sd, X2[:8], L2[:8], R2[:8], impl="aten")
    assert float((out2[:8].cpu() - ref).abs().max()) < TIGHT


def test_real_weights_kat_gru(ss, golden_dir):
    """inactive/word_model_5.pt (1-layer BiGRU 83->64): trained weights through the GEMM + recurrence kernels."""
    from silent_speech_amd import _lib as L
    from silent_speech_amd import engine as E

    d = np.load(os.path.join(golden_dir, "kat_word_model_5.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]).cuda() for k in d.files if k.startswith("w::")}
    x = torch.from_numpy(d["x"]).cuda()
    B, T, In = x.shape
    H, N = 64, B * T
    gi = torch.empty(2, N, 3 * H, device="cuda")
    for di, suf in enumerate(("", "_reverse")):
        E.gemm(1, 1, N, 3 * H, In, x.data_ptr(), In, sd["gru.weight_ih_l0" + suf].data_ptr(), In,
               gi.data_ptr() + di * N * 3 * H * 4, 3 * H, bias=sd["gru.bias_ih_l0" + suf].data_ptr())
    out = torch.empty(N, 2 * H, device="cuda")
    lens = torch.full((B,), T, device="cuda", dtype=torch.int32)
    L.call("ss_gru_fwd", gi.data_ptr(), sd["gru.weight_hh_l0"].data_ptr(), sd["gru.weight_hh_l0_reverse"].data_ptr(),
           sd["gru.bias_hh_l0"].data_ptr(), sd["gru.bias_hh_l0_reverse"].data_ptr(), lens.data_ptr(), B, T, H,
           out.data_ptr(), None, None, L.stream())
    torch.cuda.synchronize()
    err = float((out.view(B, T, 2 * H).cpu() - torch.from_numpy(d["gru_out"])).abs().max())
    assert err < 2e-5, err


def test_harness_fit_evaluate_checkpoint(ss, tmp_path):
    """scan -> split -> class-balanced epochs on the device-resident store -> evaluate -> save-best ->
    reload with the reference's loader schema.  Three separable 'words' (a constant offset per class in a few features)."""
    from silent_speech_amd import data as Dm
    from silent_speech_amd import harness as Hn

    rng = np.random.default_rng(0)
    clip_dir = tmp_path / "clips_npz"
    clip_dir.mkdir()
    words = ["aura", "no", "yes"]
    for k in range(45):
        lab = words[k % 3]
        T = int(rng.integers(14, 22))
        X = (0.05 * rng.normal(size=(T, 20))).astype(np.float32)
        X[:, (k % 3) * 4:(k % 3) * 4 + 4] += 0.5
        roi = rng.integers(0, 256, (T, 32, 32), dtype=np.uint8)
        Dm.save_clip(str(clip_dir / f"{k:03d}.npz"), X, np.arange(T), lab, "me", np.arange(4), roi)
    out = str(tmp_path / "word_model_points_roi.pt")
    logs = []
    best = Hn.fit(str(clip_dir), out, epochs=6, batch_size=16, patience=3, max_t=24, lr=3e-3, log=logs.append)
    assert best >= 0.8, (best, logs)
    assert any("saved" in ln for ln in logs) and logs[0].startswith("ep 01 | train loss")
    model, id_to_label, max_t, use_roi = ss.load_classifier(out)
    assert max_t == 24 and use_roi and sorted(id_to_label.values()) == words
    ck = torch.load(out, map_location="cpu", weights_only=False)
    assert {"model", "x_dim", "max_t", "use_roi", "roi_w", "roi_h", "labels", "label_to_id", "id_to_label", "seed"} <= set(ck)
    # evaluate() agrees with the CPU oracle's forward + CE on the validation clips
    info = Hn.scan_clips(str(clip_dir))
    _, val_files = Hn.split_by_label(info["files"], info["labels"], seed=42)
    store = ss.DeviceClipStore(val_files, info["label_to_id"], max_t=24)
    loss, acc, y_true, y_pred = Hn.evaluate(model, store, batch_size=4)
    X, T, R, y = store.batch(range(len(store)))
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = MR.forward(sd, X.cpu(), T.cpu(), R.cpu(), impl="aten")
    ref_loss = float(MR.ce_label_smoothing(ref, y.cpu(), 0.05))
    assert abs(loss - ref_loss) < 1e-4 and y_pred == ref.argmax(1).tolist() and y_true == y.cpu().tolist()
    assert abs(acc - best) < 1e-9


def test_stream_server_matches_the_restated_loop(ss):
    """rings of 7 streams fed with random subsets for 45 ticks == one Python deque per stream
    (oracle/stream_ref.py): who is due, the zero-padded windows (bit-exact), the gate, and the logits of the windows."""
    from oracle import stream_ref as SR

    rng = np.random.default_rng(4)
    S, max_t, Dm, hw = 7, 16, 84, (32, 32)
    sd = W.make_state_dict(3, Dm, 5, True)
    m = ss.BiGRUClassifier(Dm, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    srv = ss.StreamServer(m, S, max_t, roi_hw=hw, cache_embeddings=False)  # (the pixel-level ring: windows are compared as ROI bytes)
    refs = [SR.StreamRef(max_t, Dm, hw) for _ in range(S)]
    n_pred = 0
    for tick in range(45):
        ids = np.flatnonzero(rng.random(S) < 0.8)
        if len(ids) == 0:
            continue
        feats = rng.normal(size=(len(ids), Dm)).astype(np.float32)
        rois = rng.integers(0, 256, (len(ids),) + hw, dtype=np.uint8)
        # float64 openness hovering around the open/close thresholds (0.02): a float32 EMA decides differently here
        op = 0.02 + np.where(rng.random(len(ids)) < 0.5, 1e-9, 0.02) * rng.normal(size=len(ids))
        got = srv.push(ids, torch.from_numpy(feats), torch.from_numpy(rois), torch.from_numpy(op))
        want = {int(s): refs[s].push(feats[k], rois[k], float(op[k])) for k, s in enumerate(ids)}
        due = [s for s, w in want.items() if w is not None]
        if not due:
            assert got is None
            continue
        g_ids, logits, T = got
        assert g_ids.tolist() == due
        X, T2, R = srv.windows(np.asarray(due))
        for k, s in enumerate(due):
            assert np.array_equal(X[k].cpu().numpy(), want[s]["X"]) and int(T[k]) == want[s]["T"] == int(T2[k])
            assert np.array_equal(R[k].cpu().numpy(), want[s]["R"])
        ref = MR.forward(sd, X.cpu(), T.cpu(), R.cpu(), impl="aten")
        assert float((logits.cpu() - ref).abs().max()) < TIGHT
        n_pred += len(due)
        assert np.array_equal(srv.ema.cpu().numpy(), np.asarray([r.ema for r in refs], np.float64))  # bit for bit, float64
        assert srv.mouth_open.cpu().numpy().astype(bool).tolist() == [r.open for r in refs]
    assert n_pred > 40


def test_stream_server_embedding_cache_changes_nothing(ss):
    """A frame's ROI embedding depends on that frame alone (per-frame normalisation, train_model_official.py:286-291), so the
    sliding-window server makes it ONCE, when the frame is pushed, and keeps torch.cat((features, embedding)) in the ring instead
    of the pixels.  Against the server that keeps pixels and recomputes every window: the same streams are due on the same ticks,
    the logits are bit-equal, the cached rows are [features | TinyROICNN(frame)], and faceless ticks (``skip``) count alike."""
    rng = np.random.default_rng(21)
    S, max_t, Dm, hw = 6, 12, 84, (48, 96)
    sd = W.make_state_dict(4, Dm, 5, True)
    m = ss.BiGRUClassifier(Dm, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    cached = ss.StreamServer(m, S, max_t, roi_hw=hw)
    pixels = ss.StreamServer(m, S, max_t, roi_hw=hw, cache_embeddings=False)
    assert cached.cache and cached.ring_r is None and cached.D == Dm + 32 and not pixels.cache
    n_pred = 0
    for tick in range(40):
        ids = np.flatnonzero(rng.random(S) < 0.8)
        gone = np.setdiff1d(np.flatnonzero(rng.random(S) < 0.1), ids)
        cached.skip(gone)
        pixels.skip(gone)
        if len(ids) == 0:
            continue
        feats = torch.from_numpy(rng.normal(size=(len(ids), Dm)).astype(np.float32)).cuda()
        rois = torch.from_numpy(rng.integers(0, 256, (len(ids),) + hw, dtype=np.uint8)).cuda()
        a, b = cached.push(ids, feats, rois), pixels.push(ids, feats, rois)
        assert (a is None) == (b is None)
        if a is None:
            continue
        assert a[0].tolist() == b[0].tolist() and torch.equal(a[2], b[2])
        assert torch.equal(a[1], b[1]), float((a[1] - b[1]).abs().max())
        n_pred += len(a[0])
        if tick % 9 == 0:  # the rows of a cached window
            Z, T, R = cached.windows(a[0])
            Xp, Tp, Rp = pixels.windows(a[0])
            assert R is None and torch.equal(T, Tp) and torch.equal(Z[:, :, :Dm], Xp)
            k, t = 0, int(T[0]) - 1
            assert torch.equal(Z[k, t, Dm:], m.embed_rois(Rp[k, t:t + 1])[0])
            assert float(Z[k, int(T[0]):].abs().max() if int(T[0]) < max_t else 0.0) == 0.0
    assert n_pred > 40
    ref = MR.forward(sd, Xp.cpu(), Tp.cpu(), Rp.cpu(), impl="aten")
    assert float((m.forward_embedded(Z, T).cpu() - ref).abs().max()) < TIGHT
    with pytest.raises(ValueError):
        ss.StreamServer(ss.BiGRUClassifier(Dm, 5, use_roi=False).cuda(), S, max_t, cache_embeddings=True)


def test_live_chain_one_entry_point_band_leave_and_reentry(ss):
    """landmarks + camera frames in, windows + logits out, as ONE device entry point
    (``StreamServer.push_landmarks`` = live_infer_official.py:264-296 per stream: width gate -> extract_feature with the stream's
    ``prev_xy`` -> crop_roi_gray -> buffers; then the sliding-window rule).  The trace leaves the 60-150 px band and comes back:
    frames outside are not appended, ``prev_xy`` is cleared (:295-296) so the first frame after re-entry has velocity exactly 0,
    and the windows equal the per-stream Python restatement (features to the feature kernel's tolerance, ROI bytes bit for bit)."""
    from oracle import features_ref as FR
    from oracle import resize_ref as RR
    from oracle import stream_ref as SR
    from silent_speech_amd import features as Fm

    rng = np.random.default_rng(11)
    idxs = Fm.FIXED_IDXS_88
    K, Dm = len(idxs), 2 * len(idxs) + 4
    S, max_t, hw, w, h = 5, 10, (32, 32), 320, 240
    anchors = Fm.anchor_positions(idxs)
    sd = W.make_state_dict(5, Dm, 10, True)
    m = ss.BiGRUClassifier(Dm, 10, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    srv = ss.StreamServer(m, S, max_t, roi_hw=hw, cache_embeddings=False)
    srv.attach_front_end(idxs, (w, h), variant="live")
    with pytest.raises(ValueError):
        ss.StreamServer(m, S, max_t, roi_hw=hw).attach_front_end(idxs[:40], (w, h))
    refs = [SR.StreamRef(max_t, Dm, hw) for _ in range(S)]
    prev = [None] * S
    base = rng.uniform(0.35, 0.65, (S, K, 2)).astype(np.float32)

    def width_of(s, tick):  # mouth width in pixels the synthetic face is given
        if s == 1 and 8 <= tick < 12:
            return 30.0       # too far away: leaves the band for four ticks, then re-enters
        if s == 2 and tick % 7 == 3:
            return 190.0      # too close, single frames
        if s == 4 and tick < 5:
            return 20.0       # starts outside
        if s == 3:
            return (60.0, 150.0, 59.99, 150.01)[tick % 4]  # on and just off the band's edges
        return 95.0 + 20.0 * np.sin(0.3 * tick + s)

    n_pred, reentry_checked, dropped = 0, 0, 0
    for tick in range(26):
        ids = np.flatnonzero(rng.random(S) < 0.9)
        if len(ids) == 0:
            continue
        lm = (base[ids] + rng.normal(0, 0.003, (len(ids), K, 2))).astype(np.float32)
        for k, s in enumerate(ids):
            half = np.float32(width_of(int(s), tick) / w / 2)
            lm[k, anchors[0]] = [np.float32(0.5) - half, 0.55]
            lm[k, anchors[1]] = [np.float32(0.5) + half, 0.55]
        frames = rng.integers(0, 256, (len(ids), h, w, 3), dtype=np.uint8)
        kept, got = srv.push_landmarks(ids, torch.from_numpy(lm).cuda(), torch.from_numpy(frames).cuda())
        want = {}
        for k, s in enumerate(ids):
            s = int(s)
            mw = FR.mouth_width_px(lm[k, anchors[0]], lm[k, anchors[1]], w, h, "live")
            in_range = Fm.MOUTH_W_MIN_PX <= mw <= Fm.MOUTH_W_MAX_PX
            assert bool(kept[k]) == in_range, (tick, s, mw)
            if not in_range:
                prev[s] = None
                dropped += 1
                continue
            was_reset = prev[s] is None
            feat, prev[s], c, fourth = FR.extract_feature(lm[k], w, h, *anchors, prev_xy=prev[s], variant="live")
            if was_reset:
                assert feat[2 * K] == 0.0
                reentry_checked += 1
            roi = RR.crop_gray_resize(frames[k], FR.crop_box(c, fourth, w, h, "live"), hw[0], hw[1], "area")
            want[s] = refs[s].push(feat, roi)
        due = [s for s, wv in want.items() if wv is not None]
        if not due:
            assert got is None
            continue
        g_ids, logits, T = got
        assert g_ids.tolist() == due
        X, T2, R = srv.windows(np.asarray(due))
        for k, s in enumerate(due):
            assert int(T[k]) == want[s]["T"] == int(T2[k])
            xg, xw = X[k].cpu().numpy(), want[s]["X"]
            np.testing.assert_allclose(xg[:, : 2 * K], xw[:, : 2 * K], rtol=0, atol=1.2e-7)
            np.testing.assert_allclose(xg[:, 2 * K:], xw[:, 2 * K:], rtol=1e-6, atol=1e-7)
            assert np.array_equal(xg[:, 2 * K] == 0.0, xw[:, 2 * K] == 0.0)  # the reset frames: velocity exactly 0 in both
            assert np.array_equal(R[k].cpu().numpy(), want[s]["R"])
        ref = MR.forward(sd, X.cpu(), T.cpu(), R.cpu(), impl="aten")
        assert float((logits.cpu() - ref).abs().max()) < TIGHT
        n_pred += len(due)
    assert n_pred > 20 and reentry_checked >= S + 3 and dropped > 10


def test_mouth_gate_is_float64_near_the_thresholds(ss):
    """the reference's ``mouth_ema`` is a Python float.  Traces built to land the EMA within a few
    float64 ulps of the thresholds -- where a float32 EMA flips the other way -- must give the restated loop's states."""
    from oracle import stream_ref as SR
    from silent_speech_amd import _lib as L

    rng = np.random.default_rng(0)
    S, ticks = 64, 200
    refs = [SR.StreamRef(4, 1) for _ in range(S)]
    ids = torch.arange(S, dtype=torch.int32, device="cuda")
    ema = torch.zeros(S, device="cuda", dtype=torch.float64)
    st = torch.zeros(S, device="cuda", dtype=torch.uint8)
    flips32 = 0
    ema32 = np.zeros(S, np.float32)
    for t in range(ticks):
        # choose the openness that would put the EMA at threshold * (1 + k * 2^-52), k in [-3, 3], for half the streams
        prev = np.asarray([r.ema for r in refs])
        target = 0.02 * (1.0 + rng.integers(-3, 4, S) * 2.0 ** -52)
        op = np.where(rng.random(S) < 0.5, (target - 0.75 * prev) / 0.25, rng.uniform(0.0, 0.05, S))
        for s_, r in enumerate(refs):
            r.push(np.zeros(1, np.float32), None, float(op[s_]))
        opd = torch.from_numpy(op).cuda()
        L.call("ss_mouth_gate", ids.data_ptr(), S, opd.data_ptr(), 0.25, 0.02, 0.02, ema.data_ptr(), st.data_ptr(), L.stream())
        assert np.array_equal(ema.cpu().numpy(), np.asarray([r.ema for r in refs]))
        assert st.cpu().numpy().astype(bool).tolist() == [r.open for r in refs], t
        ema32 = (np.float32(0.75) * ema32 + np.float32(0.25) * op.astype(np.float32)).astype(np.float32)
        flips32 += int(((ema32 > np.float32(0.02)) != (np.asarray([r.ema for r in refs]) > 0.02)).sum())
    assert flips32 > 0, "the traces never separated a float32 EMA from the float64 one"


def test_mouth_openness_kernel(ss):
    """important_landmarks.py:131-133 / live_test_5.py:92-94 on synthetic faces against the literal Python-float restatement:
    the y-range form bit for bit, the eye-span form to one ulp (its ``** 0.5`` is libm pow, the kernel takes sqrt)."""
    from oracle import stream_ref as SR
    from silent_speech_amd.serving import mouth_openness

    rng = np.random.default_rng(2)
    n, K = 300, 478
    lm = rng.uniform(0.2, 0.8, (n, K, 2)).astype(np.float32)
    lm[:5, 14, 1] = lm[:5, 13, 1]          # closed mouth: exactly 0
    lm[5:8, 263] = lm[5:8, 33]             # degenerate eye span: the 1e-6 guard
    got = mouth_openness(torch.from_numpy(lm).cuda(), "eye_span").cpu().numpy()
    want = np.asarray([SR.openness_eye_span(lm[i]) for i in range(n)])
    assert got.dtype == np.float64 and np.all(got[:5] == 0.0)
    assert np.all(np.abs(got - want) <= 2.3e-16 * np.abs(want)), float(np.max(np.abs(got - want) / np.maximum(want, 1e-300)))
    assert (got == want).mean() > 0.99
    idx = rng.choice(K, 88, replace=False)
    sub = np.ascontiguousarray(lm[:, idx])
    got2 = mouth_openness(torch.from_numpy(sub).cuda(), "y_range").cpu().numpy()
    assert np.array_equal(got2, np.asarray([SR.openness_y_range(sub[i]) for i in range(n)]))


def test_mouth_openness_against_the_reference_functions(ss, golden_dir):
    """The three openness signals on the faces of tests/golden/serving.npz, whose expected values the REFERENCE's own functions
    returned (important_landmarks.dist2d + :131-133, inactive/live_test_5.compute_openness, inactive/live_feed.
    extract_83_and_openness): y-range and width-normalised forms bit for bit, the eye-span form to one ulp of a double (the
    reference's ``** 0.5`` is libm pow, the kernel takes the correctly rounded root)."""
    import os

    from silent_speech_amd.serving import mouth_openness

    d = np.load(os.path.join(golden_dir, "serving.npz"))
    lm = torch.from_numpy(d["lm"]).cuda()
    got = mouth_openness(lm, "eye_span", (int(d["mouth_top"]), int(d["mouth_bottom"]), int(d["eye_l"]), int(d["eye_r"]))).cpu().numpy()
    want = d["openness_eye"]
    assert np.all(np.abs(got - want) <= 2.3e-16 * np.abs(want)) and (got == want).mean() > 0.95
    assert got[0] == 0.0
    sub = lm[:, torch.from_numpy(d["idxs"].astype(np.int64)).cuda()].contiguous()
    assert np.array_equal(mouth_openness(sub, "y_range").cpu().numpy(), d["openness_yrange"])
    assert np.array_equal(mouth_openness(lm, "width_norm", (13, 14, 61, 291)).cpu().numpy(), d["openness83"])


@pytest.mark.parametrize("variant", ["live", "record"])
def test_live_front_end_against_the_references_own_loop_statements(ss, golden_dir, variant):
    """tests/golden/live_loop.npz = what the reference's own statements of live_infer_official.py:272-296 did on a 260-frame trace
    (mouth width through and across both band edges, recording switched off and on).  ``LiveFrontEnd`` (``ss_feature_fuse_stream``,
    the state of ``prev_xy`` in HBM) on the same trace: the same frames kept, velocity exactly 0 wherever the reference had
    ``prev_xy is None``, feature rows within the feature kernel's tolerance of the reference's."""
    import os

    from silent_speech_amd import features as Fm
    from silent_speech_amd.serving import LiveFrontEnd

    d = np.load(os.path.join(golden_dir, variant + "_loop.npz"))  # "record": the recorder's loop, record_landmarks_official.py:182-201
    idxs = [int(i) for i in d["idxs"]]
    w, h = (int(v) for v in d["wh"])
    K = len(idxs)
    fe = LiveFrontEnd(1, idxs, (w, h), roi_hw=None, variant=variant)
    lm = torch.from_numpy(d["lm"]).cuda()
    was_rec, n_zero_vel = False, 0
    for f in range(len(d["lm"])):
        rec = bool(d["recording"][f])
        if rec and not was_rec:
            fe.reset([0])  # "r" pressed: live_infer_official.py:334-336
        was_rec = rec
        kept, X, _ = fe([0], lm[f:f + 1], None, recording=torch.tensor([1 if rec else 0], dtype=torch.uint8))
        # the front end reports the band decision; the caller appends when it is recording (the reference's ``recording and in_range``)
        assert bool(kept[0]) == (bool(d["in_range"][f]) and rec) == bool(d["appended"][f]), f
        assert bool(fe.has_prev[0]) == bool(d["has_prev"][f]), f
        if d["appended"][f]:
            got, ref = X[0].cpu().numpy(), d["feats"][f]
            np.testing.assert_allclose(got[: 2 * K], ref[: 2 * K], rtol=0, atol=2e-7)
            np.testing.assert_allclose(got[2 * K:], ref[2 * K:], rtol=5e-7, atol=2e-7)
            if variant == "live":
                assert got[2 * K + 2] == ref[2 * K + 2]  # the mouth width itself: bit for bit
            if ref[2 * K] == 0.0:
                assert got[2 * K] == 0.0
                n_zero_vel += 1
    assert n_zero_vel >= 10 and int(d["appended"].sum()) > 100


def test_serving_kernels_against_the_references_own_loop_statements(ss, golden_dir):
    """SURVEY 8f-4, pinned end to end: tests/golden/serving_loops.npz holds what the reference's OWN statements -- taken out of
    its capture loops and executed frame by frame (make_golden.py:gen_serving_loops) -- did on three traces.  The kernels are
    fed the same traces: (1) ``ss_mouth_openness`` + ``ss_mouth_gate``: the EMA to an ulp (the openness kernel's root is
    correctly rounded, the reference's ``** 0.5`` is not always), every open / close decision identical; (2)
    ``ss_mouth_openness`` (y range) + ``ClipGateServer``: speaking state, counters, which frames end a clip and its length,
    identical, the clip's rows bit-equal; (3) ``StreamServer.push`` / ``skip``: which camera frames predict and the
    zero-padded window handed to the model, identical."""
    import os

    from silent_speech_amd import _lib as L
    from silent_speech_amd.serving import mouth_openness

    d = np.load(os.path.join(golden_dir, "serving_loops.npz"))
    # ---- (1) EMA + hysteresis, important_landmarks.py:130-144
    F = len(d["gate_pts"])
    alpha, thr_open, thr_close = (float(v) for v in d["gate_consts"])
    op = mouth_openness(torch.from_numpy(d["gate_pts"]).cuda(), "eye_span", (0, 1, 2, 3))
    want_op = d["gate_openness"]
    assert np.all(np.abs(op.cpu().numpy() - want_op) <= 2.3e-16 * np.abs(want_op))
    ids = torch.zeros(1, dtype=torch.int32, device="cuda")
    ema = torch.zeros(1, device="cuda", dtype=torch.float64)
    st = torch.zeros(1, device="cuda", dtype=torch.uint8)
    got_ema, got_open = np.zeros(F), np.zeros(F, np.uint8)
    for f in range(F):
        L.call("ss_mouth_gate", ids.data_ptr(), 1, op[f:f + 1].data_ptr(), alpha, thr_open, thr_close, ema.data_ptr(), st.data_ptr(), L.stream())
        got_ema[f], got_open[f] = float(ema[0]), int(st[0])
    assert np.all(np.abs(got_ema - d["gate_ema"]) <= 4.5e-16 * np.abs(d["gate_ema"]))
    assert np.array_equal(got_open, d["gate_open"])

    # ---- (2) openness-gated clips, inactive/live_test_5.py:229-272 and :293-301
    from oracle import stream_ref as SR

    lm = torch.from_numpy(d["clip_lm"]).cuda()
    openv = mouth_openness(lm, "y_range")
    face_rows = ~d["clip_noface"]
    assert np.array_equal(openv.cpu().numpy()[face_rows], d["clip_openv"][face_rows])
    idxs = [int(i) for i in d["clip_idxs"]]
    Dx = 2 * len(idxs) + 1
    m = ss.BiGRUClassifier(Dx, 5, use_roi=False)
    m.cuda().eval()
    thr, start_n, end_n, max_clip = float(d["clip_consts"][0]), *(int(v) for v in d["clip_consts"][1:])
    srv = ss.serving.ClipGateServer(m, 1, open_thresh=thr, start_n=start_n, end_n=end_n, max_clip=max_clip)
    for f in range(len(d["clip_lm"])):
        face = {i: (float(x), float(y)) for i, (x, y) in zip(idxs, d["clip_lm"][f])}
        xvec = SR.face_to_xvec(face, idxs, True)  # (pinned by serving.npz; the gate takes whatever rows it is given)
        row, emit, res = srv.push([0], torch.from_numpy(xvec[None]), openv[f:f + 1],
                                  face_present=torch.tensor([0 if d["clip_noface"][f] else 1], dtype=torch.uint8))
        stt = srv.state[0].cpu().numpy()
        assert (bool(stt[0]), int(stt[1]), int(stt[2])) == (bool(d["clip_speaking"][f]), int(d["clip_above"][f]), int(d["clip_below"][f])), f
        assert int(emit[0]) == int(d["clip_emit_len"][f]), f
        if int(emit[0]):
            clip = srv.clip_x[0, :int(emit[0])].cpu().numpy().astype(np.float64)
            assert abs(float(clip.sum()) - float(d["clip_emit_sum"][f])) < 1e-9 and res is not None and int(res[2][0]) == int(emit[0])
        else:
            assert res is None
    assert int((d["clip_emit_len"] > 0).sum()) >= 8

    # ---- (3) the sliding window, inactive/live_feed.py:173, 197-207
    max_t, warm, every = (int(v) for v in d["win_consts"])
    m83 = ss.BiGRUClassifier(83, 7, use_roi=False)
    m83.cuda().eval()
    win = ss.StreamServer(m83, 1, max_t, pred_every=every, warmup_min=warm)
    feats = torch.from_numpy(d["win_feats"]).cuda()
    for f in range(len(d["win_feats"])):
        src = d["win_src"][f]
        if d["win_noface"][f]:
            win.skip([0])
            continue
        got = win.push([0], feats[f:f + 1])
        assert (got is not None) == bool(src[0] >= 0), f
        if got is not None:
            X, T, _ = win.windows(np.asarray([0]))
            want = np.zeros((max_t, 83), np.float32)
            want[src >= 0] = d["win_feats"][src[src >= 0]]
            assert int(T[0]) == int((src >= 0).sum()) and np.array_equal(X[0].cpu().numpy(), want), f
    assert int(win.frames_seen[0]) == len(d["win_feats"])  # every camera frame counted, with or without a face


def test_clip_gate_server_matches_the_restated_state_machine(ss):
    """inactive/live_test_5.py:233-272 batched over 9 streams: random open/close traces (runs of open frames of random
    length, values within an ulp of OPEN_THRESH, missing faces, clips cut at MAX_CLIP) give the same appends, the same
    finished clips (bit for bit) and logits of those clips within the forward's tolerance."""
    from oracle import stream_ref as SR

    rng = np.random.default_rng(8)
    S, Dm, hw, max_clip = 9, 84, (32, 32), 20
    sd = W.make_state_dict(3, Dm, 5, True)
    m = ss.BiGRUClassifier(Dm, 5, use_roi=True)
    m.load_state_dict(sd)
    m.cuda().eval()
    srv = ss.serving.ClipGateServer(m, S, roi_hw=hw, max_clip=max_clip)
    refs = [SR.ClipGateRef(max_clip=max_clip) for _ in range(S)]
    rrefs = [[] for _ in range(S)]  # ROI frames of the clip being collected, kept beside the restated machine
    level = rng.random(S) < 0.5
    n_clips = n_cut = 0
    for tick in range(260):
        ids = np.flatnonzero(rng.random(S) < 0.9)
        if len(ids) == 0:
            continue
        level ^= rng.random(S) < 0.12  # open/closed runs
        base = np.where(level[ids], 0.3, 0.05)
        near = rng.random(len(ids)) < 0.2
        op = np.where(near, 0.18 * (1.0 + rng.integers(-2, 3, len(ids)) * 2.0 ** -52), base + 0.02 * rng.normal(size=len(ids)))
        face = (rng.random(len(ids)) > 0.02).astype(np.uint8)
        feats = rng.normal(size=(len(ids), Dm)).astype(np.float32)
        rois = rng.integers(0, 256, (len(ids),) + hw, dtype=np.uint8)
        row, emit, res = srv.push(ids, torch.from_numpy(feats), torch.from_numpy(op), torch.from_numpy(rois), torch.from_numpy(face))
        row, emit = row.cpu().numpy(), emit.cpu().numpy()
        done = {}
        for k, s_ in enumerate(ids):
            was_speaking = refs[s_].speaking
            n_before = len(refs[s_].clip_buf)
            appended, clip = refs[s_].push(float(op[k]), feats[k], bool(face[k]))
            if not face[k] or (not was_speaking and refs[s_].speaking):
                rrefs[s_] = []
            assert (row[k] >= 0) == appended and (not appended or row[k] == n_before)
            if appended:
                rrefs[s_].append(rois[k])
            assert int(emit[k]) == (0 if clip is None else len(clip))
            if clip is not None:
                done[int(s_)] = (clip, np.stack(rrefs[s_]))
                n_cut += len(clip) == max_clip
        if not done:
            assert res is None
            continue
        g_ids, logits, T = res
        assert g_ids.tolist() == list(done.keys())
        Xr = np.zeros((len(done), max_clip, Dm), np.float32)
        Rr = np.zeros((len(done), max_clip) + hw, np.uint8)
        for k, (clip, rr) in enumerate(done.values()):
            assert np.array_equal(srv.clip_x[g_ids[k], :len(clip)].cpu().numpy(), clip)
            assert np.array_equal(srv.clip_r[g_ids[k], :len(clip)].cpu().numpy(), rr)
            Xr[k, :len(clip)], Rr[k, :len(clip)] = clip, rr
        ref = MR.forward(sd, torch.from_numpy(Xr), T.cpu(), torch.from_numpy(Rr), impl="aten")
        assert float((logits.cpu() - ref).abs().max()) < TIGHT
        n_clips += len(done)
    assert n_clips >= 10 and n_cut >= 1, (n_clips, n_cut)
    # mis-shaped arguments are turned away on the host: the kernel would index past them
    ok = dict(feats=torch.zeros(2, Dm), openness=torch.zeros(2, dtype=torch.float64), rois=torch.zeros((2,) + hw, dtype=torch.uint8))
    for bad in (dict(feats=torch.zeros(1, Dm)), dict(feats=torch.zeros(2, Dm - 1)), dict(openness=torch.zeros(1, dtype=torch.float64)),
                dict(rois=torch.zeros((2, 16, 16), dtype=torch.uint8)), dict(rois=torch.zeros((2,) + hw)),
                dict(face_present=torch.ones(3, dtype=torch.uint8))):
        with pytest.raises(ValueError):
            srv.push([0, 1], **{**ok, **bad})
    with pytest.raises(ValueError):
        srv.push([0, S], **ok)
