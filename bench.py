#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the fused training step (forward + CE + backward + [all-reduce] +
clip + Adam) on BASELINE.json config 2 -- landmark (K=40 -> D=84) + 64x64 grayscale ROI CNN + 2-layer
BiGRU(192), T=30, batch 256 per GPU, fp32 -- on synthetic clips resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --config 5        # BASELINE config 5: 100 words, 96x96 ROI, CNN 16/32/64/96, BiGRU(512), bf16 MFMA

The default (config 2, one GPU) line also carries ``config4`` (BASELINE config 4: 4 096 x T=60 windows, forward-only,
hipGraph replay, softmax + top-3 on device, with its own roofline) and ``cpu_baseline`` for config 2 at the bench batch
(``also``: config 2 at batch 32 and config 1, landmark-only at batch 8 -- SURVEY.md 8d's three CPU rows).

One JSON line on rank 0.  ``value`` = clips of all ranks / max-over-ranks wall time of exactly K steps.
``roofline`` prices the slowest kernel of the step against the f32-MFMA peak with its ALGORITHMIC FLOPs
(2 x MACs of the contraction it implements) over its HIP-event launch duration.
``cpu_baseline`` = the CPU oracle's train step (same ATen kernels as the reference) on this box's host cores.
"""
import argparse
import gc
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: Peak BF16 MFMA, dense (AMD's 5 PF figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0
C5 = dict(channels=(16, 32, 64, 96), roi=96, emb=64, hidden=512, classes=100)  # BASELINE config 5 (build-defined widths)


def c5_gflop_per_step(B, T, D):
    """Algorithmic GFLOP (2 per MAC) per training step of every kernel family of the config-5 path, keyed by launch tag."""
    N = B * T
    c1, c2, c3, c4 = C5["channels"]
    E, H = C5["emb"], C5["hidden"]
    conv = {1: 96 * 96 * c1 * 9, 2: 48 * 48 * c2 * 9 * c1, 3: 24 * 24 * c3 * 9 * c2, 4: 12 * 12 * c4 * 9 * c3}
    ih = (D + E) + 2 * H  # input widths of the two layers
    f = lambda macs: 2.0 * macs / 1e9
    return {
        "ss_c5_conv12_fwd": f(N * (conv[1] + conv[2])), "ss_c5_conv3_fwd": f(N * conv[3]),
        "ss_c5_conv_last_fwd": f(N * (conv[4] + c4 * E)),
        "ss_c5_conv_last_wgrad": f(N * (conv[4] + c4 * E)), "ss_c5_conv_last_dgrad": f(N * (conv[4] + c4 * E)),
        "ss_c5_conv3_wgrad": f(N * conv[3]), "ss_c5_conv3_dgrad": f(N * conv[3]),
        "ss_c5_conv2_wgrad_rc": f(N * conv[2]), "ss_c5_conv2_dgrad": f(N * conv[2]), "ss_c5_conv1_wgrad": f(N * conv[1]),
        "ss_c5_conv2_dgrad_conv1_wgrad": f(N * (conv[2] + conv[1])),
        "gemm_bf16_ih": f(2 * N * 3 * H * ih), "gemm_bf16_dX": f(2 * N * 3 * H * (E + 2 * H)),
        "gemm_bf16_dW": f(2 * N * 3 * H * ih + 2 * 2 * N * 3 * H * H),
        "ss_gru_bf16_fwd": f(2 * 2 * N * 3 * H * H), "ss_gru_bf16_bwd": f(2 * 2 * N * 3 * H * H),
    }


def c5_step_gflop_per_clip(T, D, mid=128):
    c1, c2, c3, c4 = C5["channels"]
    E, H, C = C5["emb"], C5["hidden"], C5["classes"]
    conv1 = 96 * 96 * c1 * 9
    cnn = conv1 + 48 * 48 * c2 * 9 * c1 + 24 * 24 * c3 * 9 * c2 + 12 * 12 * c4 * 9 * c3 + c4 * E
    gru = 2 * (3 * H * (D + E) + 3 * H * H) + 2 * (3 * H * 2 * H + 3 * H * H)
    fwd = T * (cnn + gru) + 2 * H * mid + mid * C
    return 2.0 * (3 * fwd - T * conv1) / 1e9


def algorithmic_gflop(tag, B, T, D, E, H, C, roi):
    """FLOPs (2 per MAC) one launch of each kernel family implements at this config (per launch, not per step)."""
    N = B * T
    Hh, Ww = roi
    px1, px2, px3 = Hh * Ww, (Hh // 2) * (Ww // 2), (Hh // 4) * (Ww // 4)
    cnn_fwd = 2.0 * N * (px1 * 8 * 9 + px2 * 16 * 72 + px3 * 24 * 144 + 24 * E)
    # backward: dW3 + da2 + dW2 + da1 + dW1 (conv1 has no dX) + fc
    cnn_bwd = 2.0 * N * (2 * px3 * 24 * 144 + 2 * px2 * 16 * 72 + px1 * 8 * 9 + 2 * 24 * E)
    gru_rec = 2.0 * 2 * N * 3 * H * H  # both directions of one layer
    return {
        "ss_roi_cnn_fwd_stash": cnn_fwd / 1e9,
        "ss_roi_cnn_bwd": cnn_bwd / 1e9,
        "ss_gru_fwd": gru_rec / 1e9,
        "ss_gru_bwd": gru_rec / 1e9,
    }.get(tag)


def step_gflop_per_clip(T, D, E, H, C, roi, mid=128, fwd_only=False):
    """Algorithmic FLOPs (2 per MAC) of one clip's forward + backward, SURVEY.md section 8(d): backward = 2 x forward minus the
    input gradient of conv1.  0.591 GFLOP at config 2."""
    Hh, Ww = roi
    px1, px2, px3 = Hh * Ww, (Hh // 2) * (Ww // 2), (Hh // 4) * (Ww // 4)
    conv1 = px1 * 8 * 9
    cnn = conv1 + px2 * 16 * 72 + px3 * 24 * 144 + 24 * E
    gru = 2 * (3 * H * (D + E) + 3 * H * H) + 2 * (3 * H * 2 * H + 3 * H * H)  # both directions, two layers, per frame
    head = 2 * H * mid + mid * C
    fwd = T * (cnn + gru) + head
    if fwd_only:
        return 2.0 * fwd / 1e9
    return 2.0 * (3 * fwd - T * conv1) / 1e9


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box gives one
    GPU's share of a large host; os.cpu_count() would oversubscribe it many times over)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("SS_BENCH_CPU_THREADS", "16"))))


def _cpu_time_steps(sd, X, L, R, y, seconds, max_steps):
    from oracle import model_ref as MR

    state = {}
    t0 = time.perf_counter()
    MR.train_step(sd, state, X, L, R, y, impl="aten")  # warm-up
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = 0
    while True:
        MR.train_step(sd, state, X, L, R, y, impl="aten")
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= max_steps or (n == 1 and max(warm, el) > seconds):
            break
    return n, el


def cpu_baseline(args, D, C, config):
    """Oracle (``port``: the reference's own ATen CPU kernels behind the restated module) timed on the host cores, bounded
    samples.  The headline object is the bench's own workload at its own batch (a few steps); SURVEY.md 8(d)'s other two
    rows -- BASELINE config 1 (landmark-only, B = 8) and config 2 at B = 32 -- ride along under ``also``."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W

    cores = host_cores()
    torch.set_num_threads(cores)
    T = args.frames
    what = f"dropout off) of the CPU oracle, torch {torch.__version__} ATen CPU kernels, {cores} threads"
    if config == 5:
        kw = dict(roi_emb=C5["emb"], hidden=C5["hidden"], cnn_channels=C5["channels"])
        Bc = min(args.batch, 32)
        sd = W.make_state_dict(0, D, C, True, **kw)
        X, L, R, y = W.make_inputs(0, Bc, T, D, C, (96, 96), lengths=[T] * Bc)
        n, el = _cpu_time_steps(sd, X, L, R, y, args.cpu_seconds, 5)
        return {"value": round(Bc * n / el, 2), "unit": "clips/s", "cores": cores, "kind": "port",
                "sample": f"{n} f32 train steps (fwd+CE+bwd+clip+Adam, {what}; config-5 widths (96x96 ROI, CNN 16/32/64/96, H=512, "
                          f"100 words) at batch {Bc}, T={T}"}
    out = None
    also = []
    for name, Bc, use_roi, secs, cap in (("config 2 at the bench batch", args.batch, True, args.cpu_seconds, 4),
                                          ("config 2 at batch 32", args.cpu_batch, True, args.cpu_seconds / 3, 12),
                                          ("config 1: landmark-only GRU, batch 8", 8, False, args.cpu_seconds / 5, 40)):
        sd = W.make_state_dict(0, D, C, use_roi)
        X, L, R, y = W.make_inputs(0, Bc, T, D, C, (args.roi, args.roi) if use_roi else None, lengths=[T] * Bc)
        n, el = _cpu_time_steps(sd, X, L, R, y, secs, cap)
        rec = {"value": round(Bc * n / el, 2), "unit": "clips/s", "cores": cores, "kind": "port",
               "sample": f"{n} train steps (fwd+CE+bwd+clip+Adam, {what}; {name}, T={T}"
                         + (f", {args.roi}x{args.roi} ROI" if use_roi else ""), "ms_per_step": round(1000 * el / n, 1)}
        if out is None:
            out = rec
        else:
            also.append(rec)
    out["also"] = also
    return out


def config4_block(ss, dev, D, C, roi, steps, warmup):
    """BASELINE config 4 inside the default run: 4 096 sliding 60-frame windows, forward-only, one hipGraph replay per step
    (softmax + top-3 included).  Whole-forward roofline: 0.4056 GFLOP per window against the f32-MFMA peak."""
    B, T = 4096, 60
    g = torch.Generator(device=dev).manual_seed(4321)
    X = torch.randn(B, T, D, device=dev, generator=g)
    R = torch.randint(0, 256, (B, T, roi, roi), device=dev, dtype=torch.uint8, generator=g)
    lengths = torch.full((B,), T, device=dev, dtype=torch.int64)
    model = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).eval()
    gi = ss.GraphedInference(model, B, T, (roi, roi), topk=3)
    gi(X, lengths, R)
    for _ in range(warmup):
        gi()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gi()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gpw = step_gflop_per_clip(T, D, 32, 192, C, (roi, roi), fwd_only=True)
    tf = B * steps / el * gpw / 1e3
    del gi, model, X, R
    torch.cuda.empty_cache()
    return {"metric": "windows/sec (60-frame, forward-only, hipGraph, softmax + top-3 on device)", "value": round(B * steps / el, 1),
            "unit": "windows/s", "steps": steps, "warmup": warmup, "ms_per_step": round(1000 * el / steps, 3), "dtype": "f32",
            "config": {"workload": f"BASELINE config 4: {B} sliding windows x T={T}, landmark + {roi}x{roi} ROI CNN + BiGRU, forward-only, hipGraph replay"},
            "roofline": {"bound": "mfma", "gflop_per_window": round(gpw, 4), "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None}}



PROFILE_NAME = re.compile(r"^round(\d+)_([a-z0-9]+)_(?:(c5|ship)_)?pmc_(traffic|mfma|issue)\.json$")


def latest_profile(kind, block="config2", names=None):
    """Newest committed PMC summary of ``kind`` ("traffic" | "mfma" | "issue") that was TAKEN ON ``block``'s workload:
    ``round<N>_<tag>_pmc_<kind>.json`` = the headline (config 2, 64x64 ROI), ``..._c5_pmc_...`` = config 5, ``..._ship_pmc_...`` =
    the reference's shipped shape (48x96 ROI, T = 90).  The name must match exactly -- a profile of another workload is never a
    stand-in (round 3's driver line carried the 48x96 kernel's counters for config 2 because ``_ship_`` sorted last).
    ``names`` (a directory listing) is for the CPU test."""
    pdir = os.path.join(ROOT, "profiles")
    if names is None:
        names = os.listdir(pdir) if os.path.isdir(pdir) else []
    want = {"config2": None, "config5": "c5", "shipped": "ship"}[block]
    best = None
    for n in names:
        m = PROFILE_NAME.match(n)
        if m and m.group(3) == want and m.group(4) == kind:
            key = (int(m.group(1)), m.group(2))
            if best is None or key > best[0]:
                best = (key, n)
    return os.path.join(pdir, best[1]) if best else None


def pmc_lookup(path, stems):
    """Entry of a tools/pmc_summary.py JSON whose kernel name contains every string of ``stems`` -- exactly one kernel may
    match (two instantiations of one template in a file is a lookup that needs a sharper stem, not an average)."""
    if not path or not os.path.exists(path):
        return None
    want = [st for st in stems if not st.startswith("!")]
    never = [st[1:] for st in stems if st.startswith("!")]  # "!text": the name must NOT contain text
    hits = [d for name, d in json.load(open(path))["kernels"].items()
            if all(st in name for st in want) and not any(st in name for st in never)]
    return hits[0] if len(hits) == 1 else None


def kernel_stems(tag, roi_hw=None):
    """Substrings a profile's kernel name must contain for launch tag ``tag``; the ROI kernels are templated on the frame
    geometry, so the geometry is part of the stem: a profile of another frame size yields None, not another kernel's numbers."""
    st = list(KERNEL_STEMS.get(tag, [tag]))
    if tag in ("ss_roi_cnn_bwd", "ss_roi_cnn_fwd_stash") and roi_hw is not None:
        st.append("Geom<%d, %d>" % tuple(roi_hw))
    if tag == "ss_roi_cnn_bwd":
        # two kernels since round 4: <Geom, true> walks a frame list (ragged batches, and the first step of any run), <Geom, false>
        # all frames -- the one the full clips of every bench workload run
        st.append("!, true>")
    return st


# launch tag -> substrings of the kernel's name in the rocprofv3 summaries
KERNEL_STEMS = {
    "ss_roi_cnn_bwd": ["roi_cnn_bwd_kernel"], "ss_roi_cnn_fwd_stash": ["roi_cnn_fwd_kernel"], "ss_gru_fwd": ["gru_split_fwd_kernel"],
    "ss_gru_bwd": ["gru_split_bwd_kernel"], "ss_gru_bf16_fwd": ["gru_pers_fwd_kernel"], "ss_gru_bf16_bwd": ["gru_pers_bwd_kernel"],
    "ss_c5_conv12_fwd": ["conv12_fwd_kernel"], "ss_c5_conv1_wgrad": ["conv1_wgrad_kernel"], "ss_c5_conv2_wgrad_rc": ["conv_wgrad_kernel<16, 32"],
    "ss_c5_conv2_dgrad": ["conv_dgrad_kernel<16, 32"], "ss_c5_conv2_dgrad_conv1_wgrad": ["conv2_dgrad_w1_kernel"], "ss_c5_conv3_wgrad": ["conv_wgrad_kernel<32, 64"],
    "ss_c5_conv3_dgrad": ["conv_dgrad_kernel<32, 64"], "ss_c5_conv_last_wgrad": ["conv_wgrad_kernel<64, 96"],
    "ss_c5_conv_last_dgrad": ["conv_dgrad_kernel<64, 96"], "ss_c5_conv3_fwd": ["conv_fwd_kernel<32, 64"],
    "ss_c5_conv_last_fwd": ["conv_fwd_kernel<64, 96"], "gemm_bf16_dW": ["gemm_bf16_kernel<0, 0>"],
}


def synth_inputs(L, dev, rank, B, T, K, roi_hw, C, with_landmarks=False):
    """Synthetic clips of SURVEY.md 8(d), generated on the device: jittered landmarks -> ss_feature_fuse, uint8 ROI, full lengths."""
    D = 2 * K + 4
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    base = torch.rand(B, 1, K, 2, device=dev, generator=g) * torch.tensor([0.4, 0.4], device=dev) + torch.tensor([0.3, 0.4], device=dev)
    lm = base + 0.004 * torch.randn(B, T, K, 2, device=dev, generator=g)
    lm[:, :, 8] = torch.tensor([0.42, 0.61], device=dev)   # landmark 61 sits at position 8 of the sorted 40-lip list
    lm[:, :, 25] = torch.tensor([0.58, 0.61], device=dev)  # 291
    lm[:, :, 1] = torch.tensor([0.50, 0.59], device=dev)   # 13
    lm[:, :, 2] = torch.tensor([0.50, 0.63], device=dev)   # 14
    lm = (lm + 0.002 * torch.randn(B, T, K, 2, device=dev, generator=g)).contiguous()
    X = torch.empty(B, T, D, device=dev)
    L.call("ss_feature_fuse", lm.data_ptr(), None, B, T, K, 640, 480, 8, 25, 1, 2, 0, X.data_ptr(), D, None, None, L.stream())
    R = torch.randint(0, 256, (B, T, roi_hw[0], roi_hw[1]), device=dev, dtype=torch.uint8, generator=g)
    lengths = torch.full((B,), T, device=dev, dtype=torch.int64)
    y = torch.randint(0, C, (B,), device=dev, generator=g)
    if with_landmarks:
        return X, lengths, R, y, lm
    return X, lengths, R, y


def spec_for(config, B, T, K, roi, C):
    """What a training block needs to know about a BASELINE config: model arguments, dtype, roofline denominator, FLOP counts."""
    D = 2 * K + 4
    if config == 5:
        E, H = C5["emb"], C5["hidden"]
        return dict(config=5, B=B, T=T, K=K, D=D, C=C, roi_hw=(C5["roi"], C5["roi"]), E=E, H=H, dtype="bf16", peak=BF16_MFMA_PEAK_TFLOPS,
                    model_kw=dict(roi_emb=E, hidden=H, cnn_channels=C5["channels"], precision="bf16"),
                    gflop_per_clip=c5_step_gflop_per_clip(T, D),
                    workload=(f"BASELINE config 5: landmark (K={K}, D={D}) + 96x96 uint8 ROI CNN (16,32,64,96) + 2-layer BiGRU({H}), T={T}, "
                              f"C={C} words, bf16 MFMA operands / f32 accumulation and master weights, train step = fwd + CE(ls .05) + "
                              "bwd + grad all-reduce + clip(1.0) + Adam, dropout on"))
    roi_hw = roi if isinstance(roi, tuple) else (roi, roi)
    E, H = 32, 192
    name = "BASELINE config 2" if config == 2 else "the reference's shipped configuration (train_model_official.py:29-38)"
    return dict(config=config, B=B, T=T, K=K, D=D, C=C, roi_hw=roi_hw, E=E, H=H, dtype="f32", peak=F32_MFMA_PEAK_TFLOPS, model_kw={},
                gflop_per_clip=step_gflop_per_clip(T, D, E, H, C, roi_hw),
                workload=(f"{name}: landmark (K={K}, D={D}) + {roi_hw[0]}x{roi_hw[1]} uint8 ROI CNN + 2-layer BiGRU(192), T={T}, C={C}, "
                          "train step = fwd + CE(ls .05) + bwd + grad all-reduce + clip(1.0) + Adam, dropout on"))


def train_block(ss, L, dev, world, rank, spec, X, lengths, R, y, steps, warmup, kernel_times, micro_batches, dist_on, extra_windows=0):
    """W warm-up steps, then exactly K timed steps of the fused training step between barriers (max over ranks); then K more
    steps on one stream with a HIP-event pair around every launch for the per-kernel table and the dominant kernel's roofline."""
    import torch.distributed as dist

    B, T, D, C, E, H = spec["B"], spec["T"], spec["D"], spec["C"], spec["E"], spec["H"]
    torch.manual_seed(0)
    model = ss.BiGRUClassifier(D, C, use_roi=True, **spec["model_kw"]).to(dev).train()
    if dist_on:
        dist.broadcast(model.flat_params, src=0)
    trainer = ss.Trainer(model, world_size=world, micro_batches=micro_batches, always_allreduce=dist_on)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    gb = B * world
    for _ in range(warmup):
        trainer.step(X, lengths, R, y, global_batch=gb)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = trainer.step(X, lengths, R, y, global_batch=gb)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    final_loss = float(loss)
    assert final_loss == final_loss, "loss is NaN"
    model.check_health()
    # run-to-run spread: four more windows of K steps, timed the same way (``value`` stays the contract's first window)
    windows = [elapsed]
    for _ in range(extra_windows):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.step(X, lengths, R, y, global_batch=gb)
        barrier()
        w = time.perf_counter() - t0
        if dist_on:
            tw = torch.tensor([w], device=dev, dtype=torch.float64)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            w = float(tw)
        windows.append(w)

    kernels, roof, kernel_tf, allreduce_ms = {}, None, None, None
    if kernel_times:
        from silent_speech_amd import engine, engine_bf16

        engine.USE_SIDE_STREAM = engine_bf16.USE_SIDE_STREAM = False  # one stream, so each event pair brackets exactly one kernel
        mb, trainer.micro_batches = trainer.micro_batches, 1
        L.PROFILE = {}
        trainer.allreduce_events = [] if dist_on else None
        for _ in range(steps):
            trainer.step(X, lengths, R, y, global_batch=gb)
        torch.cuda.synchronize()
        prof, L.PROFILE = L.PROFILE, None
        engine.USE_SIDE_STREAM = engine_bf16.USE_SIDE_STREAM = True
        trainer.micro_batches = mb
        if trainer.allreduce_events:
            allreduce_ms = sum(a.elapsed_time(b) for a, b in trainer.allreduce_events) / len(trainer.allreduce_events)
        trainer.allreduce_events = None
        for tag, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            kernels[tag] = {"launches_per_step": len(ms) / steps, "avg_ms": sum(ms) / len(ms), "ms_per_step": sum(ms) / steps}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        full_size = (B, T) == (256, 30)
        if spec["config"] == 5:
            gfs = c5_gflop_per_step(B, T, D)
            gf = gfs.get(dom)
            if gf is not None:
                achieved = gf / kernels[dom]["ms_per_step"]  # GFLOP per step / ms per step = TFLOP/s
                tr_file = latest_profile("traffic", "config5") if full_size else None
                tr_ = pmc_lookup(tr_file, kernel_stems(dom))
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                        "traffic": int(tr_["hbm_bytes_per_launch"] * kernels[dom]["launches_per_step"]) if tr_ else None,
                        "algorithmic_gflop_per_step": round(gf, 2), "launches_per_step": kernels[dom]["launches_per_step"],
                        "ms_per_step": round(kernels[dom]["ms_per_step"], 4)}
                mb_file = latest_profile("mfma", "config5") if full_size else None
                mb_ = pmc_lookup(mb_file, kernel_stems(dom))
                if mb_:
                    roof["mfma_busy_frac"] = mb_.get("mfma_busy_frac")
                roof["profile_files"] = [os.path.basename(f) for f, d in ((tr_file, tr_), (mb_file, mb_)) if d]
            kernel_tf = {k: round(gfs[k] / v["ms_per_step"], 1) for k, v in kernels.items() if k in gfs}
        else:
            gf = algorithmic_gflop(dom, B, T, D, E, H, C, spec["roi_hw"])
            if gf is not None:
                achieved = gf / kernels[dom]["avg_ms"]  # GFLOP / ms = TFLOP/s
                # counters exist for the headline (B = 256, T = 30, 64x64) and for the shipped shape at batch 256 (T = 90, 48x96)
                block = "config2" if spec["config"] == 2 else "shipped"
                profiled = (B, T) == ((256, 30) if block == "config2" else (256, SHIPPED["T"]))
                stems = kernel_stems(dom, spec["roi_hw"])
                tr_file = latest_profile("traffic", block) if profiled else None
                if block == "shipped" and tr_file and "batch 256 only" not in json.load(open(tr_file)).get("source", ""):
                    tr_file = None  # round 3's ship summaries average the batch-16 and batch-256 launches of one kernel name
                tr_ = pmc_lookup(tr_file, stems)
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": int(tr_["hbm_bytes_per_launch"]) if tr_ else None,
                        "algorithmic_gflop_per_launch": round(gf, 3), "avg_launch_ms": round(kernels[dom]["avg_ms"], 4)}
                mb_file = latest_profile("mfma", block) if profiled else None
                if block == "shipped" and mb_file and "batch 256 only" not in json.load(open(mb_file)).get("source", ""):
                    mb_file = None
                mb_ = pmc_lookup(mb_file, stems)
                if mb_:
                    roof["mfma_busy"] = mb_
                roof["profile_files"] = [os.path.basename(f) for f, d in ((tr_file, tr_), (mb_file, mb_)) if d]

    out = {
        "metric": "clips/sec (%d-frame, fwd+bwd)" % T, "value": round(B * world * steps / elapsed, 1), "unit": "clips/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(1000 * elapsed / steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": spec["dtype"], "data": "synthetic",
        "config": {"workload": spec["workload"], "batch_per_gpu": B, "global_batch": B * world, "frames": T, "parallelism": f"dp{world}",
                   "micro_batches_in_flight": micro_batches},
        "final_loss": round(final_loss, 5),
    }
    if len(windows) > 1:
        ms = sorted(1000 * w / steps for w in windows)
        out["extra"] = {"windows": len(ms), "steps_per_window": steps, "ms_per_step_windows": [round(1000 * w / steps, 4) for w in windows],
                        "ms_per_step_median": round(ms[len(ms) // 2], 4), "ms_per_step_min": round(ms[0], 4),
                        "clips_per_s_median": round(B * world / ms[len(ms) // 2] * 1e3, 1), "clips_per_s_best": round(B * world / ms[0] * 1e3, 1)}
    if roof:
        out["roofline"] = roof
    # whole-step figure of SURVEY.md 8(d): clips/s x algorithmic GFLOP per clip against the dense MFMA peak (of the dtype) of the GPUs used
    tf = out["value"] * spec["gflop_per_clip"] / 1e3
    out["step_roofline"] = {"bound": "mfma", "gflop_per_clip": round(spec["gflop_per_clip"], 4), "achieved": round(tf, 2),
                            "peak": round(spec["peak"] * world, 1), "unit": "TFLOP/s", "frac": round(tf / (spec["peak"] * world), 4)}
    if "ragged" in spec:  # --min-len-frac: the work figures above count all B*T frames, the kernels walked only the clips' own
        for k in ("roofline", "step_roofline"):
            if k in out:
                out[k]["note"] = "algorithmic work counts all B*T frames; %s" % spec["ragged"]
    if kernels:
        out["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        if kernel_tf:
            out["kernels_tflops"] = kernel_tf
    if allreduce_ms is not None:
        out["allreduce_ms"] = round(allreduce_ms, 4)  # the flat-bucket all-reduce (RCCL), HIP events around the collective
        out["allreduce_bytes"] = int(model.flat_grads.numel()) * 4
    del trainer, model
    torch.cuda.empty_cache()
    return out


SHIPPED = dict(K=88, T=90, roi_hw=(48, 96), C=10)  # train_model_official.py:29-38, record_landmarks_official.py:30-44: D = 180


def ragged(lengths, T, frac, spec):
    """--min-len-frac f < 1: lengths uniform in [ceil(f*T), T] (clip 0 keeps T), and the workload string says so."""
    if frac >= 1.0:
        return lengths
    lo = max(1, -int(-frac * T // 1))
    g = torch.Generator(device=lengths.device).manual_seed(99)
    out = torch.randint(lo, T + 1, lengths.shape, device=lengths.device, generator=g, dtype=torch.int64)
    out[0] = T
    spec["ragged"] = "clip lengths uniform in [%d, %d], %.0f %% of the B*T frames inside a clip" % (
        lo, T, 100.0 * float(out.sum()) / (T * out.numel()))
    spec["workload"] += "; " + spec["ragged"]
    return out


def stream_block(ss, dev, D, C, roi, S, Ts, steps):
    """S camera streams, one new frame per stream and tick, a prediction per stream every 2nd tick on its zero-padded
    sliding window (inactive/live_feed.py:155-213) -- rings, window assembly and forward all on the device.  Two servers on the
    same inputs: the one that ships (a frame's ROI embedding made once, when it is pushed, and kept in the ring) and the one that
    keeps pixels and runs the CNN over every window again -- the same logits either way (tests/test_gpu_model.py)."""
    model = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).eval()
    ids = list(range(S))
    feats = torch.randn(S, D, device=dev)
    rois = torch.randint(0, 256, (S, roi, roi), device=dev, dtype=torch.uint8)
    op = torch.rand(S, device=dev) * 0.05
    res, el, n_keep = {}, 1.0, 0
    for name, cache in (("embeddings_cached", True), ("every_window_from_pixels", False)):
        srv = ss.StreamServer(model, S, Ts, roi_hw=(roi, roi), device=dev, cache_embeddings=cache)
        for _ in range(Ts + (Ts % 2)):  # fill the rings; ends on an even frame count
            srv.push(ids, feats, rois, op)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_pred = 0
        for _ in range(2 * steps):
            got = srv.push(ids, feats, rois, op)
            n_pred += 0 if got is None else len(got[0])
        torch.cuda.synchronize()
        el_ = time.perf_counter() - t0
        res[name] = {"windows_per_sec": round(n_pred / el_, 1), "ms_per_tick_pair": round(1000 * el_ / steps, 3)}
        if cache:
            el, n_keep = el_, n_pred
        del srv
        torch.cuda.empty_cache()
    return {"metric": "windows/sec (sliding %d-frame windows of %d streams, push + assemble + forward)" % (Ts, S),
            "value": round(n_keep / el, 1), "unit": "windows/s", "n_gpus": 1, "steps": steps, "warmup": 0,
            "ms_per_step": round(1000 * el / steps, 3), "higher_is_better": True, "dtype": "f32",
            "data": "synthetic", "frames_ingested_per_sec": round(2 * steps * S / el, 1), "variants": res,
            "config": {"workload": "SURVEY 8f-4: %d streams x T=%d, one frame per stream and tick, prediction every "
                                   "2nd tick, landmark + %dx%d ROI CNN + BiGRU forward" % (S, Ts, roi, roi)}}


def padded_block(ss, L, dev, B, T, K, roi, C, X, R, y, steps=20, warmup=5):
    """The headline's workload with clips of unequal length, as a recorded data set has them (the reference pads every clip to T
    frames, train_model_official.py:93-172): lengths uniform in [0.3 T, T].  Timed twice on the same inputs: the fused ROI-CNN
    kernels walking only the frames inside their clips (the default), and every one of the B*T frames as the reference computes
    them (SS_CNN_SKIP_PADDING=0).  Same logits and gradients either way (tests/test_gpu_model.py); not the headline."""
    from silent_speech_amd import engine as E

    spec = spec_for(2, B, T, K, roi, C)
    lengths = ragged(torch.full((B,), T, device=dev, dtype=torch.int64), T, 0.3, spec)
    res = {"lengths": spec["ragged"], "unit": "clips/s", "steps": steps, "warmup": warmup}
    keep = E.SKIP_PADDED_FRAMES
    gc.collect()  # (the blocks before this one left models and workspaces behind: freed now, not inside a timed window)
    torch.cuda.empty_cache()
    try:
        for name, skip in (("frames_inside_clips_only", True), ("every_frame", False)):
            E.SKIP_PADDED_FRAMES = skip  # (read when a workspace is built: each model below builds its own)
            torch.manual_seed(0)
            model = ss.BiGRUClassifier(spec["D"], C, use_roi=True).to(dev).train()
            trainer = ss.Trainer(model)
            for _ in range(warmup):
                trainer.step(X, lengths, R, y)
            wins = []
            for _ in range(3):  # three windows, the median reported: one collector pass of the host inside a window is 1 ms per step
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    trainer.step(X, lengths, R, y)
                torch.cuda.synchronize()
                wins.append(1000 * (time.perf_counter() - t0) / steps)
            ms = sorted(wins)[1]
            res[name] = {"value": round(B / ms * 1e3, 1), "ms_per_step": round(ms, 3), "ms_per_step_windows": [round(v, 3) for v in wins]}
            E.USE_SIDE_STREAM = False  # ten more steps on one stream with an event pair around every launch: the two CNN kernels' share
            L.PROFILE = {}
            for _ in range(10):
                trainer.step(X, lengths, R, y)
            torch.cuda.synchronize()
            prof, L.PROFILE = L.PROFILE, None
            E.USE_SIDE_STREAM = True
            res[name]["cnn_kernels_ms_per_step"] = {k: round(sum(a.elapsed_time(b) for a, b in prof[k]) / 10, 4)
                                                    for k in ("ss_roi_cnn_fwd_stash", "ss_roi_cnn_bwd", "ss_train_prologue") if k in prof}
            del trainer, model
            gc.collect()
            torch.cuda.empty_cache()
    finally:
        E.SKIP_PADDED_FRAMES = keep
    return res


def shipped_blocks(ss, L, dev, world, rank, args, dist_on):
    """The configuration the reference actually runs: BATCH_SIZE 16 (as shipped) and 256 clips of MAX_T = 90 frames, 88 landmarks
    (D = 180), ROI 96 wide x 48 high, 10 words."""
    out = {}
    for B in [int(v) for v in args.shipped_batches.split(",")]:
        name = "b%d" % B
        sp = spec_for("shipped", B, SHIPPED["T"], SHIPPED["K"], SHIPPED["roi_hw"], SHIPPED["C"])
        X, l, R, y = synth_inputs(L, dev, rank, B, sp["T"], sp["K"], sp["roi_hw"], sp["C"])
        l = ragged(l, sp["T"], args.min_len_frac, sp)
        out[name] = train_block(ss, L, dev, world, rank, sp, X, l, R, y, args.steps, max(2, args.warmup // 2),
                                not args.no_kernel_times, 1, dist_on)
        del X, R
        torch.cuda.empty_cache()
    return out


def cpu_baseline_c5(args, D, T, seconds):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W

    cores = host_cores()
    torch.set_num_threads(cores)
    kw = dict(roi_emb=C5["emb"], hidden=C5["hidden"], cnn_channels=C5["channels"])
    Bc = 32
    sd = W.make_state_dict(0, D, C5["classes"], True, **kw)
    X, L_, R, y = W.make_inputs(0, Bc, T, D, C5["classes"], (96, 96), lengths=[T] * Bc)
    n, el = _cpu_time_steps(sd, X, L_, R, y, seconds, 5)
    return {"value": round(Bc * n / el, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{n} f32 train steps (fwd+CE+bwd+clip+Adam, dropout off) of the CPU oracle, torch {torch.__version__} ATen CPU "
                      f"kernels, {cores} threads; config-5 widths (96x96 ROI, CNN 16/32/64/96, H=512, 100 words) at batch {Bc}, T={T}",
            "ms_per_step": round(1000 * el / n, 1)}


def cpu_baseline_shipped(seconds):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W

    cores = host_cores()
    torch.set_num_threads(cores)
    D, T, C, Bc = 2 * SHIPPED["K"] + 4, SHIPPED["T"], SHIPPED["C"], 16
    sd = W.make_state_dict(0, D, C, True)
    X, L_, R, y = W.make_inputs(0, Bc, T, D, C, SHIPPED["roi_hw"], lengths=[T] * Bc)
    n, el = _cpu_time_steps(sd, X, L_, R, y, seconds, 6)
    return {"value": round(Bc * n / el, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps (fwd+CE+bwd+clip+Adam, dropout off) of the CPU oracle, torch {torch.__version__} ATen CPU kernels, "
                      f"{cores} threads; the shipped configuration: batch 16, T=90, D=180, 48x96 ROI, 10 words",
            "ms_per_step": round(1000 * el / n, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU (weak scaling)")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--landmarks", type=int, default=40)
    ap.add_argument("--roi", type=int, default=64)
    ap.add_argument("--min-len-frac", type=float, default=1.0,
                    help="train mode: clip lengths uniform in [ceil(f*T), T] instead of all T (the headline is quoted on full clips; "
                         "the roofline objects of such a run still count B*T frames)")
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-times", action="store_true")
    ap.add_argument("--mode", choices=["train", "infer", "assemble", "stream", "crop", "live"], default="train",
                    help="train = the headline metric; infer = BASELINE config 4 (T=60, B=4096 windows, forward-only, hipGraph)")
    ap.add_argument("--micro-batches", type=int, default=1, help="slices of the per-GPU batch kept in flight on separate streams")
    ap.add_argument("--config", type=int, choices=[2, 5], default=2,
                    help="2 = the headline (f32, 64x64 ROI, H=192, 5 words); 5 = BASELINE config 5 (bf16 MFMA, 96x96 ROI, CNN "
                         "16/32/64/96, H=512, 100 words)")
    ap.add_argument("--no-padded", action="store_true", help="skip the padded-batches block of the default run")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 (hipGraph serving) block of the default run")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 (bf16, 96x96 ROI, H=512) block of the default run")
    ap.add_argument("--no-shipped", action="store_true", help="skip the block at the reference's shipped configuration")
    ap.add_argument("--shipped", dest="shipped_only", action="store_true",
                    help="only the reference's shipped configuration (48x96 ROI, 88 landmarks -> D=180, 10 words, T=90; batch 16 as "
                         "shipped and batch 256), one JSON line with its CPU row")
    ap.add_argument("--extra-windows", type=int, default=4,
                    help="further timed windows of --steps steps behind the contract's one: median / min under 'extra'")
    ap.add_argument("--shipped-batches", default="16,256", help="batches of the shipped-configuration block (a profile takes 256 only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="join an 'nccl' (= RCCL) process group and run the data-parallel leg -- broadcast of the parameters, the flat "
                         "gradient all-reduce every step, barriers, max-over-ranks timing -- even with one rank")
    args = ap.parse_args()
    if args.config == 5:
        args.roi, args.classes = C5["roi"], C5["classes"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    import silent_speech_amd as ss
    from silent_speech_amd import _lib as L

    if args.shipped_only:
        out = shipped_blocks(ss, L, dev, world, rank, args, dist_on)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_shipped(seconds=args.cpu_seconds)
        if dist_on:
            dist.destroy_process_group()
        if rank == 0:
            first = "b16" if "b16" in out else sorted(out)[0]
            head = dict(out[first])  # the contract's keys describe the batch the reference ships; the others ride along
            for k in out:
                if k.startswith("b") and k != first:
                    head[k] = out[k]
            if "cpu_baseline" in out:
                head["cpu_baseline"] = out["cpu_baseline"]
            print(json.dumps(head), flush=True)
        return
    if args.mode == "infer" and args.batch == 256 and args.frames == 30:
        args.batch, args.frames = 4096, 60
    B, T, K, C, roi = args.batch, args.frames, args.landmarks, args.classes, args.roi
    D = 2 * K + 4
    X, lengths, R, y, lm = synth_inputs(L, dev, rank, B, T, K, (roi, roi), C, with_landmarks=True)

    torch.manual_seed(0)
    if args.mode == "infer":
        model = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).eval()
        g_inf = ss.GraphedInference(model, B, T, (roi, roi))
        g_inf(X, lengths, R)
        for _ in range(args.warmup):
            g_inf()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g_inf()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps({"metric": "windows/sec (60-frame, forward-only, hipGraph)", "value": round(B * args.steps / el, 1),
                          "unit": "windows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1000 * el / args.steps, 3), "higher_is_better": True, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": f"BASELINE config 4: {B} sliding windows x T={T}, "
                                                          f"landmark + {roi}x{roi} ROI CNN + BiGRU, forward-only, hipGraph replay"}}))
        return
    if args.mode == "crop":
        # landmarks -> feature fuse -> crop box -> BGR2GRAY + resize, 640x480 camera frames resident in HBM
        from silent_speech_amd import features as Fm

        Nf, hh, ww = 512, 480, 640
        frames = torch.randint(0, 256, (Nf, hh, ww, 3), device=dev, dtype=torch.uint8)
        lmf = lm.reshape(-1, K, 2)[:Nf].reshape(1, Nf, K, 2).contiguous()
        Xf = torch.empty(1, Nf, D, device=dev)
        center = torch.empty(1, Nf, 2, device=dev)
        fourth = torch.empty(1, Nf, device=dev, dtype=torch.float64)
        L.call("ss_feature_fuse", lmf.data_ptr(), None, 1, Nf, K, ww, hh, 8, 25, 1, 2, 0, Xf.data_ptr(), D, center.data_ptr(),
               fourth.data_ptr(), L.stream())
        boxes = Fm.crop_boxes(center, fourth, ww, hh, "record")
        out = {}
        for variant in ("record", "live"):
            for _ in range(args.warmup):
                Fm.crop_rois(frames, boxes, (48, 96), variant)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                Fm.crop_rois(frames, boxes, (48, 96), variant)
            e1.record()
            torch.cuda.synchronize()
            out[variant] = e0.elapsed_time(e1) / args.steps
        bx = boxes.reshape(-1, 5).cpu()
        crop_bytes = int(((bx[:, 1] - bx[:, 0]) * (bx[:, 3] - bx[:, 2]) * bx[:, 4]).sum()) * 3 + Nf * 48 * 96
        ms = out["live"]  # INTER_AREA reads every byte of the crop; INTER_LINEAR only four taps per output pixel
        print(json.dumps({"metric": "frames/sec cropped, greyed and resized to 48x96 (640x480 BGR frames in HBM, INTER_AREA)",
                          "value": round(Nf / ms * 1e3, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms, 4), "higher_is_better": True, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": "SURVEY 8f-2: crop box -> BGR2GRAY -> resize INTER_AREA (live); INTER_LINEAR "
                                                 "(recorder) %.4f ms per %d frames" % (out["record"], Nf)},
                          "roofline": {"kernel": "ss_crop_gray_resize", "bound": "hbm", "achieved": round(crop_bytes / ms / 1e6, 1),
                                       "peak": 8000.0, "unit": "GB/s", "frac": round(crop_bytes / ms / 1e6 / 8000.0, 4), "traffic": None,
                                       "algorithmic_bytes_per_launch": crop_bytes, "avg_launch_ms": round(ms, 4)}}))
        return
    if args.mode == "stream":
        S, Ts = (4096, 60) if (args.batch, args.frames) == (256, 30) else (args.batch, args.frames)
        print(json.dumps(stream_block(ss, dev, D, C, roi, S, Ts, args.steps)))
        return
    if args.mode == "live":
        # the live chain as one device entry point (live_infer_official.py:264-296 per stream + the sliding-window rule): landmarks and
        # 640x480 BGR camera frames resident in HBM -> width gate -> feature fuse with per-stream velocity state -> crop box -> gray +
        # INTER_AREA resize -> ring push -> one forward over the streams that are due (every 2nd tick)
        from silent_speech_amd import features as Fm

        S, Ts = (1024, 60) if (args.batch, args.frames) == (256, 30) else (args.batch, args.frames)
        Kl = len(Fm.FIXED_IDXS_88)
        Dl = 2 * Kl + 4
        hh, ww, roi_hw = 480, 640, (48, 96)
        model = ss.BiGRUClassifier(Dl, C, use_roi=True).to(dev).eval()
        srv = ss.StreamServer(model, S, Ts, roi_hw=roi_hw, device=dev)
        srv.attach_front_end(Fm.FIXED_IDXS_88, (ww, hh), variant="live")
        g = torch.Generator(device=dev).manual_seed(7)
        lmk = torch.rand(S, Kl, 2, device=dev, generator=g) * 0.2 + 0.4
        al, ar = Fm.anchor_positions(Fm.FIXED_IDXS_88)[:2]
        lmk[:, al] = torch.tensor([0.42, 0.61], device=dev)   # mouth corners 0.16 x 640 = 102 px apart: inside the 60-150 px band
        lmk[:, ar] = torch.tensor([0.58, 0.61], device=dev)
        lmk[::16, ar, 0] = 0.45                                # every 16th stream too far away: dropped by the gate
        frames = torch.randint(0, 256, (S, hh, ww, 3), device=dev, dtype=torch.uint8, generator=g)
        ids = list(range(S))
        for _ in range(Ts + (Ts % 2)):
            srv.push_landmarks(ids, lmk, frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_pred = n_kept = 0
        for _ in range(2 * args.steps):
            kept, got = srv.push_landmarks(ids, lmk, frames)
            n_kept += int(kept.sum())
            n_pred += 0 if got is None else len(got[0])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(2 * args.steps):
            srv.front(ids, lmk, frames)       # gate + features + crop + resize alone (state advances, rings untouched)
        torch.cuda.synchronize()
        front_ms = 1000 * (time.perf_counter() - t0) / (2 * args.steps)
        print(json.dumps({"metric": "camera frames/sec through the live chain (%d streams: gate + features + crop + gray/resize + ring + "
                                    "forward every 2nd tick on %d-frame windows)" % (S, Ts),
                          "value": round(2 * args.steps * S / el, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": 0,
                          "ms_per_step": round(1000 * el / args.steps, 3), "higher_is_better": True, "dtype": "f32 / u8", "data": "synthetic",
                          "windows_per_sec": round(n_pred / el, 1), "frames_kept_per_sec": round(n_kept / el, 1),
                          "front_end_ms_per_tick": round(front_ms, 3),
                          "config": {"workload": "live_infer_official.py:264-296 for %d streams per tick, 640x480 BGR frames and 88 landmarks "
                                                 "resident in HBM, ROI 48x96, D=%d, T=%d" % (S, Dl, Ts)}}))
        return
    if args.mode == "assemble":
        # a training batch gathered out of a clip store that lives in HBM (noise, frame drop, pad / trim)
        import tempfile

        from silent_speech_amd import data as Dm

        rs = __import__("numpy").random.default_rng(0)
        with tempfile.TemporaryDirectory() as tmp:
            files = []
            for k in range(64):  # 64 distinct clips, visited many times: the store is what matters, not the file count
                Tk = int(rs.integers(T, T + 8))
                p_ = os.path.join(tmp, f"c{k}.npz")
                Dm.save_clip(p_, rs.normal(size=(Tk, D)).astype("float32"), range(Tk), "w%d" % (k % C), "me", range(4),
                             rs.integers(0, 256, (Tk, roi, roi), dtype="uint8"))
                files.append(p_)
            store = ss.DeviceClipStore(files, {"w%d" % c: c for c in range(C)}, max_t=T, device=dev)
        order = [int(v) for v in rs.integers(0, len(store), B)]
        gen = __import__("numpy").random.default_rng(1)
        for _ in range(args.warmup):
            store.batch(order, augment=True, rng="device", generator=gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            Xb, Tb, Rb, yb = store.batch(order, augment=True, rng="device", generator=gen)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # the two gather launches alone, HIP events on their stream
        L.PROFILE = {}
        for _ in range(args.steps):
            store.batch(order, augment=True, rng="device", generator=gen)
        torch.cuda.synchronize()
        prof, L.PROFILE = L.PROFILE, None
        k_ms = {tag: sum(a.elapsed_time(b) for a, b in evs) / len(evs) for tag, evs in prof.items()}
        frame_bytes = roi * roi
        alg = 2 * B * T * frame_bytes + B * T * 4  # ROI gather: read + write + its map
        u8_ms = k_ms.get("ss_batch_gather_u8", float("nan"))
        print(json.dumps({"metric": "batches/sec assembled on device (B=%d, T=%d, %dx%d ROI, augment on)" % (B, T, roi, roi),
                          "value": round(args.steps / el, 1), "unit": "batches/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1000 * el / args.steps, 3), "higher_is_better": True,
                          "dtype": "u8/f32", "data": "synthetic",
                          "config": {"workload": "SURVEY 8f-1: NPZWordDataset + collate_fn rules on a clip store resident in HBM"},
                          "roofline": {"kernel": "ss_batch_gather_u8", "bound": "hbm", "achieved": round(alg / u8_ms / 1e6, 1),
                                       "peak": 8000.0, "unit": "GB/s", "frac": round(alg / u8_ms / 1e6 / 8000.0, 4), "traffic": None,
                                       "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(u8_ms, 4)},
                          "kernels_ms": {k: round(v, 4) for k, v in k_ms.items()}}))
        return
    spec = spec_for(args.config, B, T, K, roi, C)
    lengths = ragged(lengths, T, args.min_len_frac, spec)
    out = train_block(ss, L, dev, world, rank, spec, X, lengths, R, y, args.steps, args.warmup, not args.no_kernel_times,
                      args.micro_batches, dist_on, extra_windows=args.extra_windows)
    solo = rank == 0 and world == 1
    # ---- the other BASELINE configs ride along in the default run, so that the driver's one command measures them too
    if args.config == 2 and not args.no_config5:
        # BASELINE config 5 is a data-parallel training config like the headline: every rank runs it (one all-reduce per step)
        torch.cuda.empty_cache()
        s5 = spec_for(5, 256, 30, K, C5["roi"], C5["classes"])
        X5, l5, R5, y5 = synth_inputs(L, dev, rank, 256, 30, K, s5["roi_hw"], s5["C"])
        out["config5"] = train_block(ss, L, dev, world, rank, s5, X5, l5, R5, y5, args.steps, max(2, args.warmup // 2),
                                     not args.no_kernel_times, 1, dist_on)
        del X5, R5
    if args.config == 2 and not args.no_shipped and not args.shipped_only:
        out["shipped"] = shipped_blocks(ss, L, dev, world, rank, args, dist_on)
    if solo and args.config == 2 and not args.no_config4:
        torch.cuda.empty_cache()
        out["config4"] = config4_block(ss, dev, D, C, roi, 20, 3)
    if solo and args.config == 2 and args.min_len_frac >= 1.0 and not args.no_padded:
        out["padded_batches"] = padded_block(ss, L, dev, B, T, K, roi, C, X, R, y)
    if solo and args.config == 2 and not args.no_config4:
        # beside config 4 (4 096 INDEPENDENT windows through the whole model): the same count of camera streams served as what they
        # are, windows that slide by two frames -- not config 4's number, which recomputes every frame of every window
        torch.cuda.empty_cache()
        out["sliding_window_serving"] = stream_block(ss, dev, D, C, roi, 4096, 60, 10)
    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, D, C, args.config)
        if "config5" in out:
            out["config5"]["cpu_baseline"] = cpu_baseline_c5(args, D, 30, seconds=6.0)
        if "shipped" in out:
            out["shipped"]["cpu_baseline"] = cpu_baseline_shipped(seconds=5.0)
    if dist_on:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
