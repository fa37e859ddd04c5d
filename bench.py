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
import json
import os
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


PMC_TRAFFIC = os.path.join("profiles", "round2_c_pmc_traffic.json")
PMC_MFMA = os.path.join("profiles", "round2_c_pmc_mfma.json")


def pmc_mfma_busy(kernel_tag):
    """MFMA-pipe busy fraction of a kernel from the committed rocprofv3 PMC summary (SQ_VALU_MFMA_BUSY_CYCLES over
    SQ_BUSY_CU_CYCLES-equivalent time; tools/pmc_summary.py mfma)."""
    path = os.path.join(ROOT, PMC_MFMA)
    stem = {"ss_roi_cnn_bwd": "roi_cnn_bwd_kernel", "ss_roi_cnn_fwd_stash": "roi_cnn_fwd_kernel"}.get(kernel_tag)
    if not stem or not os.path.exists(path):
        return None
    for name, d in json.load(open(path))["kernels"].items():
        if stem in name:
            return d
    return None


def pmc_traffic(kernel_tag):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary (made by tools/pmc_summary.py from
    separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    path = os.path.join(ROOT, PMC_TRAFFIC)
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "round1_g_pmc_traffic.json")
    stem = {"ss_roi_cnn_bwd": "roi_cnn_bwd_kernel", "ss_roi_cnn_fwd_stash": "roi_cnn_fwd_kernel", "ss_gru_fwd": "gru_split_fwd_kernel",
            "ss_gru_bwd": "gru_split_bwd_kernel"}.get(kernel_tag)
    if not stem or not os.path.exists(path):
        return None
    for name, d in json.load(open(path))["kernels"].items():
        if stem in name:
            return int(d["hbm_bytes_per_launch"])
    return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU (weak scaling)")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--landmarks", type=int, default=40)
    ap.add_argument("--roi", type=int, default=64)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-times", action="store_true")
    ap.add_argument("--mode", choices=["train", "infer", "assemble", "stream", "crop"], default="train",
                    help="train = the headline metric; infer = BASELINE config 4 (T=60, B=4096 windows, forward-only, hipGraph)")
    ap.add_argument("--micro-batches", type=int, default=1, help="slices of the per-GPU batch kept in flight on separate streams")
    ap.add_argument("--config", type=int, choices=[2, 5], default=2,
                    help="2 = the headline (f32, 64x64 ROI, H=192, 5 words); 5 = BASELINE config 5 (bf16 MFMA, 96x96 ROI, CNN "
                         "16/32/64/96, H=512, 100 words)")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 (hipGraph serving) block of the default run")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    import silent_speech_amd as ss
    from silent_speech_amd import _lib as L

    if args.mode == "infer" and args.batch == 256 and args.frames == 30:
        args.batch, args.frames = 4096, 60
    B, T, K, C, roi = args.batch, args.frames, args.landmarks, args.classes, args.roi
    D, E, H = 2 * K + 4, 32, 192
    if args.config == 5:
        E, H = C5["emb"], C5["hidden"]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    # synthetic clips: landmarks -> K1 feature fuse on device; uint8 ROI; full lengths
    base = torch.rand(B, 1, K, 2, device=dev, generator=g) * torch.tensor([0.4, 0.4], device=dev) + torch.tensor([0.3, 0.4], device=dev)
    lm = base + 0.004 * torch.randn(B, T, K, 2, device=dev, generator=g)
    lm[:, :, 8] = torch.tensor([0.42, 0.61], device=dev)   # landmark 61 sits at position 8 of the sorted 40-lip list
    lm[:, :, 25] = torch.tensor([0.58, 0.61], device=dev)  # 291
    lm[:, :, 1] = torch.tensor([0.50, 0.59], device=dev)   # 13
    lm[:, :, 2] = torch.tensor([0.50, 0.63], device=dev)   # 14
    lm = (lm + 0.002 * torch.randn(B, T, K, 2, device=dev, generator=g)).contiguous()
    X = torch.empty(B, T, D, device=dev)
    L.call("ss_feature_fuse", lm.data_ptr(), None, B, T, K, 640, 480, 8, 25, 1, 2, 0, X.data_ptr(), D, None, None, L.stream())
    R = torch.randint(0, 256, (B, T, roi, roi), device=dev, dtype=torch.uint8, generator=g)
    lengths = torch.full((B,), T, device=dev, dtype=torch.int64)
    y = torch.randint(0, C, (B,), device=dev, generator=g)

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
        # S camera streams, one new frame per stream and tick, a prediction per stream every 2nd tick on its
        # zero-padded sliding window (inactive/live_feed.py:155-213) -- rings, window assembly and forward all on the device
        S, Ts = (4096, 60) if (args.batch, args.frames) == (256, 30) else (args.batch, args.frames)
        model = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).eval()
        srv = ss.StreamServer(model, S, Ts, roi_hw=(roi, roi), device=dev)
        ids = list(range(S))
        feats = torch.randn(S, D, device=dev)
        rois = torch.randint(0, 256, (S, roi, roi), device=dev, dtype=torch.uint8)
        op = torch.rand(S, device=dev) * 0.05
        for _ in range(Ts + (Ts % 2)):  # fill the rings; ends on an even frame count
            srv.push(ids, feats, rois, op)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_pred = 0
        for _ in range(2 * args.steps):
            got = srv.push(ids, feats, rois, op)
            n_pred += 0 if got is None else len(got[0])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps({"metric": "windows/sec (sliding %d-frame windows of %d streams, push + assemble + forward)" % (Ts, S),
                          "value": round(n_pred / el, 1), "unit": "windows/s", "n_gpus": 1, "steps": args.steps, "warmup": 0,
                          "ms_per_step": round(1000 * el / args.steps, 3), "higher_is_better": True, "dtype": "f32",
                          "data": "synthetic", "frames_ingested_per_sec": round(2 * args.steps * S / el, 1),
                          "config": {"workload": "SURVEY 8f-4: %d streams x T=%d, one frame per stream and tick, prediction every "
                                                 "2nd tick, landmark + %dx%d ROI CNN + BiGRU forward" % (S, Ts, roi, roi)}}))
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
    if args.config == 5:
        model = ss.BiGRUClassifier(D, C, use_roi=True, roi_emb=E, hidden=H, cnn_channels=C5["channels"], precision="bf16").to(dev).train()
    else:
        model = ss.BiGRUClassifier(D, C, use_roi=True).to(dev).train()
    if world > 1:
        dist.broadcast(model.flat_params, src=0)
    trainer = ss.Trainer(model, world_size=world, micro_batches=args.micro_batches)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gb = B * world
    for _ in range(args.warmup):
        trainer.step(X, lengths, R, y, global_batch=gb)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.step(X, lengths, R, y, global_batch=gb)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    final_loss = float(loss)
    assert final_loss == final_loss, "loss is NaN"

    # ---- per-kernel launch durations over K more steps, HIP events on the launch stream
    kernels, roof = {}, None
    if not args.no_kernel_times:
        from silent_speech_amd import engine, engine_bf16

        engine.USE_SIDE_STREAM = engine_bf16.USE_SIDE_STREAM = False  # one stream, so each event pair brackets exactly one kernel
        mb, trainer.micro_batches = trainer.micro_batches, 1
        L.PROFILE = {}
        for _ in range(args.steps):
            trainer.step(X, lengths, R, y)
        torch.cuda.synchronize()
        prof, L.PROFILE = L.PROFILE, None
        engine.USE_SIDE_STREAM = engine_bf16.USE_SIDE_STREAM = True
        trainer.micro_batches = mb
        for tag, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            kernels[tag] = {"launches_per_step": len(ms) / args.steps, "avg_ms": sum(ms) / len(ms),
                            "ms_per_step": sum(ms) / args.steps}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        if args.config == 5:
            gfs = c5_gflop_per_step(B, T, D)
            gf = gfs.get(dom)
            if gf is not None:
                achieved = gf / kernels[dom]["ms_per_step"]  # GFLOP per step / ms per step = TFLOP/s
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                        "algorithmic_gflop_per_step": round(gf, 2), "launches_per_step": kernels[dom]["launches_per_step"],
                        "ms_per_step": round(kernels[dom]["ms_per_step"], 4)}
            kernel_tf = {k: round(gfs[k] / v["ms_per_step"], 1) for k, v in kernels.items() if k in gfs}
        else:
            gf = algorithmic_gflop(dom, B, T, D, E, H, C, (roi, roi))
            if gf is not None:
                achieved = gf / kernels[dom]["avg_ms"]  # GFLOP / ms = TFLOP/s
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": pmc_traffic(dom) if (B, T, roi) == (256, 30, 64) else None,
                        "algorithmic_gflop_per_launch": round(gf, 3), "avg_launch_ms": round(kernels[dom]["avg_ms"], 4)}
                mb_ = pmc_mfma_busy(dom) if (B, T, roi) == (256, 30, 64) else None
                if mb_:
                    roof["mfma_busy"] = mb_

    if args.config == 5:
        workload = (f"BASELINE config 5: landmark (K={K}, D={D}) + 96x96 uint8 ROI CNN (16,32,64,96) + 2-layer BiGRU({H}), T={T}, "
                    f"C={C} words, bf16 MFMA operands / f32 accumulation and master weights, train step = fwd + CE(ls .05) + bwd + "
                    "grad all-reduce + clip(1.0) + Adam, dropout on")
        peak, gpc = BF16_MFMA_PEAK_TFLOPS, c5_step_gflop_per_clip(T, D)
    else:
        workload = (f"BASELINE config 2: landmark (K={K}, D={D}) + {roi}x{roi} uint8 ROI CNN + 2-layer BiGRU(192), "
                    f"T={T}, C={C}, train step = fwd + CE(ls .05) + bwd + grad all-reduce + clip(1.0) + Adam, dropout on")
        peak, gpc = F32_MFMA_PEAK_TFLOPS, step_gflop_per_clip(T, D, E, H, C, (roi, roi))
    out = {
        "metric": "clips/sec (30-frame, fwd+bwd)", "value": round(B * world * args.steps / elapsed, 1), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.config == 5 else "f32",
        "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "global_batch": B * world, "frames": T, "parallelism": f"dp{world}",
                   "micro_batches_in_flight": args.micro_batches},
        "final_loss": round(final_loss, 5),
    }
    if roof:
        out["roofline"] = roof
    # whole-step figure of SURVEY.md 8(d): clips/s x algorithmic GFLOP per clip against the dense MFMA peak (of the dtype) of the GPUs used
    tf = out["value"] * gpc / 1e3
    out["step_roofline"] = {"bound": "mfma", "gflop_per_clip": round(gpc, 4), "achieved": round(tf, 2),
                            "peak": round(peak * world, 1), "unit": "TFLOP/s", "frac": round(tf / (peak * world), 4)}
    if kernels:
        out["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        if args.config == 5:
            out["kernels_tflops"] = kernel_tf
    if rank == 0 and world == 1 and args.config == 2 and not args.no_config4:
        del trainer, model
        torch.cuda.empty_cache()
        out["config4"] = config4_block(ss, dev, D, C, roi, 20, 3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, D, C, args.config)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
