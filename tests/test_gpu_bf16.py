"""GPU: the bf16-MFMA kernels of BASELINE config 5 (gemm_bf16.hip, gru_bf16.hip, cnn_bf16*.hip) through the C ABI.

Checker: torch on the CPU with the operands rounded to bf16 exactly where the kernels round them (weights and the
activations that enter an MFMA) and f32 accumulation -- what is left between kernel and checker is summation order, so the
tolerances are f32-tight.  The distance of that bf16 arithmetic from the f32 oracle (oracle/model_ref.py) is measured
separately at model level (tests/test_gpu_model_c5.py) against the tolerance DESIGN.md states."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from silent_speech_amd import _lib

    _lib.load()
    return _lib


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


INT_MAX = 2**31 - 1


def to_bf16_rows(L, x, drop_p=0.0, seed=0, offset=0):
    """f32 (rows, cols) device tensor -> int16 (rows, ceil8(cols)) of bf16 bit patterns, zero-padded (ss_cvt_bf16_rows)."""
    rows, cols = x.shape
    ld = (cols + 7) // 8 * 8
    y = torch.full((rows, ld), 0x7fc0, device="cuda", dtype=torch.int16)
    L.call("ss_cvt_bf16_rows", x.data_ptr(), cols, y.data_ptr(), ld, rows, cols, drop_p, seed, offset, L.stream())
    return y


@pytest.mark.parametrize("src16", [0, 1])
@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(200, 148, 148), (128, 128, 32), (260, 64, 1000), (16, 1536, 512)])
def test_gemm_bf16_layouts(L, akc, bkc, M, N, K, src16):
    """``src16``: the operands already are bf16 in HBM (flags bit 3), leading dimensions zero-padded to a multiple of 8."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + akc * 2 + bkc)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = (bf(A).double() @ bf(B).double()).float() + bias
    A_st = (A if akc else A.t()).contiguous().cuda()     # [M][K] or [K][M]
    B_st = (B.t() if bkc else B).contiguous().cuda()     # [N][K] or [K][N]
    if src16:
        A_st, B_st = to_bf16_rows(L, A_st), to_bf16_rows(L, B_st)
        assert torch.equal(A_st.cpu().view(torch.bfloat16).float()[:, : (K if akc else M)], bf(A if akc else A.t()))
        assert int(A_st[:, (K if akc else M):].abs().sum()) == 0
    C = torch.full((M, N), 7.0, device="cuda")
    bias_d = bias.cuda()
    L.call("ss_gemm_bf16_batched", akc, bkc, M, N, K, A_st.data_ptr(), A_st.shape[1], INT_MAX, 0, 0, B_st.data_ptr(),
           B_st.shape[1], INT_MAX, 0, 0, C.data_ptr(), N, bias_d.data_ptr(), 8 * src16, 1, 1, 0, 0, 0, 0, L.stream())
    torch.cuda.synchronize()
    err = float((C.cpu() - ref).abs().max())
    assert err < 2e-4 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("src16", [0, 1])
def test_gemm_bf16_batch_splitk_remap(L, src16):
    """Two problems per launch (both GRU directions), K sliced over workgroups with float atomics into an initialised C,
    and the storage-row remap that pairs dG[b][t] with h[b][t-1] (group T-1 of stride T)."""
    g = torch.Generator().manual_seed(1)
    Bc, T, Mm, Nn = 12, 9, 96, 72
    rows = Bc * T
    dG = torch.randn(2, rows, Mm, generator=g)      # A stored [k = (b,t)][m]
    Hs = torch.randn(2, rows, Nn, generator=g)      # B stored [k = (b,t)][n]
    C0 = torch.randn(2, Mm, Nn, generator=g)
    K = Bc * (T - 1)
    ref = C0.clone()
    for d in range(2):
        a = dG[d].view(Bc, T, Mm)[:, 1:].reshape(K, Mm)      # rows (b, t >= 1)
        h = Hs[d].view(Bc, T, Nn)[:, :-1].reshape(K, Nn)     # rows (b, t - 1)
        ref[d] += (bf(a).double().t() @ bf(h).double()).float()
    C = C0.clone().cuda()
    dGd, Hd = dG.cuda(), Hs.cuda()
    if src16:
        dGd, Hd = to_bf16_rows(L, dGd.view(2 * rows, Mm)), to_bf16_rows(L, Hd.view(2 * rows, Nn))
    L.call("ss_gemm_bf16_batched", 0, 0, Mm, Nn, K, dGd.data_ptr(), Mm, T - 1, T, 1, Hd.data_ptr(), Nn, T - 1, T, 0,
           C.data_ptr(), Nn, None, 1 + 8 * src16, 3, 2, rows * Mm, rows * Nn, Mm * Nn, 0, L.stream())
    torch.cuda.synchronize()
    assert float((C.cpu() - ref).abs().max()) < 3e-4 * float(ref.abs().max())


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K,mode", [(256, 128, 64, "store"), (300, 200, 192, "store"), (512, 136, 128, "accumulate"),
                                        (132, 64, 256, "store"), (776, 384, 640, "splitk"), (1024, 256, 1024, "store"),
                                        (384, 72, 320, "splitk")])
def test_gemm_bf16_ring(L, akc, bkc, M, N, K, mode):
    """The LDS-DMA ring kernel (bf16 operands, K a multiple of 64, M >= 128): every operand layout, ragged M and N (zeros
    arrive for rows / columns outside the operand, stores are guarded), plain / accumulating stores through the LDS staging
    image, K slices with float atomics; two problems per launch.  Run twice: the three-stage ring must leave nothing behind."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + akc * 2 + bkc)
    A = torch.randn(2, M, K, generator=g)
    B = torch.randn(2, K, N, generator=g)
    bias = torch.randn(2, N, generator=g)
    C0 = torch.randn(2, M, N, generator=g)
    ref = torch.stack([(bf(A[d]).double() @ bf(B[d]).double()).float() for d in range(2)])
    A_st = (A if akc else A.transpose(1, 2)).contiguous().cuda()
    B_st = (B.transpose(1, 2) if bkc else B).contiguous().cuda()
    ra, rb = A_st.shape[1], B_st.shape[1]
    A16 = to_bf16_rows(L, A_st.view(2 * ra, -1))
    B16 = to_bf16_rows(L, B_st.view(2 * rb, -1))
    lda, ldb = A16.shape[1], B16.shape[1]
    for rep in range(2):
        C = C0.clone().cuda()
        if mode == "store":
            flags, splits, want, bias_p = 8, 1, ref + bias[:, None, :], bias.cuda()
        elif mode == "accumulate":
            flags, splits, want, bias_p = 8 | 1, 1, ref + C0, None
        else:
            flags, splits, want, bias_p = 8 | 1, 3, ref + C0, None
        L.call("ss_gemm_bf16_batched", akc, bkc, M, N, K, A16.data_ptr(), lda, INT_MAX, 0, 0, B16.data_ptr(), ldb, INT_MAX, 0, 0,
               C.data_ptr(), N, L.ptr(bias_p), flags, splits, 2, ra * lda, rb * ldb, M * N, N, L.stream())
        torch.cuda.synchronize()
        err = float((C.cpu() - want).abs().max())
        assert err < 3e-4 * max(1.0, float(want.abs().max())), (rep, err)


def test_gemm_bf16_ring_remap_and_kcat(L):
    """Ring kernel: the storage-row remap of k-major operands that pairs dG[b][t] with h[b][t-1] (K = B (T - 1) = 128 rows picked
    out of B T), and the K-concatenated form (flags bit 4: the two directions' products summed into ONE C by plain stores)."""
    g = torch.Generator().manual_seed(5)
    Bc, T, Mm, Nn = 16, 9, 192, 136
    rows = Bc * T
    dG = torch.randn(2, rows, Mm, generator=g)
    Hs = torch.randn(2, rows, Nn, generator=g)
    C0 = torch.randn(2, Mm, Nn, generator=g)
    K = Bc * (T - 1)
    ref = C0.clone()
    for d in range(2):
        a = dG[d].view(Bc, T, Mm)[:, 1:].reshape(K, Mm)
        h = Hs[d].view(Bc, T, Nn)[:, :-1].reshape(K, Nn)
        ref[d] += (bf(a).double().t() @ bf(h).double()).float()
    C = C0.clone().cuda()
    dGd, Hd = to_bf16_rows(L, dG.cuda().view(2 * rows, Mm)), to_bf16_rows(L, Hs.cuda().view(2 * rows, Nn))
    L.call("ss_gemm_bf16_batched", 0, 0, Mm, Nn, K, dGd.data_ptr(), Mm, T - 1, T, 1, Hd.data_ptr(), Nn, T - 1, T, 0,
           C.data_ptr(), Nn, None, 1 + 8, 2, 2, rows * Mm, rows * Nn, Mm * Nn, 0, L.stream())
    torch.cuda.synchronize()
    assert float((C.cpu() - ref).abs().max()) < 3e-4 * float(ref.abs().max())
    # K concatenation: d layer_in = dG_f W_f + dG_r W_r
    M, N, K = 400, 256, 192
    A = torch.randn(2, M, K, generator=g)      # [row][k]
    W = torch.randn(2, K, N, generator=g)      # [k][n]
    want = sum((bf(A[d]).double() @ bf(W[d]).double()).float() for d in range(2))
    A16, W16 = to_bf16_rows(L, A.cuda().view(2 * M, K)), to_bf16_rows(L, W.cuda().view(2 * K, N))
    C = torch.full((M, N), 3.0, device="cuda")
    L.call("ss_gemm_bf16_batched", 1, 0, M, N, K, A16.data_ptr(), K, INT_MAX, 0, 0, W16.data_ptr(), N, INT_MAX, 0, 0,
           C.data_ptr(), N, None, 8 | 16, 1, 2, M * K, K * N, 0, 0, L.stream())
    torch.cuda.synchronize()
    assert float((C.cpu() - want).abs().max()) < 3e-4 * float(want.abs().max())


@pytest.mark.parametrize("Bc,T,H,Kin", [(64, 5, 128, 136), (128, 3, 256, 512), (64, 5, 512, 1024)])
def test_gemm_bf16_splitk_group(L, Bc, T, H, Kin):
    """The weight gradients of one GRU layer as ONE grouped launch (ss_gemm_bf16_splitk_group): d W_ih and the two pieces of
    d W_hh, both directions each, K = B T resp. B (T - 1) with the row remap; the K tiles of the whole group are dealt evenly
    over the CUs, so workgroups cross tile and problem boundaries.  C is accumulated into; run twice.  The last case has 144
    output tiles (at least half a chip): one workgroup per tile over all of K, C += straight from the registers."""
    g = torch.Generator().manual_seed(Bc + T)
    N = Bc * T
    dG = torch.randn(2, N, 4 * H, generator=g)
    X = torch.randn(N, Kin, generator=g)
    out = torch.randn(N, 2 * H, generator=g)
    G_ih0 = torch.randn(2, 3 * H, Kin, generator=g)
    G_hh0 = torch.randn(2, 3 * H, H, generator=g)
    want_ih, want_hh = G_ih0.clone(), G_hh0.clone()
    for d in range(2):
        want_ih[d] += (bf(dG[d][:, :3 * H]).double().t() @ bf(X).double()).float()
        dg3 = dG[d].view(Bc, T, 4 * H)
        o3 = out.view(Bc, T, 2, H)[:, :, d]
        a = (dg3[:, 1:] if d == 0 else dg3[:, :-1]).reshape(-1, 4 * H)
        h = (o3[:, :-1] if d == 0 else o3[:, 1:]).reshape(-1, H)
        want_hh[d][:2 * H] += (bf(a[:, :2 * H]).double().t() @ bf(h).double()).float()
        want_hh[d][2 * H:] += (bf(a[:, 3 * H:]).double().t() @ bf(h).double()).float()
    dG16 = to_bf16_rows(L, dG.cuda().view(2 * N, 4 * H))
    X16 = to_bf16_rows(L, X.cuda())
    out16 = to_bf16_rows(L, out.cuda())
    Kp = X16.shape[1]
    Kh = Bc * (T - 1)
    for rep in range(2):
        G_ih, G_hh = G_ih0.clone().cuda(), G_hh0.clone().cuda()
        dg, hp = dG16.data_ptr(), out16.data_ptr()
        sa, sb = N * 4 * H - 4 * H, 3 * H
        pr = [L.GemmProblem(0, 0, 3 * H, Kin, N, dg, 4 * H, INT_MAX, 0, 0, X16.data_ptr(), Kp, INT_MAX, 0, 0, G_ih.data_ptr(), Kin, 1, 2,
                            N * 4 * H, 0, 3 * H * Kin),
              L.GemmProblem(0, 0, 2 * H, H, Kh, dg, 4 * H, T - 1, T, 1, hp, 2 * H, T - 1, T, 0, G_hh.data_ptr(), H, 1, 2, sa, sb, 3 * H * H),
              L.GemmProblem(0, 0, H, H, Kh, dg + 3 * H * 2, 4 * H, T - 1, T, 1, hp, 2 * H, T - 1, T, 0, G_hh.data_ptr() + 2 * H * H * 4, H,
                            1, 2, sa, sb, 3 * H * H)]
        ws = torch.full((L.gemm_group_ws_floats(pr, bf16=True),), float("nan"), device="cuda")
        arr, n = L.gemm_group(pr)
        L.call("ss_gemm_bf16_splitk_group", arr, n, ws.data_ptr(), ws.numel(), L.stream())
        torch.cuda.synchronize()
        for got, want in ((G_ih, want_ih), (G_hh, want_hh)):
            assert float((got.cpu() - want).abs().max()) < 3e-4 * float(want.abs().max()), rep


def _gru_emulated(gi, whh, bhh, lengths, T, H, reverse):
    """One direction with the kernel's roundings: bf16 W_hh and bf16 previous state inside the matmul, f32 elsewhere."""
    B = gi.shape[0]
    h = gi.new_zeros(B, H)
    outs = [None] * T
    saves = [None] * T
    wb = bf(whh)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        valid = (lengths > t).float().unsqueeze(1)
        gh = bf(h) @ wb.t()
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H] + bhh[:H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H] + bhh[H:2 * H])
        hpre = gh[:, 2 * H:] + bhh[2 * H:]
        n = torch.tanh(gi[:, t, 2 * H:] + r * hpre)
        hn = (1 - z) * n + z * h
        h = valid * hn  # the kernel's state past a clip's end is zero (nothing valid follows in either direction)
        outs[t] = h
        saves[t] = torch.stack([r, z, n, hpre], 1) * valid.unsqueeze(2)
    return torch.stack(outs, 1), torch.stack(saves, 1)


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("B,T,H,drop_p", [(5, 7, 128, 0.0), (70, 4, 256, 0.0), (33, 6, 512, 0.0), (33, 6, 512, 0.1), (5, 7, 128, 0.3),
                                          (256, 5, 512, 0.1), (300, 3, 512, 0.0), (20, 9, 384, 0.0)])
def test_gru_bf16_fwd_bwd(L, B, T, H, drop_p, persistent):
    """Both forms behind the entry points: ``persistent`` = one launch per layer (H/64 workgroups per (16-clip slice, direction)
    exchanging the state through tagged granules; B = 256 fills the chip and takes the same-XCD path, B = 300 runs as two
    clip chunks, the small batches take the write-through path), otherwise one launch per time step.  Checked together with
    the by-products: the bf16 copies of out / dropout(out) / d_g and the bias-gradient column sums.
    ``drop_p`` > 0: the BPTT kernel re-draws nn.GRU's inter-layer dropout mask while it reads d_out (Philox stream of
    ss_dropout at the same seed / offset / element index): checked against the mask ss_dropout itself writes."""
    g = torch.Generator().manual_seed(B + T + H)
    N = B * T
    seed, offset = 77, 3 << 40
    lengths = torch.randint(1, T + 1, (B,), generator=g)
    lengths[0] = T
    if B > 2:
        lengths[1] = 1
    gi = torch.randn(2, B, T, 3 * H, generator=g) * 0.7
    whh = [torch.randn(3 * H, H, generator=g) / H ** 0.5 for _ in range(2)]
    bhh = [torch.randn(3 * H, generator=g) * 0.1 for _ in range(2)]
    gi_l = [gi[d].clone().requires_grad_(True) for d in range(2)]
    whh_l = [w.clone().requires_grad_(True) for w in whh]
    outs, saves = zip(*[_gru_emulated(gi_l[d], whh_l[d], bhh[d], lengths, T, H, reverse=bool(d)) for d in range(2)])
    out_ref = torch.cat(outs, 2)                                  # (B,T,2H)
    d_out = torch.randn(B, T, 2 * H, generator=g)
    dev = lambda x: x.contiguous().cuda()
    keep = torch.ones(B, T, 2 * H)
    if drop_p > 0:
        ones = torch.ones(N, 2 * H, device="cuda")
        km = torch.empty_like(ones)
        L.call("ss_dropout", ones.data_ptr(), km.data_ptr(), N * 2 * H, drop_p, seed, offset, None, L.stream())
        keep = km.cpu().view(B, T, 2 * H)
        frac = float((keep == 0).float().mean())
        assert abs(frac - drop_p) < 0.03 and float(keep.max()) == pytest.approx(1.0 / (1.0 - drop_p))
    (out_ref * d_out * keep).sum().backward()

    wb = torch.empty(2, 3 * H, H, device="cuda", dtype=torch.int16)
    wtb = torch.empty(2, H, 3 * H, device="cuda", dtype=torch.int16)
    w_f, w_r = dev(whh[0]), dev(whh[1])
    Kin = 148 if H == 512 else 2 * H  # W_ih rides along: (3H, K) -> (3H, K rounded up to 8) bf16, zero-padded
    Kp = (Kin + 7) // 8 * 8
    wih = [torch.randn(3 * H, Kin, generator=g) for _ in range(2)]
    wi_f, wi_r = dev(wih[0]), dev(wih[1])
    wib = torch.full((2, 3 * H, Kp), 0x7fc0, device="cuda", dtype=torch.int16)
    L.call("ss_gru_bf16_prep", w_f.data_ptr(), w_r.data_ptr(), H, wb.data_ptr(), wtb.data_ptr(), wi_f.data_ptr(), wi_r.data_ptr(), Kin,
           wib.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(wb.cpu().view(torch.bfloat16).float(), bf(torch.stack(whh)))
    assert torch.equal(wtb.cpu().view(torch.bfloat16).float(), bf(torch.stack(whh)).transpose(1, 2))
    wic = wib.cpu().view(torch.bfloat16).float()
    assert torch.equal(wic[:, :, :Kin], bf(torch.stack(wih))) and float(wic[:, :, Kin:].abs().sum()) == 0
    import ctypes

    nb = ctypes.c_long(0)
    assert L.load().ss_gru_bf16_ws_bytes(B, H, ctypes.byref(nb)) == 0
    ws = torch.empty(nb.value, device="cuda", dtype=torch.uint8)
    sync = None
    if persistent:
        assert L.load().ss_gru_bf16_sync_bytes(B, T, H, ctypes.byref(nb)) == 0 and nb.value > 0
        sync = torch.zeros(nb.value // 4, device="cuda", dtype=torch.int32)
    gid = dev(gi.reshape(2, N, 3 * H))
    b_f, b_r = dev(bhh[0]), dev(bhh[1])
    lens = lengths.to(torch.int32).cuda()
    out = torch.full((N, 2 * H), 9.0, device="cuda")
    save = torch.full((2, N, 4, H), 9.0, device="cuda")
    out_bf = torch.full((N, 2 * H), 0x7fc0, device="cuda", dtype=torch.int16)
    out_drop_bf = torch.full((N, 2 * H), 0x7fc0, device="cuda", dtype=torch.int16)
    L.call("ss_gru_bf16_fwd", gid.data_ptr(), wb.data_ptr(), b_f.data_ptr(), b_r.data_ptr(), lens.data_ptr(), B, T, H,
           out.data_ptr(), save.data_ptr(), out_bf.data_ptr(), out_drop_bf.data_ptr(), drop_p, seed, offset, ws.data_ptr(),
           L.ptr(sync), L.nbytes(sync), L.stream())
    torch.cuda.synchronize()
    if sync is not None:
        assert int(sync[2]) == 0, "a bounded wait of the persistent recurrence gave up"
        if B == 256:  # placement is never assumed for correctness; on this pool round-robin dispatch puts all partners on one XCD
            print(f"same-XCD fast path taken by {int(sync[3])} of 256 workgroups")
    assert torch.equal(out_bf.cpu().view(torch.bfloat16).float(), bf(out.cpu()))
    assert torch.equal(out_drop_bf.cpu().view(torch.bfloat16).float(), bf(out.cpu() * keep.reshape(N, 2 * H)))
    err = float((out.cpu().view(B, T, 2 * H) - out_ref.detach()).abs().max())
    assert err < 2e-3, err  # bf16 rounding of the state decides differently only through v_exp/v_rcp noise: amplified by 1 bf16 ulp
    mask = (torch.arange(T)[None] < lengths[:, None]).float()[None, :, :, None, None]
    sv = save.cpu().view(2, B, T, 4, H) * mask
    assert float((sv - torch.stack([s.detach() for s in saves])).abs().max()) < 5e-3

    dG = torch.full((2, N, 4, H), 9.0, device="cuda")
    d_out_d = dev(d_out.reshape(N, 2 * H))
    dG_bf = torch.full((2, N, 4, H), 0x7fc0, device="cuda", dtype=torch.int16)
    gb = [torch.zeros(3 * H, device="cuda") for _ in range(4)]  # d b_ih fwd, d b_hh fwd, d b_ih rev, d b_hh rev
    L.call("ss_gru_bf16_bwd", d_out_d.data_ptr(), out.data_ptr(), save.data_ptr(), wtb.data_ptr(),
           lens.data_ptr(), B, T, H, dG.data_ptr(), dG_bf.data_ptr(), drop_p, seed, offset, *[t_.data_ptr() for t_ in gb],
           ws.data_ptr(), L.ptr(sync), L.nbytes(sync), L.stream())
    torch.cuda.synchronize()
    if sync is not None:
        assert int(sync[2]) == 0
    assert torch.equal(dG_bf.cpu().view(torch.bfloat16).float(), bf(dG.cpu()))
    if sync is not None:  # the f32 gate gradients are optional there: only the bf16 copy
        dG_bf2 = torch.full_like(dG_bf, 0x7fc0)
        L.call("ss_gru_bf16_bwd", d_out_d.data_ptr(), out.data_ptr(), save.data_ptr(), wtb.data_ptr(), lens.data_ptr(), B, T, H,
               None, dG_bf2.data_ptr(), drop_p, seed, offset, None, None, None, None, ws.data_ptr(), L.ptr(sync), L.nbytes(sync), L.stream())
        torch.cuda.synchronize()
        assert torch.equal(dG_bf2, dG_bf)
    dGs = dG.cpu().view(2, N, 4, H).sum(1)  # (2, 4, H) column sums
    for d in range(2):
        want_ih, want_hh = dGs[d, :3].reshape(-1), torch.cat([dGs[d, 0], dGs[d, 1], dGs[d, 3]])
        for got, want in ((gb[2 * d], want_ih), (gb[2 * d + 1], want_hh)):
            assert float((got.cpu() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    dGc = dG.cpu().view(2, B, T, 4, H)
    for d in range(2):
        ref = gi_l[d].grad.view(B, T, 3, H)   # d gi = (d r_pre, d z_pre, d n_pre)
        got = dGc[d][:, :, :3]
        scale = float(ref.abs().max())
        # the kernel rounds the gate gradients to bf16 before W_hh^T takes them one step back: 2^-9 relative per step
        assert float((got - ref).abs().max()) < 2e-2 * scale, (d, float((got - ref).abs().max()), scale)
        # d W_hh from d_g: sum_t d gh_t^T h_{t-1}, d gh = (d r_pre, d z_pre, d hn)
        dgh = torch.cat([dGc[d][:, :, 0], dGc[d][:, :, 1], dGc[d][:, :, 3]], 2)         # (B,T,3H)
        hseq = outs[d].detach()
        hprev = torch.zeros_like(hseq)
        if d == 0:
            hprev[:, 1:] = hseq[:, :-1]
        else:
            hprev[:, :-1] = hseq[:, 1:]
        dW = torch.einsum("btg,bth->gh", dgh, hprev)
        refW = whh_l[d].grad
        assert float((dW - refW).abs().max()) < 3e-2 * float(refW.abs().max())


# ------------------------------------------------------------------------------------------------ ROI CNN (cnn_bf16*.hip)
import torch.nn.functional as F  # noqa: E402

C5 = [1, 16, 32, 64, 96]


def nhwc_bf16(x):
    """(N,C,H,W) f32 holding bf16 values -> device int16 tensor (N,H,W,C) of the bf16 bit patterns."""
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).view(torch.int16).cuda()


def from_nhwc(t, dtype=torch.bfloat16):
    """device (N,H,W,C) bit patterns -> CPU (N,C,H,W) f32."""
    return t.cpu().view(dtype).float().permute(0, 3, 1, 2).contiguous()


def pool_ref(c, bias):
    """raw conv output (N,C,H,W) -> (pooled ReLU(max + b) f32, argmax byte with the kernels' encoding, top-2 gap)."""
    n, ch, h, w = c.shape
    win = c.reshape(n, ch, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, ch, h // 2, w // 2, 4)
    best, bi = win.max(dim=-1)
    val = torch.relu(best + bias.view(1, -1, 1, 1))
    idx = torch.where(val > 0, bi, torch.full_like(bi, 4)).to(torch.uint8)
    top = win.topk(2, dim=-1).values
    return val, idx, top[..., 0] - top[..., 1]


def assert_bf16_close(name, got, ref, frac_exact=0.995):
    """``got`` holds bf16 values a kernel rounded from f32 sums; ``ref`` the same sums from another summation order."""
    refb = bf(ref)
    exact = (got == refb).float().mean().item()
    err = (got - ref).abs()
    tol = 2.0 ** -7 * ref.abs() + 1e-6 * max(1.0, float(ref.abs().max()))
    assert bool((err <= tol).all()), f"{name}: max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})"
    assert exact >= frac_exact, f"{name}: only {exact:.4f} of the elements round identically"


def expand_ref(da, idx):
    """pooled-grid gradient (N,C,h,w) + argmax bytes -> dense (N,C,2h,2w)."""
    n, c, h, w = da.shape
    out = torch.zeros(n, c, h, 2, w, 2)
    for e in range(4):
        out[:, :, :, e >> 1, :, e & 1] = torch.where(idx == e, da, torch.zeros_like(da))
    return out.reshape(n, c, 2 * h, 2 * w)


def normalise_like_kernel(R, standardize=True):
    """The forward kernel's per-frame normalisation, operation by operation (exact integer sums, f64 mean / variance,
    f32 table of the 256 grey levels, train_model_official.py:286-291): -> (xn f32 (N,96,96), mu (N,), sd (N,))."""
    Rn = R.numpy().astype(np.int64)
    N = Rn.shape[0]
    xn = np.empty(Rn.shape, np.float32)
    mus, sds = np.zeros(N, np.float32), np.ones(N, np.float32)
    lev = np.arange(256, dtype=np.float32) / np.float32(255.0)
    for n in range(N):
        tab = lev
        if standardize:
            tsu, tsq, nn = float(Rn[n].sum()), float((Rn[n] ** 2).sum()), float(Rn[n].size)
            mu = np.float32(np.float32(tsu / nn) / np.float32(255.0))
            var = (tsq - tsu * tsu / nn) / (nn - 1.0)
            sd = max(np.float32(np.sqrt(max(var, 0.0)) / 255.0), np.float32(1e-6))
            tab = ((lev - mu) / sd).astype(np.float32)
            mus[n], sds[n] = mu, sd
        xn[n] = tab[Rn[n]]
    return torch.from_numpy(xn), torch.from_numpy(mus), torch.from_numpy(sds)


@pytest.fixture(params=[0, 7], ids=["grid=CUs", "grid=7"])
def wg_cap(L, request):
    """The CNN kernels are persistent: a workgroup walks its share of the frames and fetches the next frame / band while it
    computes the current one.  With the default grid (one or two workgroups per CU) a test of 300 frames walks two frames on
    a few workgroups; capped at 7 workgroups every one of them walks 40+ (the frame pipeline is what the bench runs)."""
    L.call("ss_roi_cnn_set_max_workgroups", request.param)
    yield request.param
    L.call("ss_roi_cnn_set_max_workgroups", 0)


@pytest.mark.parametrize("N,standardize", [(3, 1), (300, 1), (4, 0)])
def test_c5_conv1_fwd(L, N, standardize, wg_cap):
    g = torch.Generator().manual_seed(N)
    R = torch.randint(0, 256, (N, 96, 96), generator=g, dtype=torch.uint8)
    R[0] = 7                     # constant frame: std clamp, every normalised pixel is exactly 0
    R[1] = 200
    R[1, 5, 9] = 201             # near-constant frame
    w1 = torch.randn(16, 1, 3, 3, generator=g) * 0.4
    b1 = torch.randn(16, generator=g) * 0.2
    xn, mu, sd = normalise_like_kernel(R, bool(standardize))
    c1 = F.conv2d(bf(xn).unsqueeze(1).double(), bf(w1).double(), padding=1).float()
    val, idx, gap = pool_ref(c1, b1)
    a1 = torch.empty(N, 48, 48, 16, device="cuda", dtype=torch.int16)
    i1 = torch.empty(N, 48, 48, 16, device="cuda", dtype=torch.uint8)
    st = torch.empty(N, 2, device="cuda")
    R_d, w_d, b_d = R.cuda(), w1.cuda(), b1.cuda()  # named: a temporary's memory is recycled before the kernel has run
    L.call("ss_c5_conv1_fwd", R_d.data_ptr(), N, standardize, w_d.data_ptr(), b_d.data_ptr(), a1.data_ptr(), i1.data_ptr(),
           st.data_ptr(), L.stream())
    torch.cuda.synchronize()
    if standardize:
        assert torch.equal(st[:, 0].cpu(), mu) and torch.equal(st[:, 1].cpu(), sd)
        # the statistics are torch's (train_model_official.py:288-290) to f32 rounding
        r = R.float() / 255.0
        assert float((mu - r.mean(dim=(1, 2))).abs().max()) < 1e-6
        assert float(((sd - r.std(dim=(1, 2)).clamp_min(1e-6)) / sd).abs().max()) < 1e-5
    got = from_nhwc(a1)
    if standardize:  # constant frame: conv output 0 everywhere -> ReLU(bias)
        assert torch.equal(got[0], bf(torch.relu(b1)).view(16, 1, 1).expand(16, 48, 48))
    assert_bf16_close("a1", got, val)
    sure = (gap > 1e-3) & ((val > 1e-3) | (val == 0))
    assert torch.equal(i1.cpu().permute(0, 3, 1, 2)[sure], idx[sure])


@pytest.mark.parametrize("layer,N", [(2, 2), (2, 270), (3, 3), (3, 300)])
def test_c5_conv_fwd(L, layer, N, wg_cap):
    cin, cout = C5[layer - 1], C5[layer]
    hw = 96 >> (layer - 1)
    g = torch.Generator().manual_seed(layer * 100 + N)
    a_in = bf(torch.relu(torch.randn(N, cin, hw, hw, generator=g)))
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    c = F.conv2d(a_in.double(), bf(w).double(), padding=1).float()
    val, idx, gap = pool_ref(c, b)
    out = torch.empty(N, hw // 2, hw // 2, cout, device="cuda", dtype=torch.int16)
    io = torch.empty(N, hw // 2, hw // 2, cout, device="cuda", dtype=torch.uint8)
    a_d, w_d, b_d = nhwc_bf16(a_in), w.cuda(), b.cuda()
    L.call("ss_c5_conv_fwd", layer, a_d.data_ptr(), N, w_d.data_ptr(), b_d.data_ptr(), out.data_ptr(), io.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert_bf16_close(f"a{layer}", from_nhwc(out), val)
    sure = (gap > 1e-3) & ((val > 1e-3) | (val == 0))
    assert torch.equal(io.cpu().permute(0, 3, 1, 2)[sure], idx[sure])


@pytest.mark.parametrize("N", [3, 300])
def test_c5_conv_last_fwd(L, N, wg_cap):
    g = torch.Generator().manual_seed(N)
    E, ld = 64, 148
    a3 = bf(torch.relu(torch.randn(N, 64, 12, 12, generator=g)))
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(96, generator=g) * 0.1
    wfc = torch.randn(E, 96, generator=g) / 10.0
    bfc = torch.randn(E, generator=g) * 0.1
    x = F.conv2d(a3.double(), bf(w).double(), padding=1).float() + b.view(1, -1, 1, 1)
    feat = torch.relu(x).mean(dim=(2, 3))
    z_ref = feat @ wfc.t() + bfc
    z = torch.full((N, ld), 5.0, device="cuda")
    mask = torch.empty(N, 144, 96, device="cuda", dtype=torch.uint8)
    fo = torch.empty(N, 96, device="cuda")
    a_d, w_d, b_d, wfc_d, bfc_d = nhwc_bf16(a3), w.cuda(), b.cuda(), wfc.cuda(), bfc.cuda()
    L.call("ss_c5_conv_last_fwd", a_d.data_ptr(), N, w_d.data_ptr(), b_d.data_ptr(), wfc_d.data_ptr(), bfc_d.data_ptr(), E,
           z.data_ptr() + 84 * 4, ld, mask.data_ptr(), fo.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((fo.cpu() - feat).abs().max()) < 2e-5
    assert float((z[:, 84:].cpu() - z_ref).abs().max()) < 5e-5 and float((z[:, :84] - 5.0).abs().max()) == 0.0
    m_ref = (x > 0).reshape(N, 96, 144).permute(0, 2, 1)
    sure = (x.abs() > 1e-3).reshape(N, 96, 144).permute(0, 2, 1)
    assert torch.equal(mask.cpu().bool()[sure], m_ref[sure])
    # inference form: no stash
    z2 = torch.zeros(N, E, device="cuda")
    L.call("ss_c5_conv_last_fwd", a_d.data_ptr(), N, w_d.data_ptr(), b_d.data_ptr(), wfc_d.data_ptr(), bfc_d.data_ptr(), E,
           z2.data_ptr(), E, None, None, L.stream())
    torch.cuda.synchronize()
    assert float((z2.cpu() - z_ref).abs().max()) < 5e-5
    # the form the engine runs: features only (the Linear is a GEMM over all frames), with and without the mask
    for m_ptr in (mask, None):
        fo2 = torch.zeros(N, 96, device="cuda")
        m2 = torch.full_like(mask, 7)
        L.call("ss_c5_conv_last_fwd_feat", a_d.data_ptr(), N, w_d.data_ptr(), b_d.data_ptr(), m2.data_ptr() if m_ptr is not None else None,
               fo2.data_ptr(), L.stream())
        torch.cuda.synchronize()
        assert float((fo2.cpu() - feat).abs().max()) < 2e-5
        assert m_ptr is None or torch.equal(m2.cpu().bool()[sure], m_ref[sure])


@pytest.mark.parametrize("layer,N", [(2, 2), (2, 270), (3, 3), (3, 300)])
def test_c5_conv_bwd(L, layer, N, wg_cap):
    cin, cout = C5[layer - 1], C5[layer]
    hw = 96 >> (layer - 1)
    g = torch.Generator().manual_seed(layer * 10 + N)
    a_in = bf(torch.relu(torch.randn(N, cin, hw, hw, generator=g)))
    da = bf(torch.randn(N, cout, hw // 2, hw // 2, generator=g))
    idx = torch.randint(0, 5, (N, cout, hw // 2, hw // 2), generator=g).to(torch.uint8)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)
    dy = expand_ref(da, idx)
    gw_ref = torch.nn.grad.conv2d_weight(a_in.double(), w.shape, dy.double(), padding=1).float()
    gb_ref = dy.double().sum(dim=(0, 2, 3)).float()
    din_ref = F.conv_transpose2d(dy.double(), bf(w).double(), padding=1).float()
    a_d, da_d = nhwc_bf16(a_in), nhwc_bf16(da)
    idx_d = idx.permute(0, 2, 3, 1).contiguous().cuda()
    gw = torch.ones(cout, cin, 3, 3, device="cuda")   # gradients accumulate: start from 1
    gb = torch.ones(cout, device="cuda")
    L.call("ss_c5_conv_wgrad", layer, a_d.data_ptr(), da_d.data_ptr(), idx_d.data_ptr(), N, gw.data_ptr(), gb.data_ptr(), L.stream())
    din = torch.empty(N, hw, hw, cin, device="cuda", dtype=torch.int16)
    w_d = w.cuda()
    L.call("ss_c5_conv_dgrad", layer, da_d.data_ptr(), idx_d.data_ptr(), N, w_d.data_ptr(), din.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((gw.cpu() - 1.0 - gw_ref).abs().max()) < 2e-4 * float(gw_ref.abs().max())
    assert float((gb.cpu() - 1.0 - gb_ref).abs().max()) < 2e-4 * float(gb_ref.abs().max())
    assert_bf16_close(f"da{layer - 1}", from_nhwc(din), din_ref)
    # the same through the scratch buffer of partial sums (plain stores + a reduce launch instead of float atomics onto g_w)
    part = torch.full((min(N, 256) * cout * cin * 9,), float("nan"), device="cuda")
    gw2, gb2 = torch.ones_like(gw), torch.ones_like(gb)
    L.call("ss_c5_conv_wgrad_ws", layer, a_d.data_ptr(), da_d.data_ptr(), idx_d.data_ptr(), N, gw2.data_ptr(), gb2.data_ptr(), part.data_ptr(),
           part.numel(), L.stream())
    torch.cuda.synchronize()
    assert float((gw2.cpu() - 1.0 - gw_ref).abs().max()) < 2e-4 * float(gw_ref.abs().max())
    assert float((gb2.cpu() - 1.0 - gb_ref).abs().max()) < 2e-4 * float(gb_ref.abs().max())


@pytest.mark.parametrize("N", [3, 300])
def test_c5_conv_last_bwd(L, N, wg_cap):
    g = torch.Generator().manual_seed(N + 5)
    E, ld = 64, 148
    a3 = bf(torch.relu(torch.randn(N, 64, 12, 12, generator=g)))
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    wfc = torch.randn(E, 96, generator=g) / 10.0
    dz = torch.zeros(N, ld)
    dz[:, 84:] = torch.randn(N, E, generator=g)
    mask = torch.rand(N, 96, 12, 12, generator=g) < 0.5
    feat = torch.rand(N, 96, generator=g)
    dfeat = (dz[:, 84:] @ wfc) * (1.0 / 144.0)
    dy = bf(dfeat).view(N, 96, 1, 1) * mask.float()
    gw_ref = torch.nn.grad.conv2d_weight(a3.double(), w.shape, dy.double(), padding=1).float()
    gb_ref = dy.double().sum(dim=(0, 2, 3)).float()
    gwfc_ref = dz[:, 84:].t() @ feat
    gbfc_ref = dz[:, 84:].sum(0)
    din_ref = F.conv_transpose2d(dy.double(), bf(w).double(), padding=1).float()
    m_d = mask.reshape(N, 96, 144).permute(0, 2, 1).contiguous().to(torch.uint8).cuda()
    dz_d, wfc_d, feat_d, a_d = dz.cuda(), wfc.cuda(), feat.cuda(), nhwc_bf16(a3)
    gw = torch.zeros(96, 64, 3, 3, device="cuda")
    gb = torch.zeros(96, device="cuda")
    gwfc = torch.zeros(E, 96, device="cuda")
    gbfc = torch.zeros(E, device="cuda")
    L.call("ss_c5_conv_last_wgrad", a_d.data_ptr(), dz_d.data_ptr() + 84 * 4, ld, E, wfc_d.data_ptr(), m_d.data_ptr(), feat_d.data_ptr(),
           N, gw.data_ptr(), gb.data_ptr(), gwfc.data_ptr(), gbfc.data_ptr(), L.stream())
    din = torch.empty(N, 12, 12, 64, device="cuda", dtype=torch.int16)
    w_d = w.cuda()
    L.call("ss_c5_conv_last_dgrad", dz_d.data_ptr() + 84 * 4, ld, E, wfc_d.data_ptr(), m_d.data_ptr(), N, w_d.data_ptr(),
           din.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((gw.cpu() - gw_ref).abs().max()) < 3e-4 * float(gw_ref.abs().max())
    assert float((gb.cpu() - gb_ref).abs().max()) < 3e-4 * float(gb_ref.abs().max())
    assert float((gwfc.cpu() - gwfc_ref).abs().max()) < 1e-4 * float(gwfc_ref.abs().max())
    assert float((gbfc.cpu() - gbfc_ref).abs().max()) < 1e-4 * float(gbfc_ref.abs().max())
    assert_bf16_close("da3", from_nhwc(din), din_ref, frac_exact=0.98)
    # the form the engine runs: d z . W_fc ready-made for every frame (a GEMM), the fc gradients elsewhere
    df_d = (dz[:, 84:] @ wfc).contiguous().cuda()
    gw2, gb2 = torch.zeros_like(gw), torch.zeros_like(gb)
    din2 = torch.empty_like(din)
    part = torch.full((min(N, 256) * 96 * 64 * 9 + 4,), float("nan"), device="cuda")
    L.call("ss_c5_conv_last_wgrad_df", a_d.data_ptr(), df_d.data_ptr(), m_d.data_ptr(), N, gw2.data_ptr(), gb2.data_ptr(), part.data_ptr(), part.numel(),
           L.stream())
    L.call("ss_c5_conv_last_dgrad_df", df_d.data_ptr(), m_d.data_ptr(), N, w_d.data_ptr(), din2.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((gw2.cpu() - gw_ref).abs().max()) < 3e-4 * float(gw_ref.abs().max())
    assert float((gb2.cpu() - gb_ref).abs().max()) < 3e-4 * float(gb_ref.abs().max())
    assert_bf16_close("da3", from_nhwc(din2), din_ref, frac_exact=0.98)


@pytest.mark.parametrize("N", [2, 300])
def test_c5_conv1_wgrad(L, N, wg_cap):
    g = torch.Generator().manual_seed(N + 11)
    R = torch.randint(0, 256, (N, 96, 96), generator=g, dtype=torch.uint8)
    xn, mu, sd = normalise_like_kernel(R)
    xn = bf(xn)
    da = bf(torch.randn(N, 16, 48, 48, generator=g))
    idx = torch.randint(0, 5, (N, 16, 48, 48), generator=g).to(torch.uint8)
    dy = expand_ref(da, idx)
    gw_ref = torch.nn.grad.conv2d_weight(xn.unsqueeze(1).double(), (16, 1, 3, 3), dy.double(), padding=1).float()
    gb_ref = dy.double().sum(dim=(0, 2, 3)).float()
    st = torch.stack([mu, sd], 1).contiguous().cuda()
    gw = torch.zeros(16, 1, 3, 3, device="cuda")
    gb = torch.zeros(16, device="cuda")
    R_d, da_d, idx_d = R.cuda(), nhwc_bf16(da), idx.permute(0, 2, 3, 1).contiguous().cuda()
    L.call("ss_c5_conv1_wgrad", R_d.data_ptr(), N, 1, st.data_ptr(), da_d.data_ptr(), idx_d.data_ptr(), None, None, gw.data_ptr(),
           gb.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((gw.cpu() - gw_ref).abs().max()) < 2e-4 * float(gw_ref.abs().max())
    assert float((gb.cpu() - gb_ref).abs().max()) < 2e-4 * float(gb_ref.abs().max())


@pytest.mark.parametrize("N", [3, 300])
def test_c5_conv12_fused_forward_and_recomputing_backward(L, N, wg_cap):
    """The fused forms the engine runs: conv1 + conv2 in one kernel (the pooled conv1 map stays in LDS), layer 2's weight gradient
    with that map recomputed from the frame, conv1's weight gradient with its pool winners recomputed -- against the same
    references as the per-layer kernels."""
    g = torch.Generator().manual_seed(N + 3)
    R = torch.randint(0, 256, (N, 96, 96), generator=g, dtype=torch.uint8)
    R[0] = 30
    w1 = torch.randn(16, 1, 3, 3, generator=g) * 0.4
    b1 = torch.randn(16, generator=g) * 0.2
    w2 = torch.randn(32, 16, 3, 3, generator=g) / 12.0
    b2 = torch.randn(32, generator=g) * 0.1
    xn, mu, sd = normalise_like_kernel(R)
    c1 = F.conv2d(bf(xn).unsqueeze(1).double(), bf(w1).double(), padding=1).float()
    a1, i1, gap1 = pool_ref(c1, b1)
    a1 = bf(a1)
    c2 = F.conv2d(a1.double(), bf(w2).double(), padding=1).float()
    a2, i2, gap2 = pool_ref(c2, b2)
    R_d, w1_d, b1_d, w2_d, b2_d = R.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda()
    a2_d = torch.empty(N, 24, 24, 32, device="cuda", dtype=torch.int16)
    i2_d = torch.empty(N, 24, 24, 32, device="cuda", dtype=torch.uint8)
    st = torch.empty(N, 2, device="cuda")
    L.call("ss_c5_conv12_fwd", R_d.data_ptr(), N, 1, w1_d.data_ptr(), b1_d.data_ptr(), w2_d.data_ptr(), b2_d.data_ptr(), a2_d.data_ptr(),
           i2_d.data_ptr(), st.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(st[:, 0].cpu(), mu) and torch.equal(st[:, 1].cpu(), sd)
    # a rounding flip of a conv1 output (1 bf16 ulp) moves the conv2 sums by ~1e-3 of their size: looser than the per-layer test
    got = from_nhwc(a2_d)
    err = (got - a2).abs()
    assert bool((err <= 2.0 ** -6 * a2.abs() + 2e-3 * float(a2.abs().max())).all()), float(err.max())
    sure = (gap2 > 2e-2) & ((a2 > 2e-2) | (a2 == 0))
    assert (i2_d.cpu().permute(0, 3, 1, 2)[sure] != i2[sure]).float().mean() < 1e-3

    # ---- layer 2's weight gradient from recomputed a1
    da2 = bf(torch.randn(N, 32, 24, 24, generator=g))
    idx2 = torch.randint(0, 5, (N, 32, 24, 24), generator=g).to(torch.uint8)
    dy2 = expand_ref(da2, idx2)
    gw_ref = torch.nn.grad.conv2d_weight(a1.double(), w2.shape, dy2.double(), padding=1).float()
    gb_ref = dy2.double().sum(dim=(0, 2, 3)).float()
    da2_d, idx2_d = nhwc_bf16(da2), idx2.permute(0, 2, 3, 1).contiguous().cuda()
    gw = torch.zeros(32, 16, 3, 3, device="cuda")
    gb = torch.zeros(32, device="cuda")
    L.call("ss_c5_conv2_wgrad_rc", R_d.data_ptr(), st.data_ptr(), 1, w1_d.data_ptr(), b1_d.data_ptr(), da2_d.data_ptr(), idx2_d.data_ptr(),
           N, gw.data_ptr(), gb.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((gw.cpu() - gw_ref).abs().max()) < 1e-3 * float(gw_ref.abs().max())
    assert float((gb.cpu() - gb_ref).abs().max()) < 2e-4 * float(gb_ref.abs().max())
    part = torch.full((min(N, 256) * 32 * 16 * 9,), float("nan"), device="cuda")  # partial sums through scratch instead of atomics
    gw_s, gb_s = torch.zeros_like(gw), torch.zeros_like(gb)
    L.call("ss_c5_conv2_wgrad_rc_ws", R_d.data_ptr(), st.data_ptr(), 1, w1_d.data_ptr(), b1_d.data_ptr(), da2_d.data_ptr(), idx2_d.data_ptr(),
           N, gw_s.data_ptr(), gb_s.data_ptr(), part.data_ptr(), part.numel(), L.stream())
    torch.cuda.synchronize()
    assert float((gw_s.cpu() - gw_ref).abs().max()) < 1e-3 * float(gw_ref.abs().max())
    assert float((gb_s.cpu() - gb_ref).abs().max()) < 2e-4 * float(gb_ref.abs().max())

    # ---- conv1's weight gradient with recomputed pool winners
    da1 = bf(torch.randn(N, 16, 48, 48, generator=g))
    dy1 = expand_ref(da1, i1)
    # where conv1's winner is nearly tied the recomputation may pick the other slot: compare where it is clear
    win1 = c1.reshape(N, 16, 48, 2, 48, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, 16, 48, 48, 4)
    pre1 = win1.max(dim=-1).values + b1.view(1, -1, 1, 1)      # pooled value before the ReLU
    clear = (pre1 < -1e-3) | ((pre1 > 1e-3) & (gap1 > 1e-3))    # dead in both, or alive with an unambiguous winner
    dy1 = dy1 * clear.repeat_interleave(2, 2).repeat_interleave(2, 3)
    da1c = da1 * clear
    g1_ref = torch.nn.grad.conv2d_weight(bf(xn).unsqueeze(1).double(), (16, 1, 3, 3), dy1.double(), padding=1).float()
    da1_d = nhwc_bf16(da1c)
    g1 = torch.zeros(16, 1, 3, 3, device="cuda")
    gb1 = torch.zeros(16, device="cuda")
    L.call("ss_c5_conv1_wgrad", R_d.data_ptr(), N, 1, st.data_ptr(), da1_d.data_ptr(), None, w1_d.data_ptr(), b1_d.data_ptr(), g1.data_ptr(),
           gb1.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert float((g1.cpu() - g1_ref).abs().max()) < 3e-4 * float(g1_ref.abs().max())
    assert float((gb1.cpu() - dy1.double().sum(dim=(0, 2, 3)).float()).abs().max()) < 3e-4 * float(gb1.abs().max())

    # ---- conv2's data gradient + conv1's weight gradient fused (d a1 stays in LDS): the same numbers as the two kernels in a row
    da1_two = torch.empty(N, 48, 48, 16, device="cuda", dtype=torch.int16)
    L.call("ss_c5_conv_dgrad", 2, da2_d.data_ptr(), idx2_d.data_ptr(), N, w2_d.data_ptr(), da1_two.data_ptr(), L.stream())
    g1a, gb1a = torch.zeros(16, 1, 3, 3, device="cuda"), torch.zeros(16, device="cuda")
    L.call("ss_c5_conv1_wgrad", R_d.data_ptr(), N, 1, st.data_ptr(), da1_two.data_ptr(), None, w1_d.data_ptr(), b1_d.data_ptr(),
           g1a.data_ptr(), gb1a.data_ptr(), L.stream())
    da1_f = torch.full_like(da1_two, 0x7fc0)
    g1f, gb1f = torch.zeros(16, 1, 3, 3, device="cuda"), torch.zeros(16, device="cuda")
    L.call("ss_c5_conv2_dgrad_conv1_wgrad", da2_d.data_ptr(), idx2_d.data_ptr(), N, w2_d.data_ptr(), R_d.data_ptr(), st.data_ptr(), 1,
           w1_d.data_ptr(), b1_d.data_ptr(), da1_f.data_ptr(), g1f.data_ptr(), gb1f.data_ptr(), L.stream())
    g1n, gb1n = torch.zeros(16, 1, 3, 3, device="cuda"), torch.zeros(16, device="cuda")
    L.call("ss_c5_conv2_dgrad_conv1_wgrad", da2_d.data_ptr(), idx2_d.data_ptr(), N, w2_d.data_ptr(), R_d.data_ptr(), st.data_ptr(), 1,
           w1_d.data_ptr(), b1_d.data_ptr(), None, g1n.data_ptr(), gb1n.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(da1_f, da1_two), "the fused kernel's d a1 band differs from conv_dgrad's"
    for got_w, got_b in ((g1f, gb1f), (g1n, gb1n)):
        assert float((got_w - g1a).abs().max()) < 1e-4 * float(g1a.abs().max())
        assert float((got_b - gb1a).abs().max()) < 1e-4 * float(gb1a.abs().max())

    # ---- the pool winners stashed by the forward kernel (ss_c5_conv12_fwd_i1) instead of recomputed by the fused backward kernel:
    # the same bytes as the unfused conv1 kernel writes, the same a2 / i2, the same conv1 gradient
    a1_u = torch.empty(N, 48, 48, 16, device="cuda", dtype=torch.int16)
    i1_u = torch.empty(N, 48, 48, 16, device="cuda", dtype=torch.uint8)
    L.call("ss_c5_conv1_fwd", R_d.data_ptr(), N, 1, w1_d.data_ptr(), b1_d.data_ptr(), a1_u.data_ptr(), i1_u.data_ptr(), None, L.stream())
    a2_s, i2_s = torch.empty_like(a2_d), torch.empty_like(i2_d)
    i1_s = torch.full((N, 48, 48, 16), 9, device="cuda", dtype=torch.uint8)
    L.call("ss_c5_conv12_fwd_i1", R_d.data_ptr(), N, 1, w1_d.data_ptr(), b1_d.data_ptr(), w2_d.data_ptr(), b2_d.data_ptr(), a2_s.data_ptr(),
           i2_s.data_ptr(), st.data_ptr(), i1_s.data_ptr(), L.stream())
    g1s, gb1s = torch.zeros(16, 1, 3, 3, device="cuda"), torch.zeros(16, device="cuda")
    L.call("ss_c5_conv2_dgrad_conv1_wgrad_i1", da2_d.data_ptr(), idx2_d.data_ptr(), N, w2_d.data_ptr(), R_d.data_ptr(), st.data_ptr(), 1,
           w1_d.data_ptr(), b1_d.data_ptr(), None, g1s.data_ptr(), gb1s.data_ptr(), i1_s.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(a2_s, a2_d) and torch.equal(i2_s, i2_d)
    assert float((i1_s != i1_u).float().mean()) < 1e-5, "stashed conv1 pool winners differ from the unfused kernel's"
    assert float((g1s - g1a).abs().max()) < 1e-4 * float(g1a.abs().max())
    assert float((gb1s - gb1a).abs().max()) < 1e-4 * float(gb1a.abs().max())
