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


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(200, 148, 148), (128, 128, 32), (260, 64, 1000), (16, 1536, 512)])
def test_gemm_bf16_layouts(L, akc, bkc, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + akc * 2 + bkc)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = (bf(A).double() @ bf(B).double()).float() + bias
    A_st = (A if akc else A.t()).contiguous().cuda()     # [M][K] or [K][M]
    B_st = (B.t() if bkc else B).contiguous().cuda()     # [N][K] or [K][N]
    C = torch.full((M, N), 7.0, device="cuda")
    L.call("ss_gemm_bf16_batched", akc, bkc, M, N, K, A_st.data_ptr(), A_st.shape[1], INT_MAX, 0, 0, B_st.data_ptr(),
           B_st.shape[1], INT_MAX, 0, 0, C.data_ptr(), N, bias.cuda().data_ptr(), 0, 1, 1, 0, 0, 0, 0, L.stream())
    torch.cuda.synchronize()
    err = float((C.cpu() - ref).abs().max())
    assert err < 2e-4 * max(1.0, float(ref.abs().max())), err


def test_gemm_bf16_batch_splitk_remap(L):
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
    L.call("ss_gemm_bf16_batched", 0, 0, Mm, Nn, K, dGd.data_ptr(), Mm, T - 1, T, 1, Hd.data_ptr(), Nn, T - 1, T, 0,
           C.data_ptr(), Nn, None, 1, 3, 2, rows * Mm, rows * Nn, Mm * Nn, 0, L.stream())
    torch.cuda.synchronize()
    assert float((C.cpu() - ref).abs().max()) < 3e-4 * float(ref.abs().max())


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


@pytest.mark.parametrize("B,T,H", [(5, 7, 128), (70, 4, 256), (33, 6, 512)])
def test_gru_bf16_fwd_bwd(L, B, T, H):
    g = torch.Generator().manual_seed(B + T + H)
    N = B * T
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
    (out_ref * d_out).sum().backward()

    dev = lambda x: x.contiguous().cuda()
    wb = torch.empty(2, 3 * H, H, device="cuda", dtype=torch.int16)
    wtb = torch.empty(2, H, 3 * H, device="cuda", dtype=torch.int16)
    w_f, w_r = dev(whh[0]), dev(whh[1])
    L.call("ss_gru_bf16_prep", w_f.data_ptr(), w_r.data_ptr(), H, wb.data_ptr(), wtb.data_ptr(), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(wb.cpu().view(torch.bfloat16).float(), bf(torch.stack(whh)))
    assert torch.equal(wtb.cpu().view(torch.bfloat16).float(), bf(torch.stack(whh)).transpose(1, 2))
    import ctypes

    nb = ctypes.c_long(0)
    assert L.load().ss_gru_bf16_ws_bytes(B, H, ctypes.byref(nb)) == 0
    ws = torch.empty(nb.value, device="cuda", dtype=torch.uint8)
    gid = dev(gi.reshape(2, N, 3 * H))
    b_f, b_r = dev(bhh[0]), dev(bhh[1])
    lens = lengths.to(torch.int32).cuda()
    out = torch.full((N, 2 * H), 9.0, device="cuda")
    save = torch.full((2, N, 4, H), 9.0, device="cuda")
    L.call("ss_gru_bf16_fwd", gid.data_ptr(), wb.data_ptr(), b_f.data_ptr(), b_r.data_ptr(), lens.data_ptr(), B, T, H,
           out.data_ptr(), save.data_ptr(), ws.data_ptr(), L.stream())
    torch.cuda.synchronize()
    err = float((out.cpu().view(B, T, 2 * H) - out_ref.detach()).abs().max())
    assert err < 2e-3, err  # bf16 rounding of the state decides differently only through v_exp/v_rcp noise: amplified by 1 bf16 ulp
    mask = (torch.arange(T)[None] < lengths[:, None]).float()[None, :, :, None, None]
    sv = save.cpu().view(2, B, T, 4, H) * mask
    assert float((sv - torch.stack([s.detach() for s in saves])).abs().max()) < 5e-3

    dG = torch.full((2, N, 4, H), 9.0, device="cuda")
    L.call("ss_gru_bf16_bwd", dev(d_out.reshape(N, 2 * H)).data_ptr(), out.data_ptr(), save.data_ptr(), wtb.data_ptr(),
           lens.data_ptr(), B, T, H, dG.data_ptr(), 0.0, 0, 0, ws.data_ptr(), L.stream())
    torch.cuda.synchronize()
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
