"""CPU: the C-ABI library loads and exports every symbol include/ss_hotpath.h declares; host-side
logic (module surface, flat bucket, sharding) that needs no GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ss_hotpath.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from silent_speech_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        from silent_speech_amd.build import build

        build(verbose=False)
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ss_hotpath.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.ss_abi_version() == 3
    assert lib.ss_status_string(-3) == b"unsupported shape"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from silent_speech_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_module_surface_and_flat_bucket():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W
    import silent_speech_amd as ss

    m = ss.BiGRUClassifier(84, 5, use_roi=True)
    sd = m.state_dict()
    ref = W.param_shapes(84, 5, True)
    assert list(sd) == list(ref) and all(tuple(sd[k].shape) == ref[k] for k in ref)
    assert sum(v.numel() for v in sd.values()) == 1079582
    # parameters are views of one contiguous bucket, 16-byte aligned each
    base = m.flat_params.data_ptr()
    for p in m.parameters():
        off = p.data_ptr() - base
        assert 0 <= off < m.flat_params.numel() * 4 and off % 16 == 0
    # load_state_dict writes through the views
    new = W.make_state_dict(3, 84, 5, True)
    m.load_state_dict(new)
    views = m._views_of(m.flat_params)
    for k, v in new.items():
        assert torch.equal(views[k], v)
    # CPU tensors are refused: the product has no CPU path
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(2, 4, 84), torch.tensor([4, 2]), torch.zeros(2, 4, 64, 64, dtype=torch.uint8))
    # live variant / 1 layer keeps the reference's keys too
    m1 = ss.BiGRUClassifier(84, 5, use_roi=False, gru_layers=1)
    assert list(m1.state_dict()) == list(W.param_shapes(84, 5, False, gru_layers=1))


def test_shard_range_partitions():
    from silent_speech_amd import shard_range

    for n, w in ((2048, 8), (256, 1), (10, 4), (7, 8)):
        got = [shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
    assert [hi - lo for lo, hi in (shard_range(2048, r, 8) for r in range(8))] == [256] * 8


def test_host_side_size_queries_need_no_gpu():
    """The size queries of the ABI are host code: the six stash sizes of every ROI shape the CNN kernels are built for, and the
    exchange workspace of the persistent recurrences (0 = the shape stays on the other form)."""
    import ctypes as C

    from silent_speech_amd import _lib

    lib = _lib.load()
    for H, W in ((64, 64), (48, 96), (32, 32)):
        a1, a2, i1, i2, m3, feat = _lib.cnn_stash_sizes(H, W)
        assert a1 >= 8 * (H // 2 + 2) * (W // 2 + 2) and a2 >= 16 * (H // 4 + 2) * (W // 4 + 2)
        assert (i1, i2, m3) == (8 * ((H // 2) * (W // 2) + 16), 16 * (H // 4) * (W // 4), 32 * (H // 4) * (W // 4)) and feat >= 50  # (argmax planes 16 B apart)
    with pytest.raises(RuntimeError):
        _lib.cnn_stash_sizes(40, 40)
    n = C.c_long(-1)
    assert lib.ss_gru_bf16_sync_bytes(256, 30, 512, C.byref(n)) == 0 and n.value > 0          # persistent bf16 recurrence
    big = n.value
    assert lib.ss_gru_bf16_sync_bytes(2048, 30, 512, C.byref(n)) == 0 and n.value == big       # clip chunks reuse one area
    assert lib.ss_gru_bf16_sync_bytes(256, 30, 1024, C.byref(n)) == 0 and n.value == 0         # W_hh no longer fits the registers
    assert lib.ss_gru_bf16_sync_bytes(256, 2000, 512, C.byref(n)) == 0 and n.value == 0        # step tags are 10 bits
    assert lib.ss_gru_bf16_sync_bytes(0, 30, 512, C.byref(n)) != 0
    assert _lib.gru_sync_bytes(256, 30, 192) > 0 and _lib.gru_sync_bytes(4096, 60, 192) == 0   # f32: one CU per slice for big batches


def test_bf16_weight_gradient_group_size_query_needs_no_gpu():
    """The grouped bf16 weight-gradient GEMM (ss_gemm_bf16_splitk_group): shape-only problem records of the config-5 GRU layers, the
    scratch size query (host code) and its argument checks -- K must be whole 64-deep tiles, at most eight problems per group."""
    from types import SimpleNamespace

    from silent_speech_amd import _lib
    from silent_speech_amd import engine_bf16 as E

    cfg = SimpleNamespace(hidden=512, in_dim=148)
    l1 = E.dw_problems(cfg, 256, 30, 1, 1024)
    l0 = E.dw_problems(cfg, 256, 30, 0, 152)
    assert [(q.M, q.N, q.K) for q in l1] == [(1536, 1024, 7680), (1024, 512, 7424), (512, 512, 7424)]
    assert E.dw_group_ok(l1) and E.dw_group_ok(l0)
    assert _lib.gemm_group_ws_floats(l1, bf16=True) > 0 and _lib.gemm_group_ws_floats(l1 + l0, bf16=True) > 0
    assert not E.dw_group_ok(E.dw_problems(cfg, 5, 7, 1, 1024))                 # K = 35 / 30: the per-problem launches take these
    with pytest.raises(RuntimeError):
        _lib.gemm_group_ws_floats(E.dw_problems(cfg, 5, 7, 1, 1024), bf16=True)  # ... and the entry point refuses them
    with pytest.raises(RuntimeError):
        _lib.gemm_group_ws_floats(l1 + l0 + l1, bf16=True)                      # nine problems



def test_bf16_weight_gradient_scratch_covers_every_group_backward_launches():
    """with three GRU layers backward() launches the groups {layer 2, layer 1} and {layer 0}; the scratch was
    sized for one layer or for all layers, and {2, 1} is larger than either.  The sizing now replays the launch rule."""
    from types import SimpleNamespace

    from silent_speech_amd import _lib
    from silent_speech_amd import engine_bf16 as E

    for layers in (1, 2, 3, 4):
        cfg = SimpleNamespace(hidden=512, in_dim=148, gru_layers=layers)
        kp = [152] + [1024] * (layers - 1)
        groups = E.dw_group_schedule(cfg, 256, 30, kp)
        assert sum(len(g) for g in groups) == 3 * layers and all(len(g) <= E.DW_GROUP_MAX for g in groups)
        # the schedule is what backward() does: walk top-down with the same flush rule
        pending, launched = 0, []
        for l in range(layers - 1, -1, -1):
            pending += 3
            if E.dw_flush_after(cfg, l, pending):
                launched.append(pending)
                pending = 0
        assert pending == 0 and launched == [len(g) for g in groups]
    cfg = SimpleNamespace(hidden=512, in_dim=148, gru_layers=3)
    groups = E.dw_group_schedule(cfg, 256, 30, [152, 1024, 1024])
    assert [len(g) for g in groups] == [6, 3]
    need = max(_lib.gemm_group_ws_floats(g, bf16=True) for g in groups)
    per_layer = max(_lib.gemm_group_ws_floats(E.dw_problems(cfg, 256, 30, l, k), bf16=True) for l, k in ((0, 152), (1, 1024), (2, 1024)))
    assert need >= _lib.gemm_group_ws_floats(groups[0], bf16=True) > per_layer  # the case that overflowed
