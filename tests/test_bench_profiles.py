"""CPU: which committed PMC summary each block of the bench line reads (bench.py:latest_profile / pmc_lookup).

Round 3's driver line carried the 48x96 kernel's HBM bytes and MFMA occupancy for the 64x64 headline because the newest file by
NAME was a ``_ship_`` one; the selection is by exact name pattern now and the frame geometry is part of the kernel stem."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


LISTING = ["round2_c_pmc_traffic.json", "round3_e_pmc_traffic.json", "round3_f_pmc_traffic.json", "round3_f_pmc_mfma.json",
           "round3_f_ship_pmc_traffic.json", "round3_f_ship_pmc_mfma.json", "round3_f_c5_pmc_traffic.json", "round3_f_c5_pmc_mfma.json",
           "round3_e_pmc_issue.json", "round3_f_kernel_stats.csv", "round3_f_c5_stage_timers.txt", "round10_a_pmc_traffic.json.bak"]


def test_each_block_reads_its_own_profile():
    b = _bench()
    base = os.path.basename
    assert base(b.latest_profile("traffic", "config2", LISTING)) == "round3_f_pmc_traffic.json"
    assert base(b.latest_profile("mfma", "config2", LISTING)) == "round3_f_pmc_mfma.json"
    assert base(b.latest_profile("traffic", "shipped", LISTING)) == "round3_f_ship_pmc_traffic.json"
    assert base(b.latest_profile("mfma", "config5", LISTING)) == "round3_f_c5_pmc_mfma.json"
    assert base(b.latest_profile("issue", "config2", LISTING)) == "round3_e_pmc_issue.json"
    assert b.latest_profile("issue", "shipped", LISTING) is None
    # rounds sort numerically, tags alphabetically inside a round
    more = LISTING + ["round10_a_pmc_traffic.json", "round4_b_pmc_traffic.json", "round4_ab_pmc_traffic.json"]
    assert base(b.latest_profile("traffic", "config2", more)) == "round10_a_pmc_traffic.json"
    assert base(b.latest_profile("traffic", "config2", [n for n in more if "round10" not in n])) == "round4_b_pmc_traffic.json"


def test_committed_profiles_answer_for_the_right_kernel():
    """On the real profiles/ directory: config 2's dominant kernel is looked up with its geometry, and a profile of another
    frame size yields None instead of another instantiation's counters."""
    b = _bench()
    tr = b.latest_profile("traffic", "config2")
    assert tr is not None and "_ship_" not in tr and "_c5_" not in tr
    d = b.pmc_lookup(tr, b.kernel_stems("ss_roi_cnn_bwd", (64, 64)))
    assert d is not None and 0.55e9 < d["hbm_bytes_per_launch"] < 0.80e9  # 0.62 GB algorithmic
    assert b.pmc_lookup(tr, b.kernel_stems("ss_roi_cnn_bwd", (48, 96))) is None
    mf = b.latest_profile("mfma", "config2")
    m = b.pmc_lookup(mf, b.kernel_stems("ss_roi_cnn_bwd", (64, 64)))
    assert m is not None and 0.6 < m["mfma_busy_frac"] < 0.9
    ship = b.latest_profile("traffic", "shipped")
    if ship:
        assert b.pmc_lookup(ship, b.kernel_stems("ss_roi_cnn_bwd", (64, 64))) is None
        assert b.pmc_lookup(ship, b.kernel_stems("ss_roi_cnn_bwd", (48, 96))) is not None
        assert "config 2" not in json.load(open(ship))["source"]


def test_ambiguous_stem_is_not_averaged(tmp_path):
    b = _bench()
    p = tmp_path / "x.json"
    p.write_text(json.dumps({"kernels": {"void k<1>(P)": {"v": 1}, "void k<2>(P)": {"v": 2}}}))
    assert b.pmc_lookup(str(p), ["void k<"]) is None
    assert b.pmc_lookup(str(p), ["k<2>"]) == {"v": 2}
    assert b.pmc_lookup(str(p), ["void k<", "!<1>"]) == {"v": 2}  # "!text": names that contain text are out
    # the backward ROI kernel exists with and without a frame list since round 4; the bench's full clips run the one without
    q = tmp_path / "y.json"
    q.write_text(json.dumps({"kernels": {"void roi_cnn_bwd_kernel<Geom<64, 64>, false>(CnnBwdParams)": {"v": 1},
                                         "void roi_cnn_bwd_kernel<Geom<64, 64>, true>(CnnBwdParams)": {"v": 2}}}))
    assert b.pmc_lookup(str(q), b.kernel_stems("ss_roi_cnn_bwd", (64, 64))) == {"v": 1}
