#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files kept under profiles/.

    python tools/pmc_summary.py stats  gpurun_out/prof2      profiles/round1_b_kernel_stats.csv
    python tools/pmc_summary.py pmc    gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round1_pmc_traffic.json [workload note]

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read, which is how every large read
of these kernels is issued, so the read side is doubled; WRITE_SIZE is exact for 16 B/lane stores."""
import collections
import csv
import glob
import json
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= 90 else name[:87] + "..."


def stats(src, dst):
    f = (glob.glob(src + "/*/*_kernel_stats.csv") + glob.glob(src + "/*_kernel_stats.csv"))[0]
    rows = list(csv.reader(open(f)))
    with open(dst, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(rows[0])
        for r in rows[1:]:
            w.writerow([short(r[0])] + r[1:])
    print("wrote", dst, len(rows) - 1, "kernels")


def pmc(fetch_dir, write_dir, dst, workload="bench.py config 2, B=256"):
    res = collections.defaultdict(dict)
    for key, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == key:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k.startswith(("void at::", "__amd")):
                continue
            res[k][key + "_KiB_avg_per_launch"] = sum(v) / len(v)
            res[k]["launches_" + key] = len(v)
    for k, d in res.items():
        rd = 2.0 * d.get("FETCH_SIZE_KiB_avg_per_launch", 0.0) * 1024
        wr = d.get("WRITE_SIZE_KiB_avg_per_launch", 0.0) * 1024
        d["hbm_read_bytes_per_launch_corrected"] = rd
        d["hbm_write_bytes_per_launch"] = wr
        d["hbm_bytes_per_launch"] = rd + wr
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), " + workload,
               "kernels": res}, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(res), "kernels")


def mfma(src_dir, dst, note=""):
    """MFMA-pipe occupancy per kernel from one pass of
        rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 \
                        SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace ...
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles summed over the 1 024 SIMDs against the
    GPU-active cycles of the dispatch (rocprofv3's own MfmaUtil expression; GRBM_GUI_ACTIVE arrives summed over the 8 XCDs,
    MI355X_MICROARCH.md 'DVFS give-back').  mfma_flops = MOPS x 512."""
    f = (glob.glob(src_dir + "/*/*_counter_collection.csv") + glob.glob(src_dir + "/*_counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, d in agg.items():
        if k.startswith(("void at::", "__amd")):
            continue
        avg = {c: sum(v) / len(v) for c, v in d.items()}
        out = {"launches": len(next(iter(d.values()))), **{c + "_avg_per_launch": v for c, v in avg.items()}}
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            out["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)
        fl = 512.0 * (avg.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + avg.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0))
        if fl:
            out["mfma_gflop_per_launch_counted"] = fl / 1e9
        res[k] = out
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                         "SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (one pass, kernel-trace only) " + note, "kernels": res},
              open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(res), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], *([" ".join(sys.argv[5:])] if len(sys.argv) > 5 else []))
