#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files kept under profiles/.

    python tools/pmc_summary.py stats  gpurun_out/prof2      profiles/round1_b_kernel_stats.csv
    python tools/pmc_summary.py pmc    gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round1_pmc_traffic.json

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


def pmc(fetch_dir, write_dir, dst):
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
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py config 2, B=256",
               "kernels": res}, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(res), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
