#!/usr/bin/env python3
"""Diagnostic: text timeline of one training step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 3 --warmup 2 \
        --no-cpu-baseline --no-kernel-times [--micro-batches 2]
    python tools/timeline.py gpurun_out/tl/*/tl_kernel_trace.csv

Prints, for the last complete step (delimited by the adam_clip kernel), every kernel's start/end in
microseconds relative to the step start, its queue and its grid, so overlap between streams can be read off.
"""
import csv
import glob
import sys


def short(full):
    n = full.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:52]


def main():
    paths = []
    for a in sys.argv[1:]:
        paths += glob.glob(a)
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append(r)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [k for k, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    if len(ends) < 2:
        print("fewer than two steps in the trace")
        return
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    print(f"step of {len(step)} kernels, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    busy = {}
    for r in step:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        name = short(r["Kernel_Name"])
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
        q = r.get("Queue_Id", "?")
        busy[name] = busy.get(name, 0.0) + e - s
        if e - s >= 8.0:
            print(f"  q{q:>3} {s:9.1f} -> {e:9.1f}  ({e - s:7.1f} us)  wg={grid:<6d} {name}")
    print("busy time per kernel (us):")
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  {v:9.1f}  {k}")


if __name__ == "__main__":
    main()
