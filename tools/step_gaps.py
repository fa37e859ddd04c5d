#!/usr/bin/env python3
"""Diagnostic: where the GPU is IDLE inside one training step.  From a rocprofv3 kernel trace (see tools/timeline.py for the
command) take the last complete step, merge the busy intervals of all queues and list the gaps -- which kernel ended before the
gap, which one started after it -- and the total.  A step is GPU-bound; its idle time is what launch boundaries, stream joins and
dependent launches cost.

    python tools/step_gaps.py gpurun_out/tl/*/tl_kernel_trace.csv
"""
import csv
import glob
import sys

from timeline import short


def main():
    rows = []
    for a in sys.argv[1:]:
        for p in glob.glob(a):
            rows += list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [k for k, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0 = int(rows[ends[-2]]["End_Timestamp"])  # the previous step's last kernel ends here
    iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, short(r["Kernel_Name"])) for r in step)
    total = iv[-1][1]
    cur_end, last = 0, "(previous step's adam_clip)"
    gaps = []
    for s, e, n in iv:
        if s > cur_end:
            gaps.append((s - cur_end, cur_end, last, n))
        if e > cur_end:
            cur_end, last = e, n
    idle = sum(g[0] for g in gaps)
    print(f"step {total / 1e3:.1f} us from the end of the previous step's last kernel; {len(step)} kernels; GPU idle {idle / 1e3:.1f} us in {len(gaps)} gaps")
    for d, at, a, b in sorted(gaps, reverse=True)[:25]:
        print(f"  {d / 1e3:6.2f} us at {at / 1e3:8.1f}: {a}  ->  {b}")


if __name__ == "__main__":
    main()
