#!/usr/bin/env python3
"""Diagnostic: the numbers of one bench.py JSON line that DESIGN.md quotes.   python tools/bench_digest.py gpurun_out/r4_b2.log"""
import json
import sys

l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print("config 2:", d["value"], "clips/s", d["ms_per_step"], "ms;", d.get("extra"))
r = d["roofline"]
print("  roofline:", r["kernel"], r["frac"], "avg ms", r["avg_launch_ms"], "traffic", r["traffic"], "busy", (r.get("mfma_busy") or {}).get("mfma_busy_frac"),
      r.get("profile_files"))
print("  step:", d["step_roofline"])
print("  kernels:", d["kernels_ms_per_step"])
if "config5" in d:
    c5 = d["config5"]
    print("config 5:", c5["value"], c5["ms_per_step"], c5["step_roofline"]["frac"], c5["roofline"])
    print("  kernels:", c5["kernels_ms_per_step"])
if "shipped" in d:
    for k, v in d["shipped"].items():
        if k.startswith("b"):
            print("shipped", k, v["value"], v["ms_per_step"], v["roofline"]["kernel"], v["roofline"]["frac"], v["roofline"].get("traffic"),
                  (v["roofline"].get("mfma_busy") or {}).get("mfma_busy_frac"), v["roofline"].get("profile_files"))
            print("   kernels:", v["kernels_ms_per_step"])
if "config4" in d:
    print("config 4:", d["config4"]["value"], d["config4"]["ms_per_step"], d["config4"]["roofline"]["frac"])
cb = d.get("cpu_baseline")
if cb:
    print("cpu:", cb["value"], cb["cores"], cb["sample"][:80], [(a["value"]) for a in cb.get("also", [])])
    for k in ("config5", "shipped"):
        if k in d and "cpu_baseline" in d[k]:
            print("cpu", k, d[k]["cpu_baseline"]["value"])
if "padded_batches" in d:
    print("padded batches:", d["padded_batches"])
if "sliding_window_serving" in d:
    print("sliding-window serving:", d["sliding_window_serving"]["value"], d["sliding_window_serving"]["variants"])
