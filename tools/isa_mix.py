#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc -S listing, per basic block: how many matrix, vector, scalar, LDS and
memory instructions a block issues.  The CNN kernels are bound by instruction ISSUE (f32 MFMA and VALU take turns on a SIMD:
docs/LAB_NOTES.md 8c-2 (6)), so the per-frame count of non-MFMA instructions is the quantity to drive down; blocks are listed in
program order with their first line, which is enough to match them to the stages of the source.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s silent_speech_amd/csrc/roi_cnn.hip
    python tools/isa_mix.py /tmp/k.s 'roi_cnn_fwd_kernel.*64ELi64' [--min-mfma 1]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_", "buffer_load_dword") ) and op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    min_mfma = int(sys.argv[sys.argv.index("--min-mfma") + 1]) if "--min-mfma" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.endswith(":") is False and re.match(r"^_Z\S+:", l) and pat.search(l))
    blocks, cur, name = [], collections.Counter(), "entry"
    first = start
    total = collections.Counter()
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith(".Lfunc_end"):
            break
        if not l or l.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", l):
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, first, cur))
            cur, name, first = collections.Counter(), l.split(":")[0], i + 1
            continue
        op = l.split()[0]
        cur[classify(op)] += 1
        total[classify(op)] += 1
        if op.startswith("ds_"):
            cur[op] += 0
    blocks.append((name, first, cur))
    print(f"{'block':>12} {'line':>6} {'mfma':>5} {'valu':>5} {'salu':>5} {'lds':>5} {'vmem':>5} {'wait':>5} {'bar':>4}  valu+salu+lds per mfma")
    for name, first, c in blocks:
        n = sum(c.values())
        if n == 0 or c["mfma"] < min_mfma:
            continue
        ratio = (c["valu"] + c["salu"] + c["lds"]) / c["mfma"] if c["mfma"] else float("nan")
        print(f"{name:>12} {first:6d} {c['mfma']:5d} {c['valu']:5d} {c['salu']:5d} {c['lds']:5d} {c['vmem']:5d} {c['wait']:5d} {c['barrier']:4d}  {ratio:6.2f}")
    print("total", dict(total))


if __name__ == "__main__":
    main()
