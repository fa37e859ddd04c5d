"""Build ``libss_hotpath.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m silent_speech_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libss_hotpath.so")
SOURCES = ["features.hip", "batch.hip", "stream.hip", "crop_resize.hip", "roi_cnn.hip", "roi_cnn_bwd.hip", "roi_cnn_generic.hip", "gemm.hip", "gru.hip", "pool_head.hip", "tail.hip", "optim.hip", "gemm_bf16.hip", "gru_bf16.hip", "cnn_bf16.hip", "cnn_bf16_bwd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, stamp: bool = False) -> str:
    """``stamp=True`` builds the diagnostic library (in-kernel stage timers) next to the real one."""
    if stamp:
        # A/B experiments: SS_STAMP_DEFS="-DSS_VAR=1" SS_STAMP_OUT=libss_hotpath_stampB.so builds a second variant
        out = os.path.join(HERE, os.environ.get("SS_STAMP_OUT", "libss_hotpath_stamp.so"))
        srcs = [os.path.join(CSRC, s) for s in SOURCES]
        cmd = [_hipcc(), *FLAGS, "-DSS_STAMP", *os.environ.get("SS_STAMP_DEFS", "").split(), "-shared", "-o", out, *srcs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return out
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "ss_hotpath.h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [_hipcc(), *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(OUT, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, stamp="--stamp" in sys.argv)
