#!/bin/bash
# Diagnostic: instruction-mix / stall counters of the two ROI-CNN kernels (run through gpurun from the repo root).
#   tools/pmc_issue.sh <tag>   -> gpurun_out/<tag>_issue*/…counter_collection.csv, summarised on stdout
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AB_STEPS=3
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/${tag}_issue1 -o p1 -- python3 tools/cnn_ab.py > gpurun_out/${tag}_issue1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/${tag}_issue2 -o p2 -- python3 tools/cnn_ab.py > gpurun_out/${tag}_issue2.log 2>&1
find gpurun_out/${tag}_issue* -name "*_kernel_trace.csv" -delete
find gpurun_out/${tag}_issue* -name "*agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections
for d in ("gpurun_out/${tag}_issue1", "gpurun_out/${tag}_issue2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "roi_cnn" not in k: continue
            acc[k.split("<")[0][-22:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
