#!/bin/bash
# One profiling round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench args...]      e.g.  tools/profile_round.sh r2a   /   tools/profile_round.sh r2a_c5 --config 5
# kernel-trace statistics, then the PMC passes in separate runs (counters never together with other trace domains),
# the program itself directly behind "--".  Summaries are made by tools/pmc_summary.py into profiles/ afterwards.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
X="--no-cpu-baseline --no-kernel-times --no-config4 --no-config5 --no-shipped"  # one config per profile: the headline, or --config 5
B="python3 bench.py --steps 3 --warmup 2 $X $*"
# (the statistics pass runs the default 20 + 5 steps: with 3 + 2 the cold first launches pull a kernel's average 5 % up)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o st -- python3 bench.py --steps 20 --warmup 5 $X $* > gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -o pf -- $B > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -o pw -- $B > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/${tag}_mfma -o pm -- $B > gpurun_out/${tag}_mfma.log 2>&1
# instruction mix and stalls (diagnostic): LDS / VALU issue, waits, bank conflicts
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/${tag}_issue -o pi -- $B > gpurun_out/${tag}_issue.log 2>&1
# the trace CSVs are large: keep the counter / stats tables only
find gpurun_out/${tag}_* -name "*_kernel_trace.csv" -delete
find gpurun_out/${tag}_* -name "*agent_info.csv" -delete
echo profiled $tag
