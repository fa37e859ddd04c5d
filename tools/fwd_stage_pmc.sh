#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in stop0 stop1 stop2 stop4 full; do
  if [ $v = full ]; then unset SS_HOTPATH_LIB; else export SS_HOTPATH_LIB=silent_speech_amd/_ab/lib$v.so; fi
  rocprofv3 --pmc ${PMC:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA} --kernel-trace --output-format csv -d gpurun_out/stg_$v -o x -- python3 tools/fwd_stage_pmc.py > gpurun_out/stg_$v.log 2>&1
  find gpurun_out/stg_$v -name "*_kernel_trace.csv" -delete; find gpurun_out/stg_$v -name "*agent_info.csv" -delete
  grep "per launch" gpurun_out/stg_$v.log
done
