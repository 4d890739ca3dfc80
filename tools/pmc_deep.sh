#!/bin/bash
# Deeper PMC passes over tools/model_kernel_bench.py (counters in their own runs, --kernel-trace only):
#   bash tools/pmc_deep.sh OUT.txt      (on the GPU box; needs rocprofv3)
OUT=${1:-gpurun_out/pmc_deep.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
rm -rf /tmp/pd
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pd/$i -- python "$R/tools/model_kernel_bench.py" --launches 3 > /dev/null 2>&1)
done
python "$R/tools/pmc_summary.py" $(find /tmp/pd -name '*counter_collection.csv') > "$OUT" 2>&1
wc -l "$OUT"
