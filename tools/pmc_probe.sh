#!/bin/bash
# PMC passes over ONE probe command (counters in their own runs, --kernel-trace only):
#   bash tools/pmc_probe.sh OUT.txt python tools/conv0_bwd_probe.py --launches 3
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
rm -rf /tmp/pp
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd "$R" && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pp/$i -- "$@" > /dev/null 2>&1)
done
python "$R/tools/pmc_summary.py" $(find /tmp/pp -name '*counter_collection.csv') > "$OUT" 2>&1
cat "$OUT"
