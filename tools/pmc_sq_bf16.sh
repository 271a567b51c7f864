#!/bin/bash
# SQ counters (MFMA busy, wait states, LDS conflicts) of the bf16 layer kernels and the fp32 dgrad: separate pmc passes, kernel-trace only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/pmc_sq_bf16.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcs_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcs_$i -- python tools/pmc_traffic_bf16.py > /tmp/pmcs_$i.log 2>&1
  tail -1 /tmp/pmcs_$i.log
  f=$(find /tmp/pmcs_$i -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py "$f" | tee -a gpurun_out/pmc_sq_bf16.txt
done
