#!/bin/bash
# PMC passes over the in-step kernels (workload tools/pmc_step.py): FETCH_SIZE and WRITE_SIZE in separate passes (they do not
# fit one), two SQ passes; kernel-trace only.  Writes gpurun_out/r03_pmc_{FETCH_SIZE,WRITE_SIZE,SQ1,SQ2}.txt (every dispatch of the
# selected kernels, in launch order) and gpurun_out/r03_pmc_traffic.json (what bench.py's `traffic` fields read from profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r03}
run() {   # name, counters...
  name=$1; shift
  rm -rf /tmp/pmcs_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcs_$name -- python tools/pmc_step.py > /tmp/pmcs_$name.log 2>&1
  tail -1 /tmp/pmcs_$name.log
  f=$(find /tmp/pmcs_$name -name '*counter_collection.csv' | head -1)
  python tools/pmc_step_summary.py "$f" > gpurun_out/${tag}_pmc_$name.txt
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run SQ2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
python tools/pmc_step_summary.py --json gpurun_out/${tag}_pmc_FETCH_SIZE.txt gpurun_out/${tag}_pmc_WRITE_SIZE.txt > gpurun_out/${tag}_pmc_traffic.json
cat gpurun_out/${tag}_pmc_traffic.json | head -60
