#!/bin/bash
# PMC passes over the image-block tail kernels (workload tools/pmc_img.py): FETCH_SIZE / WRITE_SIZE in separate passes, two SQ passes,
# kernel-trace only.  Writes gpurun_out/<tag>_pmc_img_{FETCH_SIZE,WRITE_SIZE,SQ1,SQ2}.txt (every dispatch, in launch order).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r04}
run() {
  name=$1; shift
  rm -rf /tmp/pmci_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmci_$name -- python tools/pmc_img.py > /tmp/pmci_$name.log 2>&1
  tail -3 /tmp/pmci_$name.log | cut -c1-200
  f=$(find /tmp/pmci_$name -name '*counter_collection.csv' | head -1)
  PMC_KEEP="img_,bn_stats_v4,bn_act_fwd_v4" python tools/pmc_step_summary.py "$f" > gpurun_out/${tag}_pmc_img_$name.txt
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run SQ2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
grep -h "shape\|calibration" /tmp/pmci_FETCH_SIZE.log > gpurun_out/${tag}_pmc_img_shapes.txt
