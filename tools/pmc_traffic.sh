#!/bin/bash
# two separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
# usage: tools/pmc_traffic.sh     (workload tools/pmc_traffic.py: the fp32 128->128 layer kernels; the bf16 cost-volume kernels have
# their own pass, tools/pmc_r05_bf16.py behind tools/pmc_r05.sh)
sfx=
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python tools/pmc_traffic$sfx.py > /tmp/pmc_$c.log 2>&1
  tail -2 /tmp/pmc_$c.log
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py "$f" | tee gpurun_out/pmc${sfx}_$c.txt
done
