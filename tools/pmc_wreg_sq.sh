#!/bin/bash
# SQ counters (MFMA busy, waits, instruction mix) of the operands-in-registers layer kernels: two separate --pmc passes,
# kernel-trace only.  Workload: tools/pmc_traffic.py (forward, dgrad, wgrad of the 128->128 cost-volume layer at B=8).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/pmc_wreg_sq.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1)); rm -rf /tmp/pmcw_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcw_$i -- python tools/pmc_traffic.py > /tmp/pmcw_$i.log 2>&1
  f=$(find /tmp/pmcw_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' | tee -a gpurun_out/pmc_wreg_sq.txt
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "wreg" not in n: continue
    k = re.search(r"(wreg_\w+<[^>]*>)", n).group(1)
    e = agg.setdefault((k, r["Counter_Name"]), [0, 0.0]); e[0] += 1; e[1] += float(r["Counter_Value"])
for (k, c), (n, v) in agg.items(): print(f"{k:42s} {c:30s} n={n:2d} avg={v / n:.4g}")
PY
done
