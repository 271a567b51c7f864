#!/bin/bash
# SQ + traffic counters of the level-1 grouping kernel alone (workload tools/time_sa_l1.py: three input densities), separate passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out; out=gpurun_out/${1:-r03}_pmc_sa_l1.txt; : > $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmcsa_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcsa_$i -- python tools/time_sa_l1.py > /tmp/pmcsa_$i.log 2>&1
  f=$(find /tmp/pmcsa_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' >> $out
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "sa_l1_kernel" not in r["Kernel_Name"]: continue
    e = agg.setdefault(r["Counter_Name"], [0, 0.0]); e[0] += 1; e[1] += float(r["Counter_Value"])
for c, (n, v) in agg.items(): print(f"sa_l1_kernel<9,true>  {c:28s} n={n:4d} avg={v / n:.5g}   (mean over the three input densities, 300 launches each)")
PY
done
cat $out
