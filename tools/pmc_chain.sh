#!/bin/bash
# SQ + traffic counters of the one-launch MLP chain kernels (csrc/mlp_chain.hip), separate passes (--pmc with --kernel-trace only).
# Workload tools/time_chain.py on case 0 (level 3: 29 184 rows, 64-64-128, K = 16; forward kernel only) and case 1 (level 4:
# 14 848 rows, 128-128-256, K = 16; forward + backward kernels).   usage: tools/pmc_chain.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp TIME_CHAIN_CASES=0,1
mkdir -p gpurun_out; out=gpurun_out/${1:-r03}_pmc_chain.txt; : > $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmcch_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcch_$i -- python tools/time_chain.py 0 > /tmp/pmcch_$i.log 2>&1
  f=$(find /tmp/pmcch_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] || { echo "# pass '$set': no counter file (counter unknown on this build?)" >> $out; continue; }
  python - "$f" <<'PY' >> $out
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    name = "chain_fwd_kernel" if "chain_fwd_kernel" in k else "chain_bwd_kernel<64>" if "chain_bwd_kernel" in k else "chain_reduce_kernel" if "chain_reduce" in k else None
    if not name: continue
    # grid size tells the cases apart: 456 blocks = level 3, 232 = level 4
    key = (name, r.get("Grid_Size", "?"), r["Counter_Name"])
    e = agg.setdefault(key, [0, 0.0]); e[0] += 1; e[1] += float(r["Counter_Value"])
for (name, grid, c), (n, v) in agg.items(): print(f"{name:22s} grid_threads={grid:>8s}  {c:30s} n={n:4d} avg={v / n:.6g}")
PY
done
cat $out
