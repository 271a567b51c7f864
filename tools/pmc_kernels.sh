#!/bin/bash
# SQ / traffic counters of selected kernels of any workload, one rocprofv3 pass per counter set (kernel-trace only):
#   tools/pmc_kernels.sh <tag> <regex of kernel names> <command ...>
# writes gpurun_out/<tag>_pmc.txt: per (kernel, counter): launches and the average value per launch.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; pat=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/${tag}_pmc.txt; : > $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmck_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmck_$i -- "$@" > /tmp/pmck_$i.log 2>&1
  f=$(find /tmp/pmck_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] || { echo "# pass '$set': no counter file" >> $out; tail -2 /tmp/pmck_$i.log >> $out; continue; }
  python - "$f" "$pat" <<'PY' >> $out
import csv, sys, collections, re
agg = collections.OrderedDict()
pat = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not pat.search(k): continue
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", k)
    name = (m.group(1) if m else k)[:60]
    e = agg.setdefault((name, r["Counter_Name"]), [0, 0.0]); e[0] += 1; e[1] += float(r["Counter_Value"])
for (name, c), (n, v) in agg.items(): print(f"{name:60s} {c:28s} n={n:4d} avg={v / n:.6g}")
PY
done
cat $out
