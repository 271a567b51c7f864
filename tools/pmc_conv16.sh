#!/bin/bash
# SQ counters of conv16_kernel (csrc/image_conv16.hip; workload tools/time_image_conv16.py), separate passes, kernel-trace only.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp I2P_TIME_CONV16_ONLY=1
mkdir -p gpurun_out; out=gpurun_out/${1:-r04}_pmc_conv16.txt; : > $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmcc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcc_$i -- python tools/time_image_conv16.py > /tmp/pmcc_$i.log 2>&1
  f=$(find /tmp/pmcc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] || { echo "# pass '$set': no counter file" >> $out; tail -2 /tmp/pmcc_$i.log >> $out; continue; }
  python - "$f" <<'PY' >> $out
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv3x3_" not in k or "fin_kernel" in k: continue
    name = k[k.index("conv3x3"):].split("(")[0]
    e = agg.setdefault((name, r["Counter_Name"]), [0, 0.0]); e[0] += 1; e[1] += float(r["Counter_Value"])
for (name, c), (n, v) in agg.items(): print(f"{name:16s} {c:32s} n={n:4d} avg={v / n:.6g}")
PY
done
cat $out
