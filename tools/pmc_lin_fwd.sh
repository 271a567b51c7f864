#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1)); rm -rf /tmp/pmcl_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcl_$i -- python tools/pmc_lin_fwd.py > /tmp/pmcl_$i.log 2>&1
  tail -1 /tmp/pmcl_$i.log
  f=$(find /tmp/pmcl_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' | tee -a gpurun_out/pmc_lin_fwd.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lin_fwd2_kernel" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in enumerate(sorted(by)):
    print("launch", k, ("full" if k < 3 else "mfma-only"), " ".join(f"{n}={v:.4g}" for n, v in by[d].items()))
PY
done
