#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel over a whole (eager) bench run, top 60 by total: a coarse scan for traffic amplification.
# Caveats: the run includes MIOpen's find trials, and a kernel's WRITE_SIZE includes write-backs of the PREVIOUS kernels' dirty L2 lines
# — only large, isolated kernels read cleanly (the dedicated passes tools/pmc_step.sh / pmc_chain.sh are the ones the bench cites).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcall_$c
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcall_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-configs 0 --loader-line 0 --graph 0 > /tmp/pmcall_$c.log 2>&1
  f=$(find /tmp/pmcall_$c -name '*counter_collection.csv' | head -1)
  python - "$f" $c <<'PY' > gpurun_out/r03_pmc_all_$c.txt
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k)[:90]
    e = agg[k]; e[0] += 1; e[1] += float(r["Counter_Value"])
tot = sorted(agg.items(), key=lambda kv: -kv[1][1])
for k, (n, v) in tot[:60]:
    print(f"{v / n:12.1f} KiB/launch  n={n:5d}  total={v / 1024:10.1f} MiB  {k}")
PY
  head -45 gpurun_out/r03_pmc_all_$c.txt
done
