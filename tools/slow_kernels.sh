#!/bin/bash
# every kernel of ONE graph-replayed training step longer than 35 us that is not a layer/conv kernel, with its grid:
# the list to scan for small kernels that are slower than their bytes justify.  usage (GPU box): bash tools/slow_kernels.sh   |   PATTERN=FillFunctor bash tools/slow_kernels.sh
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --loader-line 0 > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "proj_assign_kernel" in r["Kernel_Name"]]
a, b = marks[5], marks[6]            # one graph-replayed training step
for r in rows[a:b]:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    import os
    pat = os.environ.get("PATTERN")             # PATTERN=FillFunctor: every launch of that kernel with its neighbours
    if pat:
        if pat in n:
            i = rows.index(r)
            g = [r.get(k) for k in r if "Grid_Size" in k]
            print(f"{d:6.1f} us {g} after {rows[i - 1]['Kernel_Name'][:50]!r} before {rows[i + 1]['Kernel_Name'][:50]!r}")
    elif d > 35 and not any(k in n for k in ("lin_", "pair_bwd", "igemm", "img_", "Cijk")):
        g = [r.get(k) for k in r if "Grid" in k or "Workgroup" in k]
        print(f"{d:8.1f} us {g} {n[:60]}")
PY
