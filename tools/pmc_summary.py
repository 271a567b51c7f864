"""Average a PMC counter per kernel from a rocprofv3 counter_collection.csv:  python tools/pmc_summary.py <csv> [...]"""
import csv
import re
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])[:70]
        if not any(k in name for k in ("lin_", "bn_", "pair_", "reduce_partials", "rg_", "wgrad", "bwd_stats", "fcsk", "sa_l1", "wreg")):
            continue
        e = agg[(name, r["Counter_Name"])]
        e[0] += 1; e[1] += float(r["Counter_Value"])
    print("#", path)
    for (k, c), (n, v) in sorted(agg.items()):
        print(f"{k:72s} {c:12s} n={n:3d} avg={v / n:16.1f}")
