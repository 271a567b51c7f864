"""per-kernel totals of a rocprofv3 kernel_trace.csv: calls, average and total microseconds (usage: trace_stats.py <trace.csv> [top])"""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:6d} calls  avg {t / n:9.1f} us  total {t:11.1f} us  {name[:110]}")
