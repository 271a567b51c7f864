"""The kernels of ONE steady training step in launch order (name, duration, gap to the previous kernel's end), from a
`rocprofv3 --kernel-trace` csv of bench.py: where the small launches cluster.   python tools/step_sequence.py <trace.csv> [step]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "proj_assign_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
# a step starts a little before its projection kernel (image encoder first): cut at the optimizer's last kernel of the previous step
lo, hi = marks[k], marks[k + 1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    m = re.match(r"([\w:]+(<[^(]{0,60})?)", n)
    return (m.group(1) if m else n)[:90]


prev_end = None
tot = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
    prev_end = e
    tot += e - s
print(f"# {hi - lo} launches, kernel time {tot / 1e6:.3f} ms, span {(int(rows[hi]['Start_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e6:.3f} ms")
