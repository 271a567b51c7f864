"""One steady training step as a timeline: start time since the previous step's adam_update_kernel ended, duration, queue — for the
two-stream step (image encoder on a second stream), where launch order says nothing about overlap.
python tools/step_timeline.py <kernel_trace.csv> [step]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adam_update_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
lo, hi = marks[k] + 1, marks[k + 1] + 1


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    m = re.match(r"([\w:]+(<[^(]{0,60})?)", n)
    return (m.group(1) if m else n)[:80]


t0 = int(rows[lo - 1]["End_Timestamp"])
queues = {}
busy = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    busy[q] = busy.get(q, 0) + e - s
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  q{q}  {short(r['Kernel_Name'])}")
span = int(rows[hi - 1]["End_Timestamp"]) - t0
print(f"# {hi - lo} launches, span {span / 1e6:.3f} ms; busy per queue (ms): " + ", ".join(f"q{q} {b / 1e6:.3f}" for q, b in sorted(busy.items())))
