"""Per-kernel statistics of the TIMED steps of a `rocprofv3 --kernel-trace` run of bench.py.

`--stats` of the whole process is dominated by MIOpen's first-call solver search in the warm-up
steps.  The projection's `proj_assign_kernel` runs exactly once per training step (and once more
in the roofline section), so its timestamps delimit the steps:

    python tools/steady_stats.py <kernel_trace.csv> --warmup W --steps K [--out summary.csv]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:150]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if "proj_assign_kernel" in r["Kernel_Name"]]
    assert len(marks) >= a.warmup + a.steps + 1, (len(marks), "proj_assign_kernel launches")
    t0, t1 = marks[a.warmup], marks[a.warmup + a.steps]
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    agg = defaultdict(lambda: [0, 0])
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = agg[short(r["Kernel_Name"])]
        e[0] += 1; e[1] += d
    tot = sum(v[1] for v in agg.values())
    wall = t1 - t0
    lines = [("kernel", "calls_per_step", "total_us_per_step", "avg_us", "pct_of_kernel_time")]
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append((k, f"{c / a.steps:.1f}", f"{d / a.steps / 1e3:.1f}", f"{d / c / 1e3:.2f}", f"{100.0 * d / tot:.2f}"))
    print(f"# steps={a.steps} wall_per_step_ms={wall / a.steps / 1e6:.3f} kernel_time_per_step_ms={tot / a.steps / 1e6:.3f} "
          f"launches_per_step={len(sel) / a.steps:.0f} distinct_kernels={len(agg)}")
    for ln in lines[: a.top + 1]:
        print(" | ".join(ln))
    if a.out:
        with open(a.out, "w", newline="") as f:
            f.write(f"# steps={a.steps} wall_per_step_ms={wall / a.steps / 1e6:.3f} kernel_time_per_step_ms={tot / a.steps / 1e6:.3f} "
                    f"launches_per_step={len(sel) / a.steps:.0f}\n")
            csv.writer(f).writerows(lines)


if __name__ == "__main__":
    main()
