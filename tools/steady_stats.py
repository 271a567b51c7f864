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


def grid_of(r):
    """blocks of the launch (rocprofv3 reports the grid in work-items): launches of one kernel on tensors of different sizes are
    kept apart, so that a row's average is the duration of ONE instantiation on ONE tensor size"""
    try:
        g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) * max(1, int(r.get("Grid_Size_Y") or 1)) * max(1, int(r.get("Grid_Size_Z") or 1))
        w = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1) * max(1, int(r.get("Workgroup_Size_Y") or 1)) * max(1, int(r.get("Workgroup_Size_Z") or 1))
        return g // max(1, w)
    except (TypeError, ValueError):
        return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--train-steps", type=int, default=0, help="total training steps in the trace (capture warm-up + warmup + steps); "
                    "with --tail-out: kernels launched after them = bench.py's live roofline section")
    ap.add_argument("--tail-out", default=None)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if "proj_assign_kernel" in r["Kernel_Name"]]
    assert len(marks) >= a.warmup + a.steps + 1, (len(marks), "proj_assign_kernel launches")
    t0, t1 = marks[a.warmup], marks[a.warmup + a.steps]
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    # (kernel, grid) does not separate launches of one instantiation on tensors of different sizes when the grid is capped (the
    # 256-block layer kernels run on the cv1- and on the cv2-size tensors: 315 us and 35 us averaged to "184 us").
    # The steps are replays of one hipGraph, so the ORDINAL of a launch among the launches of its (kernel, grid) inside the step
    # names the call site; ordinals whose mean durations lie within 30 % of each other are one row (one tensor size), others apart.
    marks_sel = [m for m in marks if t0 <= m <= t1]
    per_ord = defaultdict(lambda: defaultdict(lambda: [0, 0, 0]))      # (kernel, grid) -> ordinal -> [count, total, max]
    step_i, seen = 0, defaultdict(int)
    for r in sel:
        ts = int(r["Start_Timestamp"])
        while step_i + 1 < len(marks_sel) and ts >= marks_sel[step_i + 1]:
            step_i += 1; seen = defaultdict(int)
        key = (short(r["Kernel_Name"]), grid_of(r))
        d = int(r["End_Timestamp"]) - ts
        e = per_ord[key][seen[key]]
        seen[key] += 1
        e[0] += 1; e[1] += d; e[2] = max(e[2], d)
    agg = {}
    for key, ords in per_ord.items():
        items = sorted(ords.values(), key=lambda e: e[1] / e[0])
        cluster, lo, cid = None, None, 0
        for e in items:
            mean = e[1] / e[0]
            if cluster is None or mean > 1.3 * lo:
                cluster = [0, 0, 0]; lo = mean; agg[key + (cid,)] = cluster; cid += 1
            cluster[0] += e[0]; cluster[1] += e[1]; cluster[2] = max(cluster[2], e[2])
    tot = sum(v[1] for v in agg.values())
    wall = t1 - t0
    lines = [("kernel", "blocks", "calls_per_step", "total_us_per_step", "avg_us", "pct_of_kernel_time", "max_us")]
    for (k, g, _), (c, d, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append((k, str(g), f"{c / a.steps:.1f}", f"{d / a.steps / 1e3:.1f}", f"{d / c / 1e3:.2f}", f"{100.0 * d / tot:.2f}", f"{mx / 1e3:.1f}"))
    print(f"# steps={a.steps} wall_per_step_ms={wall / a.steps / 1e6:.3f} kernel_time_per_step_ms={tot / a.steps / 1e6:.3f} "
          f"launches_per_step={len(sel) / a.steps:.0f} distinct_kernels={len(agg)}")
    for ln in lines[: a.top + 1]:
        print(" | ".join(ln))
    if a.tail_out and a.train_steps >= 2:
        tail_stats(rows, marks, a.train_steps, a.tail_out)
    if a.out:
        with open(a.out, "w", newline="") as f:
            f.write(f"# steps={a.steps} wall_per_step_ms={wall / a.steps / 1e6:.3f} kernel_time_per_step_ms={tot / a.steps / 1e6:.3f} "
                    f"launches_per_step={len(sel) / a.steps:.0f}\n")
            csv.writer(f).writerows(lines)


def tail_stats(rows, marks, n_train, out):
    """per-kernel statistics of everything launched after the last training step (bench.py's roofline section)"""
    step = marks[n_train - 1] - marks[n_train - 2]
    t_end = marks[n_train - 1] + step
    agg = defaultdict(lambda: [0, 0, 10 ** 18, 0])
    for r in rows:
        if int(r["Start_Timestamp"]) < t_end:
            continue
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = agg[(short(r["Kernel_Name"]), grid_of(r))]
        e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
    with open(out, "w", newline="") as f:
        f.write("# kernels launched by bench.py after the timed steps (live roofline timings); rocprofv3 --kernel-trace\n")
        w = csv.writer(f)
        w.writerow(("kernel", "blocks", "calls", "avg_us", "min_us", "max_us"))
        for (k, g), (c, d, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            w.writerow((k, g, c, f"{d / c / 1e3:.2f}", f"{lo / 1e3:.2f}", f"{hi / 1e3:.2f}"))


if __name__ == "__main__":
    main()
