"""Reduce a rocprofv3 counter_collection.csv of tools/pmc_step.py.

    python tools/pmc_step_summary.py <counter_collection.csv>            -> one line per (kernel, counter): n, average over the
                                                                            dispatches after the first (warm-up) iteration, and
                                                                            the per-dispatch values in launch order
    python tools/pmc_step_summary.py --json <FETCH.txt> <WRITE.txt>      -> the traffic record bench.py reads (profiles/r03_pmc_traffic.json)
"""
import csv
import json
import re
import sys
from collections import OrderedDict

import os
KEEP = tuple(os.environ["PMC_KEEP"].split(",")) if os.environ.get("PMC_KEEP") else (
    "wreg_", "sm_fwd", "sm_bwd", "outer_sum", "pair_sum", "pair_bias", "bn_stats_v4", "bn_act_fwd_v4", "sa_l1_kernel", "fcsk_kernel",
    "reduce_partials")
B = 8
ROWS = B * 228 * 468
TENSOR_KIB = ROWS * 128 * 4 / 1024.0


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def table(path):
    agg = OrderedDict()
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        k = short(r["Kernel_Name"])
        if not any(s in k for s in KEEP):
            continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return agg


def dump(path):
    print("#", path)
    for (k, c), vals in table(path).items():
        per_iter = max(len(vals) // 4, 1)                 # 4 iterations in the workload, the first is warm-up
        steady = vals[per_iter:] or vals
        print(f"{k:48s} {c:26s} n={len(vals):3d} avg={sum(steady) / len(steady):16.1f} all=" + ",".join(f"{v:.0f}" for v in vals))


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) avg=\s*([\d.]+) all=(.*)", ln)
        if m:
            out[m.group(1).strip()] = (float(m.group(4)), [float(v) for v in m.group(5).split(",")])
    return out


# algorithmic bytes per launch at batch 8 (rows = 853632) of the in-step layer kernels: rows*(channels read + written)*4 B
ALG = {"wreg_bwd_fused_kernel<64, 128>": ROWS * (2 * 64 + 2 * 128) * 4, "wreg_bwd_fused_kernel<64, 64>": ROWS * 4 * 64 * 4,
       "wreg_dgrad_kernel<64, 128, false>": ROWS * (2 * 64 + 2 * 128) * 4, "wreg_dgrad_kernel<64, 64, false>": ROWS * 4 * 64 * 4,
       "wreg_dgrad_kernel<128, 128, true>": ROWS * (2 * 128 + 2 * 128 + 64) * 4,          # + the added gradient ga3 [rows, 64]
       "wreg_fwd_kernel<128, 64, true, false>": ROWS * (128 + 64) * 4, "wreg_fwd_kernel<64, 64, true, false>": ROWS * 128 * 4,
       "wreg_fwd_kernel<128, 128, true, true>": ROWS * 256 * 4, "wreg_wgrad_kernel<64, 128, true, false>": ROWS * (2 * 64 + 128) * 4,
       "wreg_wgrad_kernel<64, 64, true, false>": ROWS * 3 * 64 * 4, "wreg_wgrad_kernel<128, 128, true, true>": ROWS * (2 * 128 + 128) * 4,
       "wreg_pair_fwd_kernel<128, 128>": ROWS * 128 * 4, "wreg_pair_dgrad_kernel<128, 128>": ROWS * 2 * 128 * 4,
       "wreg_pair_wgrad_kernel<128, 128>": ROWS * 2 * 128 * 4}


def to_json(fetch_txt, write_txt):
    f, w = parse(fetch_txt), parse(write_txt)
    cal_r = f.get("bn_stats_v4", (None,))[0]
    cal_w = w.get("bn_act_fwd_v4", (None,))[0]
    fetch_x = round(TENSOR_KIB / cal_r, 4) if cal_r else 2.0          # gfx950: FETCH_SIZE reports half the bytes of a wide streaming read
    write_x = round(TENSOR_KIB / cal_w, 4) if cal_w else 1.0
    rec = {"B": B, "rows": ROWS,
           "calibration": {"tensor_KiB": TENSOR_KIB, "bn_stats_v4_FETCH_SIZE_KiB": cal_r, "bn_act_fwd_v4_WRITE_SIZE_KiB": cal_w,
                           "fetch_factor": fetch_x, "write_factor": write_x,
                           "rule": "bytes = fetch_factor*FETCH_SIZE + write_factor*WRITE_SIZE (KiB); factors = known bytes / reported on the two "
                                   "calibration kernels in the same run (MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide streaming read on gfx950)"},
           "source": [fetch_txt.replace("gpurun_out", "profiles"), write_txt.replace("gpurun_out", "profiles")], "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if k in ("bn_stats_v4", "bn_act_fwd_v4") or k.startswith("reduce_partials"):
            continue
        fv, wv = f.get(k, (0.0, []))[0], w.get(k, (0.0, []))[0]
        e = {"B": B, "FETCH_SIZE_KiB": fv, "WRITE_SIZE_KiB": wv, "bytes_per_launch": round((fetch_x * fv + write_x * wv) * 1024.0)}
        if k in ALG:
            e["algorithmic_bytes"] = ALG[k]
            e["traffic_over_algorithmic"] = round(e["bytes_per_launch"] / ALG[k], 3)
        if "sa_l1" in k or "fcsk" in k:                   # one density per iteration: keep the per-dispatch values
            e["per_density_KiB"] = {"FETCH_SIZE": f.get(k, (0, []))[1], "WRITE_SIZE": w.get(k, (0, []))[1],
                                    "order": "warm-up scan, 8192-pt scan, 8192-pt centre-aligned, 150000-pt scan"}
        rec["kernels"][k] = e
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        to_json(sys.argv[2], sys.argv[3])
    else:
        for p in sys.argv[1:]:
            dump(p)
