"""profiles/r05_pmc_bf16_traffic.json from two outputs of tools/pmc_kernels.sh over tools/time_bf16_bwd.py at batch 16:
    <cal>  : --only cal            (bwd_stats_bf16_kernel: reads 2 x [rows,128] bf16, writes nothing; pair_fwd3_bf16_kernel: writes [rows,128] bf16)
    <main> : --only lin,2src,pair  (the kernels of interest; the one-pass kernels run with I2P defaults)
HBM bytes per launch = fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE (KiB counters), both factors from <cal> in the same
session (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide streaming reads on gfx950).
    python tools/pmc_r05_bf16.py <cal.txt> <main.txt> > profiles/r05_pmc_bf16_traffic.json"""
import json
import re
import sys

ROWS = 16 * 228 * 468


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) avg=\s*([\d.eE+]+)", ln)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
    return out


cal, main = parse(sys.argv[1]), parse(sys.argv[2])
tensor_kib = ROWS * 128 * 2 / 1024.0
fx = 2 * tensor_kib / cal["bwd_stats_bf16_kernel"]["FETCH_SIZE"]
wx = tensor_kib / cal["pair_fwd3_bf16_kernel"]["WRITE_SIZE"]
ALG = {"bwd_fused_bf16_kernel<128>": (2 * 64 + 2 * 128) * 2, "bwd_fused_bf16_kernel<64>": 4 * 64 * 2, "bwd_fused2_bf16_kernel": (2 * 128 + 3 * 64 + 128) * 2,
       "pair_bwd2_bf16_kernel": 2 * 128 * 2, "pair_fwd3_bf16_kernel": 128 * 2, "rg_fwd_kernel<2, true, false>": None, "rg_fwd_kernel<4, true, false>": 256 * 2 * 1}
rec = {"_calibration": {"rows": ROWS, "bf16_tensor_KiB": tensor_kib, "fetch_factor": round(fx, 4), "write_factor": round(wx, 4),
                        "bwd_stats_bf16_FETCH_SIZE_KiB": cal["bwd_stats_bf16_kernel"]["FETCH_SIZE"],
                        "pair_fwd3_WRITE_SIZE_KiB": cal["pair_fwd3_bf16_kernel"]["WRITE_SIZE"]}}
for name, v in main.items():
    by = (v.get("FETCH_SIZE", 0.0) * fx + v.get("WRITE_SIZE", 0.0) * wx) * 1024.0
    alg = ALG.get(name)
    rec[name] = {"FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB": v.get("WRITE_SIZE"), "bytes_per_launch": round(by),
                 "bytes_per_row": round(by / ROWS, 3), "algorithmic_bytes_per_row": alg,
                 "traffic_over_algorithmic": round(by / ROWS / alg, 3) if alg else None}
print(json.dumps(rec, indent=1))
