"""profiles/<tag>_pmc_bf16_traffic.json from the two summaries tools/pmc_traffic.sh bf16 writes (gpurun_out/pmc_bf16_{FETCH,WRITE}_SIZE.txt):
HBM bytes per launch of the bf16 roofline kernel rg_fwd_kernel<4,true,false> = fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE,
both factors calibrated in the same run on kernels with exactly known traffic on the same [853632, 128] bf16 tensor
(bwd_stats_bf16_kernel: two bf16 streaming reads, nothing written; bn_act_fwd_bf16_kernel: fp32 write of the tensor).
    python tools/pmc_bf16_json.py <FETCH.txt> <WRITE.txt> <tag>"""
import json
import re
import sys

B, ROWS, C = 8, 8 * 228 * 468, 128
TENSOR_KIB = ROWS * C * 2 / 1024.0


def parse(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) avg=\s*([\d.]+)", ln)
        if m:
            out[m.group(1).strip()] = float(m.group(4))
    return out


def find(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return None


f, w, tag = parse(sys.argv[1]), parse(sys.argv[2]), sys.argv[3]
cal_r = find(f, "bwd_stats_bf16_kernel")            # reads 2 x tensor
cal_w = find(w, "bn_act_fwd_bf16_kernel")           # writes 2 x tensor bytes (fp32)
fx = 2 * TENSOR_KIB / cal_r
wx = 2 * TENSOR_KIB / cal_w
rec = {"B": B, "rows": ROWS, "calibration": {"bf16_tensor_KiB": TENSOR_KIB, "bwd_stats_bf16_FETCH_SIZE_KiB": cal_r,
                                             "bn_act_fwd_bf16_WRITE_SIZE_KiB": cal_w, "fetch_factor": round(fx, 4), "write_factor": round(wx, 4)},
       "source": [f"profiles/{tag}_pmc_bf16_FETCH_SIZE.txt", f"profiles/{tag}_pmc_bf16_WRITE_SIZE.txt"], "kernels": {}}
ALG = {"rg_fwd_kernel<4, true, false>": ROWS * C * 2 * 2, "rg_dgrad_kernel<4, false>": ROWS * C * 2 * 4, "pair_bwd_bf16_kernel": ROWS * C * 2 * 2,
       "wreg_wgrad_bf16_kernel<128, 128, false>": ROWS * C * 2 * 3, "pair_fwd_ps_kernel": ROWS * C * 2}
for name, alg in ALG.items():
    fv, wv = find(f, name), find(w, name)
    if fv is None and wv is None:
        continue
    by = round(((fv or 0.0) * fx + (wv or 0.0) * wx) * 1024.0)
    rec["kernels"][name] = {"FETCH_SIZE_KiB": fv, "WRITE_SIZE_KiB": wv, "bytes_per_launch": by, "algorithmic_bytes": alg,
                            "traffic_over_algorithmic": round(by / alg, 3)}
k = rec["kernels"].get("rg_fwd_kernel<4, true, false>")
if k:
    rec["bytes_per_launch_at_B8"] = k["bytes_per_launch"]
    rec["algorithmic_bytes_at_B8"] = k["algorithmic_bytes"]
print(json.dumps(rec, indent=1))
