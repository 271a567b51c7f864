"""Times of the bf16 cost-volume backward kernels on configs[2]'s shapes (rows = 16 x 228 x 468 = 1 707 264; configs[4]:
8 x 171 x 468), events on the launch stream, clocks warm, against their algorithmic bytes:

  lin_backward 128->64 / 64->64     one pass (mlp_bwd_fused_bf16.hip) vs I2P_NO_FUSED_BF16=1 (rg_dgrad + wreg_wgrad_bf16)
  lin_backward_2src 64+64->128      rg_dgrad<4> + wreg_wgrad_bf16<128,128,two>
  pair_lin_backward 128->128        pair_bwd kernel
  cv_softmax_wsum_backward, pair_lin_forward, lin_forward

`python tools/time_bf16_bwd.py [--batch 16] [--nus]`"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from i2pnet_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=20, warm=40):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


def coef(c, dev, seed):
    g = torch.Generator().manual_seed(seed)
    mean = torch.randn(c, generator=g) * 0.5
    invstd = 0.5 + torch.rand(c, generator=g)
    gamma = 1.0 + 0.2 * torch.randn(c, generator=g)
    beta = 0.2 * torch.randn(c, generator=g)
    return torch.stack([mean, invstd * gamma, beta]).contiguous().to(dev), torch.cat([mean, invstd]).contiguous().to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--nus", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    be = ops.hip_backend()
    dev = "cuda"
    B, N, M = a.batch, (171 if a.nus else 228), 468
    rows = B * N * M
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    print(f"rows = {B} x {N} x {M} = {rows}")

    def report(name, t, nbytes):
        print(f"{name:44s} {t:8.1f} us  {nbytes / t / 1e3:7.0f} GB/s  = {nbytes / t / 1e3 / 8000:5.2f} of 8 TB/s   ({nbytes / 1e6:7.1f} MB algorithmic)", flush=True)

    want = lambda k: (not a.only) or k in a.only.split(",")
    if a.only and "cal" in a.only.split(","):
        # calibration of the PMC byte counters (tools/pmc_r05_bf16.py): kernels whose HBM traffic is known exactly on this tensor —
        # bwd_stats_bf16_kernel reads gz and y [rows,128] bf16 and writes nothing; pair_fwd3_bf16_kernel writes y [rows,128] bf16 and reads ~nothing
        yv, gz = rnd(rows, 128).to(BF), rnd(rows, 128, sc=0.1).to(BF)
        oc, omi = coef(128, dev, 6)
        for _ in range(10):
            be.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
        f, gg, bn_, bk_, w = rnd(B, N, 128), rnd(B, M, 128), rnd(B, N, 128), rnd(B, M, 128), rnd(128, 128, sc=128 ** -0.5)
        for _ in range(10):
            be.pair_lin_forward(f, gg, bn_, bk_, w, out_dtype=BF)
        torch.cuda.synchronize()
        print("calibration launches done")
        return
    if want("lin"):   # (under the PMC tool the A/B leg I2P_NO_FUSED_BF16 also runs: its kernels have other names)
        for cin, cout in ((128, 64), (64, 64)):
            x = rnd(rows, cin).to(BF); yv = rnd(rows, cout).to(BF); gz = rnd(rows, cout, sc=0.1).to(BF)
            w = rnd(cout, cin, sc=cin ** -0.5)
            oc, omi = coef(cout, dev, 5); ic, imi = coef(cin, dev, 6)
            ods = be.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
            run = lambda: be.lin_backward(gz, yv, oc, omi, ods, x, ic, imi, 0.1, w)
            nb = rows * (2 * cout + 2 * cin) * 2
            os.environ.pop("I2P_NO_FUSED_BF16", None)
            report(f"lin_backward {cin}->{cout} one pass", timeit(run), nb)
            os.environ["I2P_NO_FUSED_BF16"] = "1"
            report(f"lin_backward {cin}->{cout} dgrad + wgrad", timeit(run), nb)
            os.environ.pop("I2P_NO_FUSED_BF16", None)
            del x, yv, gz
    if want("2src"):
        xa, xb = rnd(rows, 64).to(BF), rnd(rows, 64).to(BF)
        yv, gz = rnd(rows, 128).to(BF), rnd(rows, 128, sc=0.1).to(BF)
        eadd = rnd(rows, 64, sc=0.1).to(BF)
        w = rnd(128, 128, sc=128 ** -0.5)
        oc, omi = coef(128, dev, 7); cfa, mia = coef(64, dev, 8); cfb, mib = coef(64, dev, 9)
        ods = be.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
        run = lambda: be.lin_backward_2src(gz, yv, oc, omi, ods, xa, cfa, mia, 0.1, xb, cfb, mib, 0.1, eadd, w)
        report("lin_backward_2src 64+64->128", timeit(run), rows * (2 * 128 + 2 * 128 + 64) * 2)
        del xa, xb, yv, gz, eadd
    if want("pair"):
        f, gg = rnd(B, N, 128), rnd(B, M, 128)
        bn_, bk_ = rnd(B, N, 128), rnd(B, M, 128)
        w = rnd(128, 128, sc=128 ** -0.5)
        yv, gz = rnd(rows, 128).to(BF), rnd(rows, 128, sc=0.1).to(BF)
        oc, omi = coef(128, dev, 6)
        ods = be.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
        run = lambda: be.pair_lin_backward(gz, f, gg, w, y=yv, out_coef=oc, out_mi=omi, out_dsums=ods)
        report("pair_lin_backward 128->128", timeit(run), rows * 2 * 128 * 2)
        report("pair_lin_forward 128->128", timeit(lambda: be.pair_lin_forward(f, gg, bn_, bk_, w, out_dtype=BF)), rows * 128 * 2)
        del yv, gz
    if want("fwd"):
        for cin, cout in ((128, 64), (64, 64)):
            x = rnd(rows, cin).to(BF); w = rnd(cout, cin, sc=cin ** -0.5); ic, _ = coef(cin, dev, 6)
            report(f"lin_forward {cin}->{cout}", timeit(lambda: be.lin_forward(x, ic, 0.1, w, out_dtype=BF)), rows * (cin + cout) * 2)
            del x
        xa, xb = rnd(rows, 64).to(BF), rnd(rows, 64).to(BF)
        cfa, _ = coef(64, dev, 8); cfb, _ = coef(64, dev, 9); w = rnd(128, 128, sc=128 ** -0.5)
        report("lin_forward_2src 64+64->128", timeit(lambda: be.lin_forward_2src(xa, cfa, 0.1, xb, cfb, 0.1, w)), rows * 256 * 2)
        del xa, xb
    if want("sm"):
        y5, y3 = rnd(rows, 64).to(BF), rnd(rows, 64).to(BF)
        c5, m5 = coef(64, dev, 3); c3, m3 = coef(64, dev, 4)
        out, msave = be.cv_softmax_wsum_forward(B, N, M, y5, c5, 0.1, y3, c3, 0.1)
        report("cv_softmax_wsum_forward", timeit(lambda: be.cv_softmax_wsum_forward(B, N, M, y5, c5, 0.1, y3, c3, 0.1)), rows * 128 * 2)
        go = rnd(B, N, 64)
        report("cv_softmax_wsum_backward", timeit(lambda: be.cv_softmax_wsum_backward(B, N, M, go, out, msave, y5, c5, m5, 0.1, y3, c3, 0.1)), rows * 256 * 2)


if __name__ == "__main__":
    main()
