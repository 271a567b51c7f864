"""Timing of the bf16 layer backward (dgrad + wgrad, wgrad alone) at the configs[2] cost-volume shape (batch 16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
from check_wreg import timeit
hip = ops.hip_backend(); dev = "cuda"; bf = torch.bfloat16
rows = 16 * 228 * 468
for cin, cout in [(128, 128), (128, 64), (64, 128), (64, 64)]:
    x = torch.randn(rows, cin, device=dev).to(bf); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    one = lambda c: (torch.ones(c, device=dev), torch.zeros(c, device=dev))
    y0, s0 = hip.lin_forward(x, None, 1.0, w, out_dtype=bf)
    oc, om = hip.bn_finalize(rows, s0, *one(cout), 1e-5)
    xin = torch.randn(rows, cin, device=dev).to(bf)
    sx = torch.zeros(ops.BN_REPLICAS * 2 * cin, dtype=torch.float64, device=dev)
    ic, im = hip.bn_finalize(rows, hip.bn_stats(xin.float()), *one(cin), 1e-5)
    gz = (torch.randn(rows, cout, device=dev) * 0.1).to(bf)
    ods = hip.bn_act_backward_stats_bf16(gz, y0, oc, om, 1.0)
    tw = timeit(lambda: hip.lin_backward(gz, y0, oc, om, ods, xin, ic, im, 0.1, w, need_gx=False), iters=20, warm=60)
    tb = timeit(lambda: hip.lin_backward(gz, y0, oc, om, ods, xin, ic, im, 0.1, w), iters=20, warm=20)
    by = rows * 2 * (2 * cout + cin)
    print(f"bf16 {cin}->{cout} on {rows} rows: wgrad {tw:7.1f} us = {by / tw / 1e3:5.0f} GB/s   backward {tb:7.1f} us", flush=True)
    del x, y0, xin, gz
for cin, cout in [(128, 128), (128, 64), (64, 128), (64, 64)]:
    x = torch.randn(rows, cin, device=dev).to(bf); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    ic, im = hip.bn_finalize(rows, hip.bn_stats(x.float()), torch.ones(cin, device=dev), torch.zeros(cin, device=dev), 1e-5)
    t = timeit(lambda: hip.lin_forward(x, ic, 0.1, w, out_dtype=bf), iters=30, warm=100)
    print(f"bf16 forward {cin}->{cout} on {rows} rows: {t:7.1f} us = {rows * 2 * (cin + cout) / t / 1e3:5.0f} GB/s", flush=True)
    del x
