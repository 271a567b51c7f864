"""Timing of the two-source 128-channel cost-volume layer (forward, backward) at the B=8 shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
from check_wreg import timeit

hip = ops.hip_backend(); dev = "cuda"
rows, ca, cb, co = 853632, 64, 64, 128
xa = torch.randn(rows, ca, device=dev); xb = torch.randn(rows, cb, device=dev); w = torch.randn(co, ca + cb, device=dev) / 11
one = lambda c: (torch.ones(c, device=dev), torch.zeros(c, device=dev))
coef_a, mi_a = hip.bn_finalize(rows, hip.bn_stats(xa), *one(ca), 1e-5)
coef_b, mi_b = hip.bn_finalize(rows, hip.bn_stats(xb), *one(cb), 1e-5)
y, st = hip.lin_forward_2src(xa, coef_a, 0.1, xb, coef_b, 0.1, w)
oc, om = hip.bn_finalize(rows, st, *one(co), 1e-5)
gz = torch.randn(rows, co, device=dev); e_add = torch.randn(rows, cb, device=dev)
ods = torch.zeros(ops.BN_REPLICAS * 2 * co, dtype=torch.float64, device=dev)
tf = timeit(lambda: hip.lin_forward_2src(xa, coef_a, 0.1, xb, coef_b, 0.1, w), iters=30, warm=100)
tb = timeit(lambda: hip.lin_backward_2src(gz, y, oc, om, ods, xa, coef_a, mi_a, 0.1, xb, coef_b, mi_b, 0.1, e_add, w), iters=30, warm=60)
print(f"two-source 64+64 -> 128 on {rows} rows: forward {tf:.1f} us, backward (dgrad + wgrad) {tb:.1f} us")
