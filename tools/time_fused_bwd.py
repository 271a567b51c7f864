"""Time of the one-pass backward (csrc/mlp_wreg_fused.hip) of the cost-volume layers on the step's shapes: lin_backward on
[853632, 64] gradients with a 128- or 64-channel input, events on the launch stream, clocks warm.  I2P_NO_FUSED_BWD=1 times the
two-kernel form (wreg_dgrad + wreg_wgrad); I2P_OPS_LIB selects an ablation build (FUSED_ABL)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from i2pnet_amd import ops  # noqa: E402

be = ops.hip_backend()
dev = "cuda"
rows = 8 * 228 * 468
for cin, cout in ((128, 64), (64, 64)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(rows, cin, device=dev, generator=g)
    w = torch.randn(cout, cin, device=dev, generator=g) / cin ** 0.5
    gam_i = torch.ones(cin, device=dev); bet_i = torch.zeros(cin, device=dev)
    gam_o = torch.ones(cout, device=dev); bet_o = torch.zeros(cout, device=dev)
    in_coef, in_mi = be.bn_finalize(rows, be.bn_stats(x), gam_i, bet_i, 1e-5)
    y, ys = be.lin_forward(x, in_coef, 0.1, w)
    out_coef, out_mi = be.bn_finalize(rows, ys, gam_o, bet_o, 1e-5)
    gz = torch.randn(rows, cout, device=dev, generator=g) * 0.1
    ods = be.bn_act_backward_stats(gz, y, out_mi, gam_o, bet_o, 0.1)
    run = lambda: be.lin_backward(gz, y, out_coef, out_mi, ods, x, in_coef, in_mi, 0.1, w)
    for _ in range(60):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    by = rows * (2 * cout + 2 * cin) * 4
    print(f"{cin:3d} -> {cout:3d}: {t:7.1f} us per backward (incl. the slab reduction)  {by / t / 1e3:7.1f} GB/s on one read of gz, y, x + one write", flush=True)

# the two-source layer 64 + 64 -> 128 (round 6): one pass (wreg_bwd_fused_kernel<128,64,TWO>) — I2P_NO_FUSED_BWD2=1 times
# wreg_dgrad_kernel<128,128,true> + wreg_wgrad_kernel<128,128,true,true>
g = torch.Generator(device=dev).manual_seed(2)
xa = torch.randn(rows, 64, device=dev, generator=g); xb = torch.randn(rows, 64, device=dev, generator=g) * 2 + 0.3
w = torch.randn(128, 128, device=dev, generator=g) / 11
one = lambda c: (torch.ones(c, device=dev), torch.zeros(c, device=dev))
coef_a, mi_a = be.bn_finalize(rows, be.bn_stats(xa), *one(64), 1e-5)
coef_b, mi_b = be.bn_finalize(rows, be.bn_stats(xb), *one(64), 1e-5)
y, ys = be.lin_forward_2src(xa, coef_a, 0.1, xb, coef_b, 0.1, w)
out_coef, out_mi = be.bn_finalize(rows, ys, *one(128), 1e-5)
gz = torch.randn(rows, 128, device=dev, generator=g) * 0.1
e_add = torch.randn(rows, 64, device=dev, generator=g) * 0.1
ods = be.bn_act_backward_stats(gz, y, out_mi, *one(128), 0.1)
run = lambda: be.lin_backward_2src(gz, y, out_coef, out_mi, ods, xa, coef_a, mi_a, 0.1, xb, coef_b, mi_b, 0.1, e_add, w)
for _ in range(60):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
by = rows * (2 * 128 + 2 * 128 + 64) * 4
print(f"64+64 -> 128: {t:7.1f} us per backward (incl. the slab reduction)  {by / t / 1e3:7.1f} GB/s on one read of gz, y, xa, xb, e_add + one write; "
      f"{4 * rows * 128 * 128 / t / 1e6:6.1f} TFLOP/s", flush=True)

# the pair layer (first cost-volume layer) backward at batch 8: one pass (wreg_pair_bwd_fused_kernel, csrc/mlp_wreg_pair_fused.hip) —
# I2P_NO_PAIR_FUSED=1 times wreg_pair_dgrad_kernel + wreg_pair_wgrad_kernel; both incl. the slab reductions of the entry
B, N, M, C = 8, 228, 468, 128
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
f, gk, bn, bk, w = rnd(B, N, C), rnd(B, M, C), rnd(B, N, C), rnd(B, M, C), rnd(C, C) / C ** 0.5
y, sy = be.pair_lin_forward(f, gk, bn, bk, w)
oc, om = be.bn_finalize(B * N * M, sy, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1e-5)
gz = rnd(B * N * M, C) * 0.1
ods = be.bn_act_backward_stats(gz, y, om, torch.ones(C, device=dev), torch.zeros(C, device=dev), 0.1)
run = lambda: be.pair_lin_backward(gz, f, gk, w, y=y, out_coef=oc, out_mi=om, out_dsums=ods)
for _ in range(40):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f"pair 128x128: {t:7.1f} us per backward (incl. zero fill and 5 slab reductions)  {4 * B * N * M * C * C / t / 1e6:6.1f} TFLOP/s", flush=True)
