import sys, torch
sys.path.insert(0, "/root/repo")
from i2pnet_amd import ops
sys.path.insert(0, "/root/repo/tools")
from time_bf16_bwd import timeit, coef
BF = torch.bfloat16
be = ops.hip_backend(); dev = "cuda"
B, N, M = 16, 228, 468; rows = B * N * M
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
en, ek = rnd(B, N, 64), rnd(B, M, 64)
xb = rnd(rows, 64).to(BF)
cfa, mia = coef(64, dev, 8); cfb, mib = coef(64, dev, 9); w = rnd(128, 128, sc=128 ** -0.5)
q16, q32, ste = be.outer_prep_bf16(en, ek)
ye, _ = be.outer_sum_bf16(en, ek)
print("outer_prep", timeit(lambda: be.outer_prep_bf16(en, ek)))
print("outer_sum ", timeit(lambda: be.outer_sum_bf16(en, ek)))
print("fwd stored", timeit(lambda: be.lin_forward_2src(ye, cfa, 0.1, xb, cfb, 0.1, w)))
print("fwd onload", timeit(lambda: be.lin_forward_2src_outer(en, q16, cfa, 0.1, xb, cfb, 0.1, w)))
yv, gz = rnd(rows, 128).to(BF), rnd(rows, 128, sc=0.1).to(BF)
eadd = rnd(rows, 64, sc=0.1).to(BF)
oc, omi = coef(128, dev, 7)
ods = be.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
print("bwd stored", timeit(lambda: be.lin_backward_2src(gz, yv, oc, omi, ods, ye, cfa, mia, 0.1, xb, cfb, mib, 0.1, eadd, w)))
print("bwd onload", timeit(lambda: be.lin_backward_2src_outer(gz, yv, oc, omi, ods, en, q16, cfa, mia, 0.1, xb, cfb, mib, 0.1, eadd, w)))
