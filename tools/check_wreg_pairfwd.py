"""Timing of the pair-layer forward at the B=8 cost-volume shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
from check_wreg import timeit
hip = ops.hip_backend(); dev = "cuda"
B, N, M, C = 8, 228, 468, 128
f = torch.randn(B, N, C, device=dev); g = torch.randn(B, M, C, device=dev); bn = torch.randn(B, N, C, device=dev); bk = torch.randn(B, M, C, device=dev)
w = torch.randn(C, C, device=dev) / C ** 0.5
t = timeit(lambda: hip.pair_lin_forward(f, g, bn, bk, w), iters=30, warm=100)
print(f"pair forward {B}x{N}x{M} rows, 128 -> 128: {t:.1f} us = {2.0 * B * N * M * C * C / t / 1e6:.1f} TF, y written at {B * N * M * C * 4 / t / 1e3:.0f} GB/s")
