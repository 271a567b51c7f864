"""Time the gen-2 backward (dgrad + wgrad) on the cost-volume shapes; run under rocprofv3 for the split."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from bench_kernels import timeit
from i2pnet_amd import ops
hip = ops.hip_backend(); dev = "cuda"
B = 8
for cin, cout in [(128, 128), (128, 64), (64, 64)]:
    rows = B * 228 * 468
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    gam = torch.ones(cin, device=dev); bet = torch.zeros(cin, device=dev)
    go = torch.ones(cout, device=dev); bo = torch.zeros(cout, device=dev)
    sx = torch.zeros(ops.BN_REPLICAS * 2 * cin, dtype=torch.float64, device=dev)
    hip._call("i2p_bn_stats", rows, cin, hip._p(x, torch.float32, "x"), hip._p(sx, torch.float64, "s"), stream=hip._stream())
    in_coef, in_mi = hip.bn_finalize(rows, sx, gam, bet, 1e-5)
    y, sy = hip.lin_forward(x, in_coef, 0.1, w)
    out_coef, out_mi = hip.bn_finalize(rows, sy, go, bo, 1e-5)
    gz = torch.randn(rows, cout, device=dev)
    ods = torch.zeros(ops.BN_REPLICAS * 2 * cout, dtype=torch.float64, device=dev)
    t = timeit(lambda: hip.lin_backward(gz, y, out_coef, out_mi, ods, x, in_coef, in_mi, 0.1, w), iters=40, warm=60)
    print(f"lin_bwd {cin}->{cout}: {t:.1f} us  {4.0 * rows * cin * cout / t / 1e6:.1f} TFLOP/s")
