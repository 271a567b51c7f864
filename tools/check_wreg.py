"""Parity + timing of the weights-in-registers layer forward (csrc/mlp_wreg.hip) against an fp64 torch evaluation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops


def timeit(fn, iters=50, warm=200):      # (long warm-up: the first ~100 ms after idle run at ramping clocks)
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    hip = ops.hip_backend(); dev = "cuda"
    torch.manual_seed(0)
    for rows, cin, cout, bn in [(853632, 128, 128, False), (853632, 128, 128, True), (853632, 128, 128, False), (853632, 128, 64, True), (853632, 64, 128, True), (853632, 64, 64, True), (66128, 128, 128, True)]:

        x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        coef = torch.stack([torch.randn(cin) * 0.1, torch.rand(cin) + 0.5, torch.randn(cin) * 0.1]).to(dev).contiguous() if bn else None
        y, sums = hip.lin_forward(x, coef, 0.1, w)
        xd = x.double()
        if bn:
            c = coef.double()
            z = (xd - c[0]) * c[1] + c[2]
            xd = torch.where(z > 0, z, z * 0.1)
        ref = xd @ w.double().t()
        err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        s = sums.view(-1, 2, cout).sum(0)
        e1 = ((s[0] - ref.sum(0)).abs().max() / ref.abs().sum(0).max()).item()
        e2 = ((s[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).max()).item()
        t = timeit(lambda: hip.lin_forward(x, coef, 0.1, w))
        fl = 2.0 * rows * cin * cout
        print(f"rows {rows} {cin}->{cout} bn={bn}: y err {err:.2e} sum err {e1:.2e} sq err {e2:.2e}   {t:8.1f} us  {fl / t / 1e6:6.1f} TF  {rows*(cin+cout)*4/t/1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
