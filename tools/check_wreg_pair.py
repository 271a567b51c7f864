"""Parity + timing of the pair-layer backward (first cost-volume layer) against fp64 torch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
from check_wreg import timeit


def reference(f, g, w, y, gz, out_coef, out_mi, rows, Co):
    om, oc = out_mi.view(-1).double(), out_coef.view(-1).double()
    xh = (y.double() - om[:Co]) * om[Co:]
    s1, s2 = gz.double().sum(0), (gz.double() * xh).sum(0)
    ods = torch.zeros(ops.BN_REPLICAS, 2, Co, dtype=torch.float64, device=gz.device); ods[0, 0] = s1; ods[0, 1] = s2
    gy = oc[Co:2 * Co] * (gz.double() - s1 / rows - xh * (s2 / rows))
    B, N, C = f.shape; M = g.shape[1]
    gy4 = gy.view(B, N, M, Co)
    P = f.double()[:, :, None, :] * g.double()[:, None, :, :]
    dw = gy.t() @ P.view(rows, C)
    dP = (gy @ w.double()).view(B, N, M, C)
    d_f = (dP * g.double()[:, None]).sum(2); d_g = (dP * f.double()[:, :, None]).sum(1)
    return ods.view(-1), d_f, d_g, gy4.sum(2), gy4.sum(1), dw


def main():
    hip = ops.hip_backend(); dev = "cuda"
    torch.manual_seed(0)
    C = Co = 128
    for B, N, M in [(2, 72, 468), (8, 228, 468)]:
        rows = B * N * M
        f = torch.randn(B, N, C, device=dev); g = torch.randn(B, M, C, device=dev)
        bn = torch.randn(B, N, Co, device=dev); bk = torch.randn(B, M, Co, device=dev)
        w = torch.randn(Co, C, device=dev) / C ** 0.5
        y, sy = hip.pair_lin_forward(f, g, bn, bk, w)
        out_coef, out_mi = hip.bn_finalize(rows, sy, torch.rand(Co, device=dev) + 0.5, torch.zeros(Co, device=dev), 1e-5)
        gz = torch.randn(rows, Co, device=dev)
        ods, *ref = reference(f, g, w, y, gz, out_coef, out_mi, rows, Co)
        got = hip.pair_lin_backward(gz, f, g, w, y=y, out_coef=out_coef, out_mi=out_mi, out_dsums=ods)
        errs = [float((a.double() - r).abs().max() / r.abs().max()) for a, r in zip(got, ref)]
        del ref
        t = timeit(lambda: hip.pair_lin_backward(gz, f, g, w, y=y, out_coef=out_coef, out_mi=out_mi, out_dsums=ods), iters=20, warm=60)
        print(f"B {B} N {N} M {M}: err d_f {errs[0]:.1e} d_g {errs[1]:.1e} d_bn {errs[2]:.1e} d_bk {errs[3]:.1e} dw {errs[4]:.1e}   {t:8.1f} us "
              f"{4.0 * rows * C * Co / t / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
