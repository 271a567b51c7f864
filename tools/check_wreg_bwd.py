"""Parity + timing of the layer backward with the weights-in-registers dgrad (csrc/mlp_wreg.hip) against fp64 torch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
from check_wreg import timeit


def main():
    hip = ops.hip_backend(); dev = "cuda"
    torch.manual_seed(0)
    R = ops.BN_REPLICAS
    for rows, cin, cout in [(853632, 128, 128), (853632, 128, 64), (853632, 64, 64), (853632, 64, 128), (65536 + 16 * 37, 128, 128)]:
        x = torch.randn(rows, cin, device=dev) * 1.5 + 0.2; w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        gam_i = torch.rand(cin, device=dev) + 0.5; bet_i = torch.randn(cin, device=dev) * 0.1
        gam_o = torch.rand(cout, device=dev) + 0.5; bet_o = torch.randn(cout, device=dev) * 0.1
        in_coef, in_mi = hip.bn_finalize(rows, hip.bn_stats(x), gam_i, bet_i, 1e-5)
        y, sy = hip.lin_forward(x, in_coef, 0.1, w)
        out_coef, out_mi = hip.bn_finalize(rows, sy, gam_o, bet_o, 1e-5)
        gz = torch.randn(rows, cout, device=dev)
        # fp64 reference
        yd, gzd, xd = y.double(), gz.double(), x.double()
        oc, om, ic, im = out_coef.view(-1), out_mi.view(-1), in_coef.view(-1), in_mi.view(-1)
        mu_o, is_o, sc_o = om[:cout].double(), om[cout:].double(), oc[cout:2 * cout].double()
        xh_o = (yd - mu_o) * is_o
        s1, s2 = gzd.sum(0), (gzd * xh_o).sum(0)
        ods = torch.zeros(R, 2, cout, dtype=torch.float64, device=dev); ods[0, 0] = s1; ods[0, 1] = s2
        gy = sc_o * (gzd - s1 / rows - xh_o * (s2 / rows))
        mu_i, is_i, sc_i, be_i = im[:cin].double(), im[cin:].double(), ic[cin:2 * cin].double(), ic[2 * cin:].double()
        z_i = (xd - mu_i) * sc_i + be_i
        ref_gin = (gy @ w.double()) * torch.where(z_i > 0, 1.0, 0.1)
        a_i = torch.where(z_i > 0, z_i, 0.1 * z_i)
        ref_dw = gy.t() @ a_i
        xh_i = (xd - mu_i) * is_i
        ref_ds = torch.stack([ref_gin.sum(0), (ref_gin * xh_i).sum(0)])
        del yd, gzd, xd, xh_o, gy, a_i, xh_i
        gin, ids, dw = hip.lin_backward(gz, y, out_coef, out_mi, ods.view(-1), x, in_coef, in_mi, 0.1, w)
        dif = (gin.double() - ref_gin).abs() / ref_gin.abs().max()
        bad = dif > 1e-4                                         # act' flips where |z_in| is at rounding level are legitimate
        nbad, zbad = int(bad.sum()), (float(z_i[bad].abs().max()) if bad.any() else 0.0)
        e_g = dif[~bad].max().item()
        e_w = ((dw.double() - ref_dw).abs().max() / ref_dw.abs().max()).item()
        got = ids.view(R, 2, cin).sum(0)
        e_s = ((got - ref_ds).abs().max() / ref_ds.abs().max()).item()
        del dif, bad, z_i, ref_gin
        t = timeit(lambda: hip.lin_backward(gz, y, out_coef, out_mi, ods.view(-1), x, in_coef, in_mi, 0.1, w), iters=30, warm=100)
        tw = timeit(lambda: hip.lin_backward(gz, y, out_coef, out_mi, ods.view(-1), x, in_coef, in_mi, 0.1, w, need_gx=False), iters=30, warm=30)
        by = rows * 4 * (2 * cout + 2 * cin)
        print(f"rows {rows} {cin}->{cout}: gz_in err {e_g:.2e} ({nbad} sign flips, max |z| there {zbad:.1e}) dsums err {e_s:.2e} dw err {e_w:.2e}   bwd {t:7.1f} us  wgrad {tw:7.1f}  dgrad {t - tw:7.1f} us = {by / (t - tw) / 1e3:5.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
