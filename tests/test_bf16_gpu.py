"""bf16 storage mode (BASELINE configs[2] / configs[4]): every bf16 kernel against a torch restatement of the same
arithmetic (fp32 / fp64 math on the bf16-rounded operands), and the whole network in bf16 mode against the fp32 golden
vectors of the reference.

Tolerances.  bf16 keeps 8 significant bits: one rounding is a relative error of at most 2^-9 ~ 2e-3 of the stored
value.  A kernel output that is itself rounded to bf16 is compared at 2^-7 relative per element (one ulp either way
where the fp32 accumulation order flips a rounding) plus a small absolute term; fp32 / fp64 outputs (statistics,
weight gradients) at 1e-3 relative to the tensor's scale.  The network-level tolerance is stated in its test."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
ULP = 2.0 ** -7


def _hip():
    from i2pnet_amd import ops
    return ops.hip_backend()


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _bfr(t):
    return t.to(BF).float()


def _close_bf16(got, want, what, rel=ULP, abs_frac=2e-3):
    got, want = got.float(), want.float()
    tol = rel * want.abs() + abs_frac * want.abs().max() * ULP + 1e-30
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - want).abs().max()), float(want.abs().max()))


def _coef(c, seed):
    """a plausible finalised BN: coef [3,c] = mean, scale, beta; mi [2c] = mean, invstd"""
    g = torch.Generator().manual_seed(seed)
    mean = torch.randn(c, generator=g) * 0.5
    invstd = 0.5 + torch.rand(c, generator=g)
    gamma = 1.0 + 0.2 * torch.randn(c, generator=g)
    beta = 0.2 * torch.randn(c, generator=g)
    coef = torch.stack([mean, invstd * gamma, beta]).contiguous().to(DEV)
    mi = torch.cat([mean, invstd]).contiguous().to(DEV)
    return coef, mi


def _fma(x, a, b):
    """fmaf(x, a, b) bit for bit: the fp32 product is exact in fp64, one rounding at the end"""
    return (x.double() * a.double() + b.double()).float()


def _bn_act(y, coef, slope):
    """the kernels' BN: z = fmaf(y, a, b), a = scale, b = beta - mean*a (two fp32 roundings)"""
    a = coef[1]; b = coef[2] - coef[0] * a
    z = _fma(y.float(), a, b)
    return torch.where(z > 0, z, z * slope), z


def _sums(t):
    from i2pnet_amd import ops
    return t.view(ops.BN_REPLICAS, 2, -1).sum(0)


@pytest.mark.parametrize("cin,cout,xbf,coef", [(12, 16, False, False), (64, 32, False, False), (136, 128, False, False),
                                               (16, 16, True, True), (16, 32, True, True), (32, 64, True, True),
                                               (64, 64, True, True), (128, 128, True, True), (128, 64, True, True),
                                               (64, 128, True, False), (64, 64, False, True)])
def test_lin_fwd_bf16(cin, cout, xbf, coef):
    hip = _hip()
    rows = 3 * 4096 + 77                        # not a multiple of the 32-row strips
    x = _rnd(rows, cin, seed=1)
    if xbf:
        x = x.to(BF)
    w = _rnd(cout, cin, seed=2, scale=cin ** -0.5)
    cf = _coef(cin, 3)[0] if coef else None
    y, sums = hip.lin_forward(x, cf, 0.1, w, out_dtype=BF)
    assert y.dtype == BF and y.shape == (rows, cout)
    a = _bn_act(x, cf, 0.1)[0] if coef else x.float()
    want = _bfr(a).double() @ _bfr(w).double().t()
    _close_bf16(y, _bfr(want.float()), "y")
    s = _sums(sums)
    yf = y.double()
    assert torch.allclose(s[0], yf.sum(0), rtol=1e-6, atol=1e-3) and torch.allclose(s[1], (yf * yf).sum(0), rtol=1e-6, atol=1e-3)


def test_lin_fwd_2src_bf16():
    hip = _hip()
    rows = 10000
    xa, xb = _rnd(rows, 64, seed=1).to(BF), _rnd(rows, 64, seed=2).to(BF)
    ca, cb = _coef(64, 3)[0], _coef(64, 4)[0]
    w = _rnd(128, 128, seed=5, scale=128 ** -0.5)
    y, sums = hip.lin_forward_2src(xa, ca, 0.1, xb, cb, 0.0, w)
    a = torch.cat([_bn_act(xa, ca, 0.1)[0], _bn_act(xb, cb, 0.0)[0]], 1)
    want = _bfr(a).double() @ _bfr(w).double().t()
    _close_bf16(y, _bfr(want.float()), "y")
    assert torch.allclose(_sums(sums)[0], y.double().sum(0), rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("C,Co", [(128, 128), (64, 32)])
def test_pair_lin_fwd_bf16(C, Co):
    hip = _hip()
    B, N, M = 2, 37, 150
    f, g = _rnd(B, N, C, seed=1), _rnd(B, M, C, seed=2)
    bn, bk = _rnd(B, N, Co, seed=3), _rnd(B, M, Co, seed=4)
    w = _rnd(Co, C, seed=5, scale=C ** -0.5)
    y, sums = hip.pair_lin_forward(f, g, bn, bk, w, out_dtype=BF)
    prod = _bfr(f.unsqueeze(2) * g.unsqueeze(1)).double()                          # [B,N,M,C]
    want = prod @ _bfr(w).double().t() + bn.double().unsqueeze(2) + bk.double().unsqueeze(1)
    _close_bf16(y.view(B, N, M, Co), _bfr(want.float()), "y")
    assert torch.allclose(_sums(sums)[0], y.double().sum(0), rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("B,N,M", [(2, 228, 468), (3, 190, 33), (1, 600, 75), (16, 228, 468), (1, 500, 40)])
def test_pair_fwd_shared_strip_matches_row_order_kernel(B, N, M, monkeypatch):
    """pair_fwd3_bf16_kernel (>= 16384 rows: the strip of 32 pixels shared by the block's four waves) against
    rg_fwd_kernel<4,false,PAIR>: same products, same accumulation order -> bit-identical bf16 output; statistics equal up to
    summation order and equal to the sums of the stored values.  Partial pixel tiles, ranges crossing several pixel tiles, M < 32."""
    hip = _hip()
    C = Co = 128
    assert B * N * M >= 16384
    f, g = _rnd(B, N, C, seed=11), _rnd(B, M, C, seed=12)
    bn, bk = _rnd(B, N, Co, seed=13), _rnd(B, M, Co, seed=14)
    w = _rnd(Co, C, seed=15, scale=C ** -0.5)
    y1, s1 = hip.pair_lin_forward(f, g, bn, bk, w, out_dtype=BF)
    monkeypatch.setenv("I2P_NO_PAIR_FWD3", "1")
    y0, s0 = hip.pair_lin_forward(f, g, bn, bk, w, out_dtype=BF)
    monkeypatch.delenv("I2P_NO_PAIR_FWD3")
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), float((y1.float() - y0.float()).abs().max())
    assert torch.allclose(_sums(s1), _sums(s0), rtol=1e-5, atol=1e-2)
    yd = y1.double()
    assert torch.allclose(_sums(s1)[0], yd.sum(0), rtol=1e-6, atol=1e-3) and torch.allclose(_sums(s1)[1], (yd * yd).sum(0), rtol=1e-6, atol=1e-3)


def _g_of(gz, y, coef, mi, dsums_rep, rows, slope_out):
    """BN backward on load as the kernels form it: A*gz' + B*y + C"""
    c = y.shape[1]
    s = _sums(dsums_rep)
    m1, m2 = (s[0] / rows).float(), (s[1] / rows).float()
    sc, mu, is_, be = coef[1], mi[:c], mi[c:], coef[2]
    z = _fma(y.float(), sc, be - mu * sc)
    gzp = torch.where(z > 0, gz.float(), gz.float() * slope_out) if slope_out != 1.0 else gz.float()
    A = sc; Bc = -(sc * m2) * is_; Cc = -(sc * m1) - Bc * mu
    return _fma(A, gzp, _fma(Bc, y.float(), Cc))


@pytest.mark.parametrize("cin,cout,xbf,in_bn,slope_out", [(128, 128, True, True, 1.0), (128, 64, True, True, 0.1),
                                                           (64, 64, True, True, 1.0), (16, 32, True, True, 1.0),
                                                           (64, 32, False, False, 1.0), (12, 16, False, False, 0.0),
                                                           (100, 128, False, False, 1.0)])
def test_lin_bwd_bf16(cin, cout, xbf, in_bn, slope_out):
    hip = _hip()
    rows = 2 * 4096 + 51
    x = _rnd(rows, cin, seed=1)
    x = x.to(BF) if xbf else x
    yv = _rnd(rows, cout, seed=2).to(BF)
    gz = _rnd(rows, cout, seed=3, scale=0.1).to(BF)
    w = _rnd(cout, cin, seed=4, scale=cin ** -0.5)
    oc, omi = _coef(cout, 5)
    ic, imi = _coef(cin, 6) if in_bn else (None, None)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, slope_out)
    # statistics kernel vs torch
    z = _bn_act(yv, oc, 1.0)[1]
    dz = torch.where(z > 0, gz.float(), gz.float() * slope_out)
    xh = (yv.float() - omi[:cout]) * omi[cout:]
    s = _sums(out_ds)
    assert torch.allclose(s[0], dz.double().sum(0), rtol=1e-5, atol=1e-2)
    assert torch.allclose(s[1], (dz.double() * xh.double()).sum(0), rtol=1e-4, atol=5e-2)
    need_gx = xbf or cin in (16, 32, 64, 128)
    gz_in, in_ds, dw = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w, need_gx=need_gx, slope_out=slope_out)
    dgam, dbet = hip.take_bn_grads()
    assert torch.allclose(dbet, s[0].float(), rtol=1e-5, atol=1e-3) and torch.allclose(dgam, s[1].float(), rtol=1e-5, atol=1e-3)
    G = _g_of(gz, yv, oc, omi, out_ds, rows, slope_out)
    xa = _bn_act(x, ic, 0.1)[0] if in_bn else x.float()
    want_dw = _bfr(G).double().t() @ _bfr(xa).double()
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max()), "dw"
    if need_gx:
        T = (_bfr(G).double() @ _bfr(w).double()).float()
        if in_bn:
            zi = _bn_act(x, ic, 1.0)[1]
            want = _bfr(torch.where(zi > 0, _bfr(T), _bfr(T) * 0.1))
            _close_bf16(gz_in, want, "gz_in", rel=2 * ULP)
            si = _sums(in_ds)
            gi = gz_in.double()
            xhi = ((x.float() - imi[:cin]) * imi[cin:]).double()
            assert torch.allclose(si[0], gi.sum(0), rtol=1e-5, atol=1e-2) and torch.allclose(si[1], (gi * xhi).sum(0), rtol=1e-4, atol=5e-2)
        else:
            assert gz_in.dtype == x.dtype
            ref = T if gz_in.dtype == torch.float32 else _bfr(T)
            assert float((gz_in.float() - ref).abs().max()) <= (1e-4 if gz_in.dtype == torch.float32 else ULP) * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("cin,rows", [(128, 65536 + 5 * 32 + 13), (64, 65536 + 32 + 31), (128, 65536), (128, 1707264), (64, 1707264)])
def test_lin_bwd_bf16_one_pass(cin, rows, monkeypatch):
    """bwd_fused_bf16_kernel (64-output-channel layers, >= 65536 rows: input gradient + weight gradient from one read of
    gz / y / x) against (a) fp64 torch on the bf16-rounded operands and (b) the two-kernel form (rg_dgrad + wreg_wgrad_bf16):
    same operands, same MFMA accumulation order for the input gradient -> bit-identical dL/dz_in.  Sizes: partial last strip,
    exact multiple, and the 1 707 264 rows of configs[2]'s all-pixel cost volume (16 x 228 x 468)."""
    hip = _hip()
    cout = 64
    x = _rnd(rows, cin, seed=1).to(BF)
    yv = _rnd(rows, cout, seed=2).to(BF)
    gz = _rnd(rows, cout, seed=3, scale=0.1).to(BF)
    w = _rnd(cout, cin, seed=4, scale=cin ** -0.5)
    oc, omi = _coef(cout, 5)
    ic, imi = _coef(cin, 6)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
    gz_in, in_ds, dw = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w)
    dgam, dbet = hip.take_bn_grads()
    monkeypatch.setenv("I2P_NO_FUSED_BF16", "1")
    gz_in0, in_ds0, dw0 = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w)
    monkeypatch.delenv("I2P_NO_FUSED_BF16")
    torch.cuda.synchronize()
    assert torch.equal(gz_in.view(torch.int16), gz_in0.view(torch.int16)), float((gz_in.float() - gz_in0.float()).abs().max())
    assert float((dw - dw0).abs().max()) <= 1e-4 * float(dw0.abs().max())
    s = _sums(out_ds)
    assert torch.allclose(dbet, s[0].float(), rtol=1e-5, atol=1e-3) and torch.allclose(dgam, s[1].float(), rtol=1e-5, atol=1e-3)
    # fp64 on the bf16-rounded operands
    G = _bfr(_g_of(gz, yv, oc, omi, out_ds, rows, 1.0))
    xa = _bfr(_bn_act(x, ic, 0.1)[0])
    want_dw = torch.zeros(cout, cin, dtype=torch.float64, device=DEV)
    step = 1 << 18
    for r0 in range(0, rows, step):                                            # (chunks: fp64 copies of 1.7 M x 128 add up)
        want_dw += G[r0:r0 + step].double().t() @ xa[r0:r0 + step].double()
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max()), "dw"
    si = _sums(in_ds)
    acc0 = torch.zeros(cin, dtype=torch.float64, device=DEV); acc1 = torch.zeros_like(acc0)
    wb = _bfr(w).double()
    for r0 in range(0, rows, step):
        sl = slice(r0, r0 + step)
        T = (G[sl].double() @ wb).float()
        zi = _bn_act(x[sl], ic, 1.0)[1]
        want = _bfr(torch.where(zi > 0, _bfr(T), _bfr(T) * 0.1))
        _close_bf16(gz_in[sl], want, "gz_in", rel=2 * ULP)
        gi = gz_in[sl].double()
        acc0 += gi.sum(0); acc1 += (gi * ((x[sl].float() - imi[:cin]) * imi[cin:]).double()).sum(0)
    scale = lambda t: float(t.abs().max())
    assert float((si[0] - acc0).abs().max()) <= 1e-5 * float(gz_in.float().abs().sum(0).max()) + 1e-2
    assert float((si[1] - acc1).abs().max()) <= 1e-5 * float(gz_in.float().abs().sum(0).max()) * 4 + 5e-2
    assert torch.allclose(_sums(in_ds0), si, rtol=1e-4, atol=5e-2)


def test_lin_bwd_2src_bf16():
    hip = _hip()
    rows, ca, cb, cout = 9000, 64, 64, 128
    xa, xb = _rnd(rows, ca, seed=1).to(BF), _rnd(rows, cb, seed=2).to(BF)
    yv, gz = _rnd(rows, cout, seed=3).to(BF), _rnd(rows, cout, seed=4, scale=0.1).to(BF)
    eadd = _rnd(rows, cb, seed=5, scale=0.1).to(BF)
    w = _rnd(cout, ca + cb, seed=6, scale=128 ** -0.5)
    oc, omi = _coef(cout, 7); cfa, mia = _coef(ca, 8); cfb, mib = _coef(cb, 9)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
    gza, dsa, gzb, dsb, dw = hip.lin_backward_2src(gz, yv, oc, omi, out_ds, xa, cfa, mia, 0.1, xb, cfb, mib, 0.1, eadd, w)
    G = _bfr(_g_of(gz, yv, oc, omi, out_ds, rows, 1.0))
    X = torch.cat([_bn_act(xa, cfa, 0.1)[0], _bn_act(xb, cfb, 0.1)[0]], 1)
    want_dw = G.double().t() @ _bfr(X).double()
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max())
    T = (G.double() @ _bfr(w).double()).float()
    za, zb = _bn_act(xa, cfa, 1.0)[1], _bn_act(xb, cfb, 1.0)[1]
    Ta = _bfr(T[:, :ca]); Tb = _bfr(T[:, ca:]) + eadd.float()
    _close_bf16(gza, _bfr(torch.where(za > 0, Ta, Ta * 0.1)), "gz_a", rel=2 * ULP)
    _close_bf16(gzb, _bfr(torch.where(zb > 0, Tb, Tb * 0.1)), "gz_b", rel=2 * ULP)
    assert torch.allclose(_sums(dsa)[0], gza.double().sum(0), rtol=1e-5, atol=1e-2)
    assert torch.allclose(_sums(dsb)[0], gzb.double().sum(0), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("rows", [65536 + 7 * 16, 1707264])
def test_lin_bwd_2src_bf16_one_pass(rows, monkeypatch):
    """bwd_fused2_bf16_kernel (64 + 64 -> 128, >= 65536 rows: both input gradients and the weight gradient from one read of
    gz / y / xa / xb / e_add) against the two-kernel form (bit-identical input gradients: same operands, same accumulation order
    over k) and against fp64 on the bf16-rounded operands; second size = configs[2]'s 16 x 228 x 468 rows."""
    hip = _hip()
    ca = cb = 64; cout = 128
    xa, xb = _rnd(rows, ca, seed=1).to(BF), _rnd(rows, cb, seed=2).to(BF)
    yv, gz = _rnd(rows, cout, seed=3).to(BF), _rnd(rows, cout, seed=4, scale=0.1).to(BF)
    eadd = _rnd(rows, cb, seed=5, scale=0.1).to(BF)
    w = _rnd(cout, ca + cb, seed=6, scale=128 ** -0.5)
    oc, omi = _coef(cout, 7); cfa, mia = _coef(ca, 8); cfb, mib = _coef(cb, 9)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
    gza, dsa, gzb, dsb, dw = hip.lin_backward_2src(gz, yv, oc, omi, out_ds, xa, cfa, mia, 0.1, xb, cfb, mib, 0.2, eadd, w)
    monkeypatch.setenv("I2P_NO_FUSED_BF16", "1")
    gza0, dsa0, gzb0, dsb0, dw0 = hip.lin_backward_2src(gz, yv, oc, omi, out_ds, xa, cfa, mia, 0.1, xb, cfb, mib, 0.2, eadd, w)
    monkeypatch.delenv("I2P_NO_FUSED_BF16")
    torch.cuda.synchronize()
    assert torch.equal(gza.view(torch.int16), gza0.view(torch.int16)), float((gza.float() - gza0.float()).abs().max())
    assert torch.equal(gzb.view(torch.int16), gzb0.view(torch.int16)), float((gzb.float() - gzb0.float()).abs().max())
    assert float((dw - dw0).abs().max()) <= 1e-4 * float(dw0.abs().max())
    assert torch.allclose(_sums(dsa), _sums(dsa0), rtol=1e-4, atol=5e-2) and torch.allclose(_sums(dsb), _sums(dsb0), rtol=1e-4, atol=5e-2)
    G = _bfr(_g_of(gz, yv, oc, omi, out_ds, rows, 1.0))
    wb = _bfr(w).double()
    want_dw = torch.zeros(cout, ca + cb, dtype=torch.float64, device=DEV)
    step = 1 << 18
    for r0 in range(0, rows, step):
        sl = slice(r0, r0 + step)
        X = _bfr(torch.cat([_bn_act(xa[sl], cfa, 0.1)[0], _bn_act(xb[sl], cfb, 0.2)[0]], 1))
        want_dw += G[sl].double().t() @ X.double()
        T = (G[sl].double() @ wb).float()
        za, zb = _bn_act(xa[sl], cfa, 1.0)[1], _bn_act(xb[sl], cfb, 1.0)[1]
        Ta = _bfr(T[:, :ca]); Tb = _bfr(T[:, ca:]) + eadd[sl].float()
        _close_bf16(gza[sl], _bfr(torch.where(za > 0, Ta, Ta * 0.1)), "gz_a", rel=2 * ULP)
        # (source b adds e_add to the rounded product: where the two cancel, one bf16 ulp of the PRODUCT is the error scale)
        want_b = _bfr(torch.where(zb > 0, Tb, Tb * 0.2))
        mag = (_bfr(T[:, ca:]).abs() + eadd[sl].float().abs()) * torch.where(zb > 0, 1.0, 0.2)
        bad = (gzb[sl].float() - want_b).abs() > 2 * ULP * mag + 1e-30
        assert not bool(bad.any()), ("gz_b", int(bad.sum()))
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max())
    assert torch.allclose(_sums(dsa)[0], gza.double().sum(0), rtol=1e-5, atol=1e-1)
    assert torch.allclose(_sums(dsb)[0], gzb.double().sum(0), rtol=1e-5, atol=1e-1)


@pytest.mark.parametrize("B,N,M,C,Co", [(2, 13, 150, 128, 128), (1, 40, 64, 64, 32), (2, 9, 468, 128, 128)])
def test_pair_lin_bwd_bf16(B, N, M, C, Co):
    hip = _hip()
    rows = B * N * M
    f, g = _rnd(B, N, C, seed=1), _rnd(B, M, C, seed=2)
    w = _rnd(Co, C, seed=3, scale=C ** -0.5)
    yv, gz = _rnd(rows, Co, seed=4).to(BF), _rnd(rows, Co, seed=5, scale=0.1).to(BF)
    oc, omi = _coef(Co, 6)
    ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
    d_f, d_g, d_bn, d_bk, dw = hip.pair_lin_backward(gz, f, g, w, y=yv, out_coef=oc, out_mi=omi, out_dsums=ds)
    G = _bfr(_g_of(gz, yv, oc, omi, ds, rows, 1.0)).double().view(B, N, M, Co)
    fd, gd = f.double(), g.double()
    T = G @ _bfr(w).double()                                                           # [B,N,M,C]
    scale = lambda t: float(t.abs().max())
    want_bn, want_bk = G.sum(2), G.sum(1)
    assert float((d_bn.double() - want_bn).abs().max()) <= 1e-4 * scale(want_bn) + 1e-6
    assert float((d_bk.double() - want_bk).abs().max()) <= 1e-4 * scale(want_bk) + 1e-6
    want_df = (T * gd.unsqueeze(1)).sum(2); want_dg = (T * fd.unsqueeze(2)).sum(1)
    assert float((d_f.double() - want_df).abs().max()) <= 1e-3 * scale(want_df)
    assert float((d_g.double() - want_dg).abs().max()) <= 1e-3 * scale(want_dg)
    # dW[co,ci] = sum f[n,ci] * sum_px G[px,co] * bf16(g[px,ci])
    want_dw = torch.einsum("bnmo,bmc,bnc->oc", G, _bfr(g).double(), fd)
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * scale(want_dw)


@pytest.mark.parametrize("B,N,M", [(2, 40, 468), (1, 300, 75), (3, 50, 200), (5, 8, 468), (1, 2500, 8), (16, 228, 468)])
def test_pair_lin_bwd_bf16_second_generation(B, N, M, monkeypatch):
    """pair_bwd2_bf16_kernel (128 x 128, >= 16384 rows: persistent blocks over 32-pixel strips) against fp64 on the bf16-rounded
    operands and against the first-generation kernel.  Shapes: partial pixel tiles (M % 32 = 20, 11, 8), ranges that cross
    several (sample, pixel tile) boundaries, fewer strips than blocks, and configs[2]'s own 16 x 228 x 468."""
    hip = _hip()
    C = Co = 128
    rows = B * N * M
    f, g = _rnd(B, N, C, seed=1), _rnd(B, M, C, seed=2)
    w = _rnd(Co, C, seed=3, scale=C ** -0.5)
    yv, gz = _rnd(rows, Co, seed=4).to(BF), _rnd(rows, Co, seed=5, scale=0.1).to(BF)
    oc, omi = _coef(Co, 6)
    ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
    d_f, d_g, d_bn, d_bk, dw = hip.pair_lin_backward(gz, f, g, w, y=yv, out_coef=oc, out_mi=omi, out_dsums=ds)
    monkeypatch.setenv("I2P_NO_PAIR_BWD2", "1")
    r_f, r_g, r_bn, r_bk, r_dw = hip.pair_lin_backward(gz, f, g, w, y=yv, out_coef=oc, out_mi=omi, out_dsums=ds)
    monkeypatch.delenv("I2P_NO_PAIR_BWD2")
    torch.cuda.synchronize()
    scale = lambda t: float(t.abs().max())
    for a, b, what, tol in ((d_f, r_f, "d_f", 1e-3), (d_g, r_g, "d_g", 1e-3), (d_bn, r_bn, "d_bn", 1e-4), (d_bk, r_bk, "d_bk", 1e-4), (dw, r_dw, "dw", 8e-3)):   # (dW: the two kernels round different operands, bf16(f g) vs f bf16(g))
        assert float((a - b).abs().max()) <= tol * scale(b) + 1e-6, what
    # fp64 on the bf16-rounded operands, sample by sample (the [N, M, 128] products of one sample fit easily)
    Gall = _bfr(_g_of(gz, yv, oc, omi, ds, rows, 1.0)).view(B, N, M, Co)
    wb = _bfr(w).double()
    want_dw = torch.zeros(Co, C, dtype=torch.float64, device=DEV)
    for b in range(B):
        G = Gall[b].double()
        fd, gd = f[b].double(), g[b].double()
        T = G @ wb                                                                     # [N,M,C]
        assert float((d_bn[b].double() - G.sum(1)).abs().max()) <= 1e-4 * scale(G.sum(1)) + 1e-6
        assert float((d_bk[b].double() - G.sum(0)).abs().max()) <= 1e-4 * scale(G.sum(0)) + 1e-6
        want_df = (T * gd.unsqueeze(0)).sum(1); want_dg = (T * fd.unsqueeze(1)).sum(0)
        assert float((d_f[b].double() - want_df).abs().max()) <= 1e-3 * scale(want_df)
        assert float((d_g[b].double() - want_dg).abs().max()) <= 1e-3 * scale(want_dg)
        # dW = G^T . bf16(f (.) g): the operand of the forward kernel
        X = _bfr(f[b].unsqueeze(1) * g[b].unsqueeze(0)).double()                        # [N,M,C]
        want_dw += torch.einsum("nmo,nmc->oc", G, X)
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * scale(want_dw)


def test_stream_kernels_bf16():
    from i2pnet_amd import ops
    hip = _hip()
    # outer sum
    B, N, M, C = 2, 30, 100, 64
    en, ek = _rnd(B, N, C, seed=1), _rnd(B, M, C, seed=2)
    ye, sums = hip.outer_sum_bf16(en, ek)
    want = (en.unsqueeze(2) + ek.unsqueeze(1)).to(BF).view(-1, C)
    assert torch.equal(ye, want)
    assert torch.allclose(_sums(sums)[0], ye.double().sum(0), rtol=1e-6, atol=1e-3)
    assert torch.allclose(_sums(sums)[1], (ye.double() ** 2).sum(0), rtol=1e-6, atol=1e-3)
    # conversion
    t = _rnd(1000, 24, seed=3)
    assert torch.equal(hip.to_bf16(t), t.to(BF))
    # chain output / max over K / unpool
    rows, c, K = 4096 * 3, 32, 16
    y = _rnd(rows, c, seed=4).to(BF)
    cf, mi = _coef(c, 5)
    out = hip.bn_act_apply_bf16(y, cf, 0.1)
    assert torch.allclose(out, _bn_act(y, cf, 0.1)[0], rtol=1e-6, atol=1e-6)
    mx, arg = hip.bn_act_maxk_forward(y, cf, 0.0, K)
    wa = _bn_act(y, cf, 0.0)[0].view(rows // K, K, c)
    wm, wi = wa.max(1)
    assert torch.allclose(mx, wm, rtol=1e-6, atol=1e-6)
    assert torch.equal(torch.gather(wa, 1, arg.long().unsqueeze(1)).squeeze(1), wm)   # an arg-max (first one on ties)
    g = _rnd(rows // K, c, seed=6)
    gd = hip.unpool_k(g, arg, K, dtype=BF)
    dense = torch.zeros(rows // K, K, c, device=DEV).scatter_(1, arg.long().unsqueeze(1), g.unsqueeze(1))
    assert torch.equal(gd.view(rows // K, K, c), dense.to(BF))


@pytest.mark.parametrize("B,N,M,C", [(2, 20, 468, 64), (1, 7, 32, 64), (2, 5, 100, 128)])
def test_cv_softmax_bf16_matches_fp32_kernels(B, N, M, C):
    hip = _hip()
    rows = B * N * M
    y5, y3 = _rnd(rows, C, seed=1).to(BF), _rnd(rows, C, seed=2).to(BF)
    c5, m5 = _coef(C, 3); c3, _ = _coef(C, 4)
    out, msave = hip.cv_softmax_wsum_forward(B, N, M, y5, c5, 0.1, y3, c3, 0.1)
    # the fp32 kernels on the same (upcast) tensors with coefficients in the bf16 kernels' a/b form are the reference
    out32, ms32 = hip.cv_softmax_wsum_forward(B, N, M, y5.float(), c5, 0.1, y3.float(), c3, 0.1)
    assert torch.allclose(out, out32, rtol=2e-4, atol=2e-5)
    go = _rnd(B, N, C, seed=5)
    gz5, ds5, ga3 = hip.cv_softmax_wsum_backward(B, N, M, go, out, msave, y5, c5, m5, 0.1, y3, c3, 0.1)
    gz5r, ds5r, ga3r = hip.cv_softmax_wsum_backward(B, N, M, go, out32, ms32, y5.float(), c5, m5, 0.1, y3.float(), c3, 0.1)
    assert gz5.dtype == BF and ga3.dtype == BF
    _close_bf16(gz5, _bfr(gz5r), "gz5", rel=2 * ULP, abs_frac=5e-2)
    _close_bf16(ga3, _bfr(ga3r), "ga3", rel=2 * ULP, abs_frac=5e-2)
    assert torch.allclose(_sums(ds5)[0], gz5.double().sum(0), rtol=1e-5, atol=1e-4)
    # factors of the position encoding
    en, ek = _rnd(B, N, C, seed=6), _rnd(B, M, C, seed=7)
    gze = _rnd(rows, C, seed=8, scale=0.1).to(BF)
    ce, me = _coef(C, 9)
    ye, _ = hip.outer_sum_bf16(en, ek)
    dse = hip.bn_act_backward_stats_bf16(gze, ye, ce, me, 1.0)
    dn, dk = hip.pair_bias_bn_backward(B, N, M, gze, en, ek, dse, ce, me)
    dn32, dk32 = hip.pair_bias_bn_backward(B, N, M, gze.float(), en, ek, dse, ce, me)
    assert torch.allclose(dn, dn32, rtol=1e-4, atol=1e-4 * float(dn32.abs().max()))
    assert torch.allclose(dk, dk32, rtol=1e-4, atol=1e-4 * float(dk32.abs().max()))


def _model_run(tag, precision, min_rows=0):
    import test_model_golden as G
    from i2pnet_amd import ops
    prev_p, prev_r = ops.set_precision(precision), ops.BF16_MIN_ROWS
    ops.BF16_MIN_ROWS = min_rows
    try:
        return G._run(tag, DEV)
    finally:
        ops.set_precision(prev_p); ops.BF16_MIN_ROWS = prev_r


@pytest.mark.parametrize("tag", ["kitti", "nus"])
def test_model_bf16_against_fp32_golden(tag):
    """The whole network with bf16 chain storage (every eligible chain: row threshold 0) against the REFERENCE's fp32
    golden vectors.  Integer outputs (neighbour indices, masks) do not depend on the storage mode (they are computed on
    fp32 coordinates before any bf16 tensor exists).  Tolerance: a bf16 store is a 2^-9 relative perturbation per
    element; behind ~25 batch-normalised layers (each renormalises to unit variance, so perturbations add in
    quadrature, ~ sqrt(25) * 2^-9 ~ 1e-2 of a unit-variance activation: measured 0.5 % (level 1) to 2.6 % (mask
    predictors) in L2, tools/diag_bf16.py) the regressed pose is required within 8e-2 of its fp32 value relative to the
    pose's scale (measured 2-5e-2), the loss within 5e-2 (measured 4e-3), every recorded activation within 1e-1 in
    relative L2 norm (and 3e-1 in max-norm).  Gradients: a 1 % forward perturbation flips ReLU / max-pool / softmax-mask decisions of this
    random-weight network, so individual gradient tensors move by 10-20 % in L2 (and two fp32 runs of the same
    gradient differ by more than that on the ill-conditioned tensors): only the norm of the
    whole well-conditioned gradient is checked (25 %), plus tests/test_bf16_gpu.py::test_bf16_training_tracks_fp32
    for the statement that matters — training in bf16 mode follows the fp32 loss curve."""
    gold, model, acts, out3, out4, loss = _model_run(tag, "bf16")
    rel = lambda a, b: float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double()).abs().max() / (torch.as_tensor(b).double().abs().max() + 1e-12))
    assert rel(out3.detach(), gold["out3"]) < 8e-2, rel(out3.detach(), gold["out3"])
    assert rel(out4.detach(), gold["out4"]) < 8e-2, rel(out4.detach(), gold["out4"])
    assert abs(loss.item() - gold["loss"][0]) / abs(gold["loss"][0]) < 5e-2
    for name, t in acts.items():
        got, want = t.detach().reshape(-1, t.shape[-1]).double().cpu(), torch.as_tensor(gold["act." + name]).double()
        r2 = float((got - want).norm() / want.norm())
        assert r2 < 1e-1, (name, r2)                      # relative L2 (measured 0.5-2.6e-2 nuScenes shapes, up to 6e-2 KITTI)
        assert rel(got, want) < 3e-1, (name, rel(got, want))
    params = dict(model.named_parameters())
    # norm of every well-conditioned parameter gradient against the reference's fp64 evaluation
    g64 = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm64"].tolist()))
    gn = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm"].tolist()))
    floor = {}
    for k in params:
        if gn[k] > 1e-4 and g64[k] > 0.0:
            m = k.split(".")[0]
            floor[m] = max(floor.get(m, 0.0), abs(gn[k] - g64[k]) / g64[k])
    # Individual tensors move by tens of percent under a 1 % forward perturbation of this random-weight, batch-2
    # network (decision flips), so the statement is global: the norm of the whole well-conditioned gradient.
    num = den = 0.0
    for k, p in params.items():
        if gn[k] <= 1e-4 or g64[k] == 0.0 or floor[k.split(".")[0]] > 1e-3 or p.grad is None:
            continue
        num += float(p.grad.double().norm()) ** 2; den += g64[k] ** 2
    assert abs((num / den) ** 0.5 - 1.0) < 0.25, (num / den) ** 0.5


def test_bf16_mode_is_actually_bf16():
    """the storage mode reaches the kernels: a chain run in bf16 mode saves bf16 tensors"""
    from i2pnet_amd import fused, ops
    prev = ops.set_precision("bf16")
    try:
        x = _rnd(ops.BF16_MIN_ROWS, 12, seed=1)
        assert fused.chain_bf16_ok(x, False, [torch.zeros(16, 12)])
        assert not fused.chain_bf16_ok(x[:100], False, [torch.zeros(16, 12)])
    finally:
        ops.set_precision(prev)
    assert not fused.chain_bf16_ok(x, False, [torch.zeros(16, 12)])


def test_bf16_training_tracks_fp32():
    """Matched loss (BASELINE north star): 40 optimisation steps on one fixed batch in fp32 and in bf16 storage mode
    from the same initial weights: both must reduce the loss, and the bf16 curve must stay within 10 % of the fp32
    curve's final value (dropout off: its mask would otherwise differ between the two runs' RNG consumption).
    Two fp32 runs from the same seed do not retrace each other either (MIOpen's split-K convolution weight gradients
    accumulate with atomics; 40 Adam steps amplify the last bits: tails 4-6 % apart, tools/diag_bf16_curve.py), so the
    fp32 curve is run twice and its own run-to-run spread is added to the allowance."""
    from i2pnet_amd import ops, synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.train import Trainer
    dev = torch.device("cuda", 0)
    batch = synth.make_batch(4, 8192, 160, 512, seed=3, device=dev)

    def curve(prec):
        prev = ops.set_precision(prec); prev_r = ops.BF16_MIN_ROWS
        ops.BF16_MIN_ROWS = 4096
        try:
            tr = Trainer(cfg=cfg, device=dev, seed=0)
            tr.net.l3_head.DP1.p = 0.0; tr.net.l4_head.DP1.p = 0.0
            return [float(tr.step(batch)[0]) for _ in range(40)]
        finally:
            ops.set_precision(prev); ops.BF16_MIN_ROWS = prev_r
    c32, c32b, c16 = curve("fp32"), curve("fp32"), curve("bf16")
    assert all(math.isfinite(v) for v in c16)
    assert c32[-1] < 0.8 * c32[0] and c16[-1] < 0.8 * c16[0], (c32[0], c32[-1], c16[0], c16[-1])
    tail = lambda c: sum(c[-5:]) / 5
    tail32, spread, tail16 = (tail(c32) + tail(c32b)) / 2, abs(tail(c32) - tail(c32b)), tail(c16)
    assert abs(tail16 - tail32) <= 0.10 * abs(tail32) + 2 * spread, (tail(c32), tail(c32b), tail16)


@pytest.mark.parametrize("cin,cout,slope_out", [(128, 128, 1.0), (128, 64, 0.1), (64, 128, 1.0), (64, 64, 1.0), (32, 32, 1.0),
                                                (32, 64, 0.1), (64, 32, 1.0), (128, 32, 1.0), (32, 128, 1.0)])
def test_wgrad_bf16_in_registers(cin, cout, slope_out):
    """wgrad of the wide bf16 layers on >= 65536 rows (csrc/mlp_wreg_bf16.hip: accumulators in registers, row pairs packed
    into the MFMA operands without LDS) against the same fp64 evaluation of the bf16-rounded operands as test_lin_bwd_bf16"""
    hip = _hip()
    rows = 65536 + 16 * 7
    x = _rnd(rows, cin, seed=11).to(BF)
    yv = _rnd(rows, cout, seed=12).to(BF)
    gz = _rnd(rows, cout, seed=13, scale=0.1).to(BF)
    w = _rnd(cout, cin, seed=14, scale=cin ** -0.5)
    oc, omi = _coef(cout, 15)
    ic, imi = _coef(cin, 16)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, slope_out)
    _, _, dw = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w, need_gx=False, slope_out=slope_out)
    G = _g_of(gz, yv, oc, omi, out_ds, rows, slope_out)
    xa = _bn_act(x, ic, 0.1)[0]
    want_dw = _bfr(G).double().t() @ _bfr(xa).double()
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max())
    _, _, dw2 = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w, need_gx=False, slope_out=slope_out)
    assert torch.equal(dw, dw2)



def test_wgrad_bf16_in_registers_two_sources():
    """the 64 + 64 -> 128 layer (input = two tensors with their own BN and slopes) on the in-register wgrad: dW columns
    [0, 64) belong to the first source, [64, 128) to the second"""
    hip = _hip()
    rows, cout = 65536 + 16 * 5, 128
    xa = _rnd(rows, 64, seed=21).to(BF); xb = _rnd(rows, 64, seed=22).to(BF)
    yv = _rnd(rows, cout, seed=23).to(BF)
    gz = _rnd(rows, cout, seed=24, scale=0.1).to(BF)
    w = _rnd(cout, 128, seed=25, scale=128 ** -0.5)
    oc, omi = _coef(cout, 26)
    ca, mia = _coef(64, 27); cb, mib = _coef(64, 28)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, 1.0)
    dw = hip.lin_backward_2src(gz, yv, oc, omi, out_ds, xa, ca, mia, 0.1, xb, cb, mib, 1.0, None, w)[4]
    G = _g_of(gz, yv, oc, omi, out_ds, rows, 1.0)
    act = torch.cat([_bn_act(xa, ca, 0.1)[0], _bn_act(xb, cb, 1.0)[0]], 1)
    want = _bfr(G).double().t() @ _bfr(act).double()
    assert float((dw.double() - want).abs().max()) <= 2e-3 * float(want.abs().max())
    dw2 = hip.lin_backward_2src(gz, yv, oc, omi, out_ds, xa, ca, mia, 0.1, xb, cb, mib, 1.0, None, w)[4]
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("cin,cout,xbf,slope_out", [(16, 16, True, 1.0), (16, 32, True, 0.1), (12, 16, False, 0.0), (16, 32, False, 1.0)])
def test_small_wgrad_bf16_streaming(cin, cout, xbf, slope_out):
    """wgrad of the narrow level-1 layers on >= 262144 rows (csrc/mlp_wreg_bf16.hip small_wgrad_bf16_kernel: one element per
    lane and k-step, fp32 MFMA on bf16-rounded operands) against the fp64 product of the bf16-rounded operands"""
    hip = _hip()
    rows = 262144 + 16 * 9
    x = _rnd(rows, cin, seed=31)
    x = x.to(BF) if xbf else x
    yv = _rnd(rows, cout, seed=32).to(BF)
    gz = _rnd(rows, cout, seed=33, scale=0.1).to(BF)
    w = _rnd(cout, cin, seed=34, scale=cin ** -0.5)
    oc, omi = _coef(cout, 35)
    ic, imi = _coef(cin, 36) if xbf else (None, None)
    out_ds = hip.bn_act_backward_stats_bf16(gz, yv, oc, omi, slope_out)
    _, _, dw = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w, need_gx=False, slope_out=slope_out)
    G = _g_of(gz, yv, oc, omi, out_ds, rows, slope_out)
    xa = _bn_act(x, ic, 0.1)[0] if xbf else x.float()
    want_dw = _bfr(G).double().t() @ _bfr(xa).double()
    assert float((dw.double() - want_dw).abs().max()) <= 2e-3 * float(want_dw.abs().max())
    _, _, dw2 = hip.lin_backward(gz, yv, oc, omi, out_ds, x, ic, imi, 0.1, w, need_gx=False, slope_out=slope_out)
    assert torch.equal(dw, dw2)
