"""CPU tests: the oracle against the only known answers the reference holds for this path,
and against independent (slow, pure-torch) restatements of the operator definitions."""
import numpy as np
import pytest
import torch

from helpers import cloud, range_image, run_fcsk, stride_grid
from oracle import oracle

FLAG_COPY, FLAG_SHIFT = 1, 2


def test_reference_smoke_case_known_answer(oracle_backend):
    """src/projectPN/fused_conv_select/fused_conv_select_k.py:29-139 with random_hw=arange(3):
    H=4,W=9, small 4x5, 1x3 window, K=5, FLAG_SHIFT, stride (1,2), distance 200.
    Hand-derived: query (0,2) -> w=[0,1,2,0,0]; query (0,0) -> w=[4,0,1,0,0] (wrap), h=0,
    mask=[1,1,1,0,0] for both."""
    H, W, SH, SW, K = 4, 9, 4, 5, 5
    p1 = torch.ones(1, H, W, 3)
    base = np.concatenate([np.arange(1, SH * (SW - 1) + 1).reshape(SH, SW - 1), np.ones((SH, 1))], 1)
    p2 = torch.from_numpy(np.tile(base.reshape(1, SH, SW, 1).astype("float32"), [1, 1, 1, 3])).contiguous()
    idx = torch.tensor([[[0, 2], [0, 0]]], dtype=torch.int32)
    sb, sh, sw, m, v1, v2 = run_fcsk(oracle_backend, p1, p2, idx, 1, 3, K, FLAG_SHIFT, 200.0, 1, 2)
    assert sw.view(2, K).tolist() == [[0, 1, 2, 0, 0], [4, 0, 1, 0, 0]]
    assert sh.view(2, K).tolist() == [[0] * 5, [0] * 5]
    assert sb.view(2, K).tolist() == [[0] * 5, [0] * 5]
    assert m.view(2, K).tolist() == [[1, 1, 1, 0, 0]] * 2
    assert float(v1.abs().sum()) == 0 and float(v2.abs().sum()) == 0     # never written


def _fcsk_slow(xyz1, xyz2, idx_n2, kH, kW, K, flag, distance, sh_, sw_):
    """Independent restatement via sorting with an explicit selection-sort (python)."""
    B, H, W, _ = xyz1.shape
    smh, smw = xyz2.shape[1:3]
    N = idx_n2.shape[1]
    ob = torch.zeros(B, N, K, dtype=torch.long); oh = ob.clone(); ow = ob.clone()
    om = torch.zeros(B, N, K)
    d2max = np.float32(distance) * np.float32(distance)
    x1 = xyz1.numpy(); x2 = xyz2.numpy()
    f = np.float32

    def sq(a, b, c):
        t = f(a * a)
        t = f(np.float32(np.float64(b) * np.float64(b) + np.float64(t)))   # fma: exact product + add, one rounding
        return f(np.float32(np.float64(c) * np.float64(c) + np.float64(t)))

    for b in range(B):
        for n in range(N):
            ch, cw = [int(v) for v in idx_n2[b, n]]
            c = x1[b, ch, cw]
            if max(sq(*c), f(1e-10)) <= f(1e-10):
                continue
            D = [f(1e10)] * 150; ih = [0] * 150; iw = [0] * 150
            for m in range(kH * kW):
                h = ch // sh_ + m // kW - kH // 2
                w = cw // sw_ + m % kW - kW // 2
                if h < 0 or h >= smh:
                    continue
                if flag & 2:
                    if w < 0: w += smw
                    if w >= smw: w -= smw
                elif w < 0 or w >= smw:
                    continue
                q = x2[b, h, w]
                if sq(*q) <= f(1e-10):
                    continue
                d = max(sq(f(c[0] - q[0]), f(c[1] - q[1]), f(c[2] - q[2])), f(1e-10))
                if d > d2max:
                    continue
                D[m] = d; ih[m] = h; iw[m] = w
            for s in range(K):
                j = s
                for t in range(s + 1, kH * kW):
                    if D[t] < D[j]: j = t
                D[s], D[j] = D[j], D[s]; ih[s], ih[j] = ih[j], ih[s]; iw[s], iw[j] = iw[j], iw[s]
                if (flag & 1) and s == 0:
                    ob[b, n, :] = b; oh[b, n, :] = ih[0]; ow[b, n, :] = iw[0]; om[b, n, :] = 1
                if D[s] < f(1e10):
                    ob[b, n, s] = b; oh[b, n, s] = ih[s]; ow[b, n, s] = iw[s]; om[b, n, s] = 1
    return ob, oh, ow, om


def test_fcsk_oracle_vs_python_restatement(oracle_backend):
    for seed, lattice, flag in [(0, False, 3), (1, True, 3), (2, True, 2), (3, False, 0), (4, True, 1)]:
        x = range_image(2, 6, 20, seed, empty_frac=0.3, lattice=lattice, scale=3.0)
        idx = stride_grid(2, 3, 10, 2, 2)
        sb, sh, sw, m, _, _ = run_fcsk(oracle_backend, x, x, idx, 3, 5, 6, flag, 2.5, 1, 1)
        ob, oh, ow, om = _fcsk_slow(x, x, idx, 3, 5, 6, flag, 2.5, 1, 1)
        assert torch.equal(sb.squeeze(-1), ob) and torch.equal(sh.squeeze(-1), oh)
        assert torch.equal(sw.squeeze(-1), ow) and torch.equal(m.squeeze(-1), om)


def test_fcsk_rejects_oversized_window(oracle_backend):
    x = range_image(1, 4, 8, 0)
    idx = stride_grid(1, 2, 2, 1, 1)
    try:
        run_fcsk(oracle_backend, x, x, idx, 11, 15, 4, 3, 1.0, 1, 1)
        assert False, "expected failure"
    except RuntimeError:
        pass


def test_opt_n_threads():
    # pointnet2/src/cuda_utils.h:10-14
    for n, want in [(1, 1), (2, 2), (3, 2), (64, 64), (1000, 512), (1023, 512), (1024, 1024), (8192, 1024),
                    (150000, 1024)]:
        assert oracle.opt_n_threads(n) == want


def _fps_tree_python(xyz, m, BS):
    """Literal python emulation of sampling_gpu.cu:93-209 on one sample (float32)."""
    n = xyz.shape[0]
    f = np.float32
    temp = np.full(n, 1e10, np.float32)
    out = [0]
    old = 0
    X = xyz.astype(np.float32)
    for _ in range(1, m):
        dx = X[:, 0] - X[old, 0]; dy = X[:, 1] - X[old, 1]; dz = X[:, 2] - X[old, 2]
        t = (dx * dx).astype(np.float32)
        t = (dy.astype(np.float64) * dy + t).astype(np.float32)
        d = (dz.astype(np.float64) * dz + t).astype(np.float32)
        temp = np.minimum(d, temp)
        D = np.full(BS, -1, np.float32); I = np.zeros(BS, np.int64)
        for tid in range(BS):
            ks = np.arange(tid, n, BS)
            if len(ks):
                j = int(np.argmax(temp[ks]))        # first max
                D[tid] = temp[ks][j]; I[tid] = ks[j]
        s = BS // 2
        while s >= 1:
            for tid in range(s):
                if D[tid + s] > D[tid]:
                    D[tid] = D[tid + s]; I[tid] = I[tid + s]
            s //= 2
        old = int(I[0]); out.append(old)
    return out


def test_fps_oracle_vs_python_tree(oracle_backend):
    for n, m, dup in [(64, 16, 0.0), (100, 30, 0.5), (257, 40, 0.3)]:
        pts = cloud(2, n, seed=n, dup_frac=dup)
        idx = torch.zeros(2, m, dtype=torch.int32)
        temp = torch.full((2, n), 1e10)
        oracle_backend.furthest_point_sampling_wrapper(2, n, m, pts, temp, idx)
        BS = oracle.opt_n_threads(n)
        for b in range(2):
            assert idx[b].tolist() == _fps_tree_python(pts[b].numpy(), m, BS)


def test_ball_query_group_gather_oracle(oracle_backend):
    B, N, M, ns, C = 2, 200, 17, 8, 5
    xyz = cloud(B, N, 5)
    new_xyz = xyz[:, :M].contiguous()
    idx = torch.zeros(B, M, ns, dtype=torch.int32)
    oracle_backend.ball_query_wrapper(B, N, M, 9.0, ns, new_xyz, xyz, idx)
    d2 = ((new_xyz[:, :, None] - xyz[:, None]) ** 2).sum(-1)
    for b in range(B):
        for q in range(M):
            hits = torch.nonzero(d2[b, q] < 81.0 - 1e-3).flatten().tolist()[:ns]
            got = idx[b, q].tolist()
            assert got[:len(hits)] == hits or len(hits) == 0
            if hits:
                assert all(g == hits[0] for g in got[len(hits):])
    feats = torch.randn(B, C, N)
    out = torch.empty(B, C, M, ns)
    oracle_backend.group_points_wrapper(B, C, N, M, ns, feats, idx, out)
    ref = torch.gather(feats.unsqueeze(2).expand(-1, -1, M, -1), 3, idx.long().unsqueeze(1).expand(-1, C, -1, -1))
    assert torch.equal(out, ref)
    g = torch.randn(B, C, M, ns)
    gp = torch.zeros(B, C, N)
    oracle_backend.group_points_grad_wrapper(B, C, N, M, ns, g, idx, gp)
    ref_g = torch.zeros(B, C, N).scatter_add_(2, idx.long().reshape(B, 1, -1).expand(-1, C, -1), g.reshape(B, C, -1))
    assert torch.allclose(gp, ref_g, atol=1e-5)


def test_three_nn_interpolate_oracle(oracle_backend):
    B, N, M, C = 2, 50, 33, 4
    unk, kn = cloud(B, N, 7), cloud(B, M, 8)
    d2 = torch.empty(B, N, 3); idx = torch.empty(B, N, 3, dtype=torch.int32)
    oracle_backend.three_nn_wrapper(B, N, M, unk, kn, d2, idx)
    full = ((unk[:, :, None] - kn[:, None]) ** 2).sum(-1)
    ref_d, ref_i = torch.topk(full, 3, dim=-1, largest=False)
    assert torch.equal(idx.long(), ref_i)
    assert torch.allclose(d2, ref_d, rtol=1e-5, atol=1e-5)
    w = torch.rand(B, N, 3); feats = torch.randn(B, C, M); out = torch.empty(B, C, N)
    oracle_backend.three_interpolate_wrapper(B, C, M, N, feats, idx, w, out)
    gathered = torch.gather(feats.unsqueeze(2).expand(-1, -1, N, -1), 3, idx.long().unsqueeze(1).expand(-1, C, -1, -1))
    assert torch.allclose(out, (gathered * w.unsqueeze(1)).sum(-1), atol=1e-5)


def _mul_q_formula(a, b):
    """the reference's elementwise formula (src/modules/warp_utils.py:41-53) in plain torch"""
    r0 = a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3]
    r1 = a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2]
    r2 = a[..., 0] * b[..., 2] - a[..., 1] * b[..., 3] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1]
    r3 = a[..., 0] * b[..., 3] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1] + a[..., 3] * b[..., 0]
    return torch.stack([r0, r1, r2, r3], -1)


def test_quat_unit_oracle_and_autograd(oracle_backend):
    """inv_q (warp_utils.py:10-22) and the pose head's normalisation (PPBackbone_center.py:562): the oracle equals the
    reference expressions, and the autograd nodes give the gradients autograd derives for those expressions."""
    from i2pnet_amd import ops, warp
    g = torch.Generator().manual_seed(13)
    conj = torch.tensor([1.0, -1.0, -1.0, -1.0])
    exprs = {0: lambda q: (q * conj) / (torch.sum(q * q, dim=-1, keepdim=True) + 1e-10),
             1: lambda q: q / (torch.sqrt(torch.sum(q * q, dim=-1, keepdim=True) + 1e-10) + 1e-10)}
    prev = ops.set_backend(oracle_backend)
    try:
        for mode, fn in ((0, warp.inv_q), (1, warp.normalise_q)):
            q = torch.randn(9, 4, generator=g) * torch.logspace(-2, 1, 9).unsqueeze(-1)
            w = torch.randn(9, 4, generator=g)
            q1, q2 = q.clone().requires_grad_(), q.clone().requires_grad_()
            o1, o2 = fn(q1), exprs[mode](q2)
            assert torch.allclose(o1, o2, rtol=1e-6, atol=1e-7)
            (o1 * w).sum().backward(); (o2 * w).sum().backward()
            assert torch.allclose(q1.grad, q2.grad, rtol=1e-5, atol=1e-6), mode
        assert warp.inv_q(torch.randn(3, 1, 4, generator=g)).shape == (3, 4)
        z = torch.zeros(2, 4, requires_grad=True)                   # all-zero quaternion: finite (the 1e-10 terms)
        warp.normalise_q(z).sum().backward()
        assert torch.isfinite(z.grad).all()
    finally:
        ops.set_backend(prev)


def test_quat_mul_oracle_and_autograd(oracle_backend):
    """oracle quat_mul == the elementwise formula bit for bit (all broadcast shapes), and the
    custom autograd node of i2pnet_amd.warp.mul_q gives the formula's gradients."""
    from i2pnet_amd import ops, warp
    g = torch.Generator().manual_seed(3)
    prev = ops.set_backend(oracle_backend)
    try:
        for na, nb in [(1, 1), (1, 37), (37, 1), (37, 37)]:
            a = torch.randn(3, na, 4, generator=g); b = torch.randn(3, nb, 4, generator=g)
            assert torch.equal(oracle_backend.quat_mul(a, b), _mul_q_formula(a, b))
            conj = torch.tensor([1.0, -1.0, -1.0, -1.0])
            assert torch.equal(oracle_backend.quat_mul(a, b, conj_a=True), _mul_q_formula(a * conj, b))
            assert torch.equal(oracle_backend.quat_mul(a, b, conj_b=True), _mul_q_formula(a, b * conj))
            a1, b1 = a.clone().requires_grad_(), b.clone().requires_grad_()
            a2, b2 = a.clone().requires_grad_(), b.clone().requires_grad_()
            w = torch.randn(3, max(na, nb), 4, generator=g)
            (warp.mul_q(a1, b1) * w).sum().backward()
            (_mul_q_formula(a2, b2) * w).sum().backward()
            assert torch.allclose(a1.grad, a2.grad, rtol=1e-5, atol=1e-5)
            assert torch.allclose(b1.grad, b2.grad, rtol=1e-5, atol=1e-5)
        # [B,4] operands and the warp of a cloud
        q = torch.randn(2, 4, generator=g); t = torch.cat([torch.zeros(2, 1), torch.randn(2, 3, generator=g)], 1)
        p = torch.randn(2, 50, 3, generator=g)
        out = warp.warp_quat_xyz(p, q, t)
        homo = torch.cat([torch.zeros(2, 50, 1), p], -1)
        want = (_mul_q_formula(_mul_q_formula(q.unsqueeze(1), homo), warp.inv_q(q).unsqueeze(1)) + t.unsqueeze(1))[..., 1:]
        assert torch.equal(out, want)
    finally:
        ops.set_backend(prev)


def test_row_unitvar_oracle_vs_torch(oracle_backend):
    """oracle row_unitvar == (x-mean)/clip(std,1e-12) and its autograd, incl. a constant row (clipped std)."""
    from i2pnet_amd import modules, ops
    g = torch.Generator().manual_seed(7)
    prev = ops.set_backend(oracle_backend)
    try:
        for c in (64, 128, 37):
            x = torch.randn(50, c, generator=g) * 3 + 1
            x[7] = 2.5                                       # zero variance: std clipped at 1e-12
            x1 = x.clone().requires_grad_(); x2 = x.clone().requires_grad_()
            y1 = modules._unit_variance(x1.view(5, 10, c))
            y2 = (x2 - torch.mean(x2, -1, keepdim=True)) / torch.clip(torch.std(x2, -1, keepdim=True), min=1e-12)
            assert torch.allclose(y1.view(50, c), y2, rtol=1e-5, atol=1e-6)
            w = torch.randn(50, c, generator=g)
            (y1.view(50, c) * w).sum().backward(); (y2 * w).sum().backward()
            keep = torch.ones(50, dtype=torch.bool); keep[7] = False        # torch's std backward is NaN at std == 0
            assert torch.allclose(x1.grad[keep], x2.grad[keep], rtol=1e-4, atol=1e-5)
            assert torch.isfinite(x1.grad).all()
    finally:
        ops.set_backend(prev)


@pytest.mark.parametrize("fused", [False, True])
def test_image_cnn_bias_skip_equivalent(oracle_backend, fused, monkeypatch):
    """training-mode image encoder without the (cancelling) conv bias == the plain Sequential:
    same output, same running statistics, zero bias gradient — for the torch tail and the fused tail
    (oracle operators on CPU)."""
    import copy
    import torch.nn as nn
    from i2pnet_amd import modules, ops
    from i2pnet_amd.modules import createCNNs
    monkeypatch.setattr(modules, "USE_FUSED_IMG", fused)
    monkeypatch.setattr(ops, "_active", oracle_backend)
    torch.manual_seed(0)
    fast = createCNNs(3, [8, 8], [2, 1])
    for m in fast:
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.bias, std=0.5)
    plain = nn.Sequential(*copy.deepcopy(list(fast)))
    x = torch.randn(2, 3, 20, 24)
    fast.train(); plain.train()
    yf = fast(x); yp = plain(x)
    assert torch.allclose(yf, yp, rtol=1e-4, atol=1e-5)
    for a, b in zip(fast.buffers(), plain.buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-5)
    yf.square().sum().backward(); yp.square().sum().backward()
    for (n, a), b in zip(fast.named_parameters(), plain.parameters()):
        if n.endswith("0.bias") or n.endswith("4.bias"):
            assert a.grad is None or float(a.grad.abs().max()) == 0.0
            assert float(b.grad.abs().max()) < 1e-3 * float(yp.square().sum())
        else:
            assert torch.allclose(a.grad, b.grad, rtol=2e-3, atol=1e-3 * float(b.grad.abs().max()))
    fast.eval(); plain.eval()
    assert torch.allclose(fast(x), plain(x), rtol=1e-4, atol=1e-5)


def _img_tail_torch(y, bn, slope, stride):
    import torch.nn.functional as F
    z = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    return F.max_pool2d(F.leaky_relu(z, slope), 3, stride, 1)


@pytest.mark.parametrize("stride,C,H,W", [(1, 16, 9, 13), (2, 16, 9, 13), (2, 32, 10, 12), (1, 128, 5, 7)])
def test_img_bn_pool_oracle_vs_torch(oracle_backend, stride, C, H, W):
    """oracle image-block tail == F.batch_norm(train) -> leaky_relu -> max_pool2d(3, stride, 1): output,
    running buffers, and gradients w.r.t. the conv output, gamma and beta."""
    import torch.nn as nn
    from i2pnet_amd import modules, ops
    g = torch.Generator().manual_seed(stride * 100 + C)
    B = 2
    y = torch.randn(B, C, H, W, generator=g).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g) * 0.3
    bn_a, bn_b = nn.BatchNorm2d(C), nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_a.weight.copy_(torch.randn(C, generator=g)); bn_a.bias.copy_(torch.randn(C, generator=g) * 0.2)   # negative gammas too
        bn_a.running_mean.normal_(generator=g); bn_a.running_var.uniform_(0.5, 1.5, generator=g)
    bn_b.load_state_dict(bn_a.state_dict())
    prev = ops.set_backend(oracle_backend)
    try:
        y1 = y.clone().requires_grad_(); y2 = y.clone().requires_grad_()
        out1 = modules._BnActPool.apply(y1, bn_a.weight, bn_a.bias, bias, bn_a.running_mean, bn_a.running_var, stride,
                                        bn_a.momentum, bn_a.eps, 0.1)
        out2 = _img_tail_torch(y2 + bias.view(1, -1, 1, 1), bn_b, 0.1, stride)
        assert out1.shape == out2.shape
        assert torch.allclose(out1, out2, rtol=1e-4, atol=1e-5)
        assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-4, atol=1e-6)
        w = torch.randn(out2.shape, generator=g)
        (out1 * w).sum().backward(); (out2 * w).sum().backward()
        sc = float(y2.grad.abs().max())
        assert torch.allclose(y1.grad, y2.grad, rtol=1e-3, atol=1e-4 * sc)
        assert torch.allclose(bn_a.weight.grad, bn_b.weight.grad, rtol=1e-3, atol=1e-4 * float(bn_b.weight.grad.abs().max()))
        assert torch.allclose(bn_a.bias.grad, bn_b.bias.grad, rtol=1e-3, atol=1e-4 * float(bn_b.bias.grad.abs().max()))
    finally:
        ops.set_backend(prev)


def test_maxk_tail_oracle_vs_torch(oracle_backend):
    """fused set-abstraction tail through mlp_stack(pool_k) == unfused stack + torch.max (values and gradients),
    incl. an input whose channel count is padded by cat_padded and duplicated neighbour rows (exact ties)."""
    from i2pnet_amd import modules, ops
    torch.manual_seed(0)
    prev = ops.set_backend(oracle_backend)
    try:
        B, N, K = 2, 13, 8
        convs = torch.nn.ModuleList([modules.Conv2d(7, 16, bn=True, leaky_relu=False), modules.Conv2d(16, 32, bn=True, leaky_relu=False)])
        for c in convs:
            torch.nn.init.normal_(c.bn_linear.weight, 1.0, 0.3); torch.nn.init.normal_(c.bn_linear.bias, 0.0, 0.3)
        a = torch.randn(B, N, K, 3); b = torch.randn(B, N, K, 4)
        a[:, :, 5:] = a[:, :, :1]; b[:, :, 5:] = b[:, :, :1]            # FLAG_COPY-style duplicated neighbours
        a1, b1 = a.clone().requires_grad_(), b.clone().requires_grad_()
        a2, b2 = a.clone().requires_grad_(), b.clone().requires_grad_()
        out1 = modules.run_stack(modules.cat_padded([a1, b1]), convs, pool_k=K)
        modules.USE_FUSED_MLP = False
        try:
            out2 = modules.run_stack(torch.cat([a2, b2], -1), convs, pool_k=K)
        finally:
            modules.USE_FUSED_MLP = True
        assert out1.shape == (B, N, 32)
        assert torch.allclose(out1, out2, rtol=1e-4, atol=1e-5)
        w = torch.randn(B, N, 32)
        g1 = torch.autograd.grad((out1 * w).sum(), [a1, b1] + [p for p in convs.parameters() if p.requires_grad])
        g2 = torch.autograd.grad((out2 * w).sum(), [a2, b2] + [p for p in convs.parameters() if p.requires_grad])
        # duplicated rows may receive the gradient at either copy: compare sums over the duplicates
        for x, y in zip(g1[:2], g2[:2]):
            xs = torch.cat([x[:, :, :1] + x[:, :, 5:].sum(2, keepdim=True), x[:, :, 1:5]], 2)
            ys = torch.cat([y[:, :, :1] + y[:, :, 5:].sum(2, keepdim=True), y[:, :, 1:5]], 2)
            assert torch.allclose(xs, ys, rtol=1e-3, atol=1e-4 * float(ys.abs().max()))
        for x, y in zip(g1[2:], g2[2:]):
            assert torch.allclose(x, y, rtol=1e-3, atol=1e-4 * float(y.abs().max()) + 1e-6)
    finally:
        ops.set_backend(prev)


def test_max_response_matches_literal_formulation():
    """closed-form backward-validation feature == the reference's literal max over the masked [B,N,M,C] product
    (PPBackbone_center.py:408-414), values and gradients, incl. a sample without any valid point."""
    from i2pnet_amd.modules import _MaxResponse
    g = torch.Generator().manual_seed(11)
    B, N, M, C = 3, 17, 9, 8
    pts = torch.randn(B, N, C, generator=g); pix = torch.randn(B, M, C, generator=g)
    valid = (torch.rand(B, N, 1, generator=g) > 0.3).float()
    valid[2] = 0.0
    p1, x1 = pts.clone().requires_grad_(), pix.clone().requires_grad_()
    p2, x2 = pts.clone().requires_grad_(), pix.clone().requires_grad_()
    r1 = _MaxResponse.apply(p1, x1, valid)
    corr = p2.unsqueeze(2) * x2.unsqueeze(1)                                        # [B,N,M,C]
    masked = corr * valid.unsqueeze(2) + -1e10 * (1 - valid.unsqueeze(2))
    r2 = masked.max(1)[0]
    assert torch.allclose(r1, r2, rtol=1e-6, atol=1e-6)
    w = torch.randn(B, M, C, generator=g)
    (r1 * w).sum().backward(); (r2 * w).sum().backward()
    assert torch.allclose(x1.grad[:2], x2.grad[:2], rtol=1e-5, atol=1e-6)
    assert torch.allclose(p1.grad[:2], p2.grad[:2], rtol=1e-5, atol=1e-6)
    assert float(p1.grad[2].abs().max()) == 0.0 and float(x1.grad[2].abs().max()) == 0.0


def test_softmax_pool_matches_torch(oracle_backend):
    """PoseHead pooling on the softmax-weighted-sum kernels == softmax(mask, 1) * value summed over points,
    incl. the -1e10 masked rows (values and both gradients)."""
    from i2pnet_amd import ops
    from i2pnet_amd.fused import softmax_pool
    g = torch.Generator().manual_seed(2)
    B, N, C = 3, 57, 64
    mask = torch.randn(B, N, C, generator=g) * 3; val = torch.randn(B, N, C, generator=g)
    mask[:, 10:20] = -1e10
    prev = ops.set_backend(oracle_backend)
    try:
        m1, v1 = mask.clone().requires_grad_(), val.clone().requires_grad_()
        m2, v2 = mask.clone().requires_grad_(), val.clone().requires_grad_()
        o1 = softmax_pool(m1, v1)
        o2 = torch.sum(v2 * torch.softmax(m2, dim=1), dim=1, keepdim=True)
        assert torch.allclose(o1, o2, rtol=1e-5, atol=1e-6)
        w = torch.randn(B, 1, C, generator=g)
        (o1 * w).sum().backward(); (o2 * w).sum().backward()
        assert torch.allclose(v1.grad, v2.grad, rtol=1e-4, atol=1e-6)
        assert torch.allclose(m1.grad, m2.grad, rtol=1e-4, atol=1e-6)
    finally:
        ops.set_backend(prev)


@pytest.mark.parametrize("l1", [True, False])
def test_pose_loss_oracle_vs_torch(oracle_backend, l1):
    """fused pose loss (value + gradient in one call) == the torch formulation of compute_loss.py:102-133 and its
    autograd, for both translation terms"""
    from i2pnet_amd import loss as L, ops

    class Cfg:
        l1_trans_loss = l1
    g = torch.Generator().manual_seed(4)
    B = 5
    o3, o4 = torch.randn(B, 7, generator=g), torch.randn(B, 7, generator=g)
    qg, tg = torch.randn(B, 4, generator=g), torch.randn(B, 3, generator=g)
    prev = ops.set_backend(oracle_backend)
    try:
        res = []
        for fused in (True, False):
            L.USE_FUSED_LOSS = fused
            a3, a4 = o3.clone().requires_grad_(), o4.clone().requires_grad_()
            wx, wq = torch.tensor([0.3], requires_grad=True), torch.tensor([-2.5], requires_grad=True)
            loss, real, dual = L.Get_loss(a3, a4, qg, tg, wx, wq, Cfg)
            (loss * 1.7).sum().backward()
            res.append((loss.detach(), real.detach().reshape(-1), dual.detach().reshape(-1), a3.grad, a4.grad, wx.grad, wq.grad))
    finally:
        L.USE_FUSED_LOSS = True
        ops.set_backend(prev)
    for a, b in zip(*res):
        assert a.shape == b.shape or a.numel() == b.numel()
        assert torch.allclose(a.reshape(-1), b.reshape(-1), rtol=1e-5, atol=1e-6)


def test_zero_arena_prefix_reset():
    """ops.zeros hands out slices of one buffer that `begin_step` clears with a single memset over the prefix ever used:
    slices come back zero, at the same addresses, and requests above the limit fall back to torch.zeros."""
    from i2pnet_amd import ops
    d = torch.device("cpu")
    ops.begin_step(d)
    a = ops.zeros((1000,), torch.float32, d); a.fill_(3)
    b = ops.zeros((300 << 10,), torch.uint8, d); b.fill_(7)
    ops.begin_step(d)
    a2 = ops.zeros((1000,), torch.float32, d); b2 = ops.zeros((300 << 10,), torch.uint8, d)
    c2 = ops.zeros((100 << 10,), torch.float32, d)                  # beyond the previous step's high-water mark
    assert a2.data_ptr() == a.data_ptr() and float(a2.abs().sum()) == 0 and int(b2.sum()) == 0 and float(c2.sum()) == 0
    c2.fill_(1)
    ops.begin_step(d)
    big = ops.zeros((5 << 20,), torch.float32, d)                   # 20 MB: above the arena's request limit (16 MB), plain torch.zeros
    a3 = ops.zeros((1000,), torch.float32, d); ops.zeros((300 << 10,), torch.uint8, d)
    c3 = ops.zeros((100 << 10,), torch.float32, d)
    assert float(big.sum()) == 0 and a3.data_ptr() == a.data_ptr() and c3.data_ptr() == c2.data_ptr() and float(c3.sum()) == 0
    ops.end_step(d)
    assert float(ops.zeros((4,), torch.float32, d).sum()) == 0      # no arena active: plain torch.zeros


def test_project_seq_is_differentiable_in_its_values(oracle_backend):
    """the reference's project_seq scatters with index_put_ (utils.py:173-177): gradients flow to the scattered
    values (not through the cell indices).  Ours: a gather along the winner map."""
    from i2pnet_amd import ops, projectpn as P, synth
    prev = ops.set_backend(oracle_backend)
    try:
        xyz = synth.lidar_scan(2, 512, torch.Generator().manual_seed(0)).requires_grad_(True)
        feat = torch.randn(2, 512, 5, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
        img, (fimg,) = P.project_seq(xyz, [feat], 16, 90, use_rank=False)
        wx = torch.randn(img.shape, generator=torch.Generator().manual_seed(2))
        wf = torch.randn(fimg.shape, generator=torch.Generator().manual_seed(3))
        ((img * wx).sum() + (fimg * wf).sum()).backward()
        # plain-torch restatement: last writer (highest index) wins a cell
        with torch.no_grad():
            cells = synth.spherical_cells(xyz.detach(), 16, 90)
        gx = torch.zeros_like(xyz); gf = torch.zeros_like(feat)
        for b in range(2):
            win = {}
            for n in range(512):
                if bool((xyz[b, n] != 0).any()) or True:
                    win[int(cells[b, n])] = n
            for cell, n in win.items():
                gx[b, n] = wx[b].reshape(-1, 3)[cell]; gf[b, n] = wf[b].reshape(-1, 5)[cell]
        assert torch.allclose(xyz.grad, gx) and torch.allclose(feat.grad, gf)
    finally:
        ops.set_backend(prev)


# ---- independent pure-Python restatements of the two remaining pointnet2 search kernels -------------------------
def _grid_cloud(B, N, seed, span=4):
    """coordinates on a 1/4 grid (every product and sum below is exact in fp32, so ties and the r^2 boundary are hit
    exactly) with repeated points"""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-4 * span, 4 * span + 1, (B, N, 3), generator=g).float() / 4


def _ball_query_python(new_xyz, xyz, radius, ns, idx):
    """pointnet2/src/ball_query_gpu.cu:9-47, line by line: first `ns` indices with d2 < r^2 in index order, the tail
    filled with the first hit, rows without a hit left untouched"""
    r2 = np.float32(radius) * np.float32(radius)
    for b in range(xyz.shape[0]):
        for q in range(new_xyz.shape[1]):
            cnt = 0
            for k in range(xyz.shape[1]):
                d = new_xyz[b, q] - xyz[b, k]
                d2 = np.float32(d[0] * d[0]) + np.float32(d[1] * d[1]) + np.float32(d[2] * d[2])
                if d2 < r2:
                    if cnt == 0:
                        idx[b, q, :] = k
                    idx[b, q, cnt] = k
                    cnt += 1
                    if cnt >= ns:
                        break
    return idx


def test_ball_query_oracle_vs_python_restatement(oracle_backend):
    B, N, M, ns = 2, 60, 25, 6
    xyz = _grid_cloud(B, N, 3)
    new_xyz = torch.cat([xyz[:, :M - 2], torch.full((B, 2, 3), 50.0)], 1).contiguous()    # the last two queries hit nothing
    for radius in (0.75, 1.25, 2.0):                         # 0.75^2 = 0.5625, 1.25^2 = 1.5625: reachable on the grid (boundary excluded)
        idx = torch.full((B, M, ns), -7, dtype=torch.int32)
        oracle_backend.ball_query_wrapper(B, N, M, radius, ns, new_xyz, xyz, idx)
        want = _ball_query_python(new_xyz.numpy(), xyz.numpy(), radius, ns, np.full((B, M, ns), -7, np.int32))
        assert np.array_equal(idx.numpy(), want), radius
        assert (idx[:, -2:] == -7).all()


def _three_nn_python(unknown, known):
    """pointnet2/src/interpolate_gpu.cu:9-52: running best three with strict `<` (the first of equal distances wins),
    double-typed bests initialised to 1e40, float distances"""
    B, N = unknown.shape[:2]
    d2 = np.zeros((B, N, 3), np.float32); idx = np.zeros((B, N, 3), np.int32)
    for b in range(B):
        for i in range(N):
            best = [1e40, 1e40, 1e40]; bi = [0, 0, 0]
            for k in range(known.shape[1]):
                v = unknown[b, i] - known[b, k]
                d = float(np.float32(v[0] * v[0]) + np.float32(v[1] * v[1]) + np.float32(v[2] * v[2]))
                if d < best[0]:
                    best = [d, best[0], best[1]]; bi = [k, bi[0], bi[1]]
                elif d < best[1]:
                    best = [best[0], d, best[1]]; bi = [bi[0], k, bi[1]]
                elif d < best[2]:
                    best[2] = d; bi[2] = k
            with np.errstate(over="ignore"):
                d2[b, i] = np.array(best, np.float64).astype(np.float32); idx[b, i] = bi
    return d2, idx


@pytest.mark.parametrize("M", [2, 3, 40])
def test_three_nn_oracle_vs_python_restatement(oracle_backend, M):
    B, N = 2, 30
    unk, kn = _grid_cloud(B, N, 11, span=2), _grid_cloud(B, M, 12, span=2)      # small span: many equal distances
    d2 = torch.empty(B, N, 3); idx = torch.empty(B, N, 3, dtype=torch.int32)
    oracle_backend.three_nn_wrapper(B, N, M, unk, kn, d2, idx)
    wd, wi = _three_nn_python(unk.numpy(), kn.numpy())
    assert np.array_equal(idx.numpy(), wi)
    assert np.array_equal(d2.numpy(), wd)                   # (M = 2: the third best stays 1e40 -> inf as float, index 0)
