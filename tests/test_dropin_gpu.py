"""The drop-in boundary as the REFERENCE would use it: `dropin/` is put on sys.path, the two extension
modules are imported by the names the reference imports (`fused_conv_select_k_cuda`, fused_conv_select_k.py:4;
`pointnet2.pointnet2_cuda`, pointnet2_utils.py:9) and called with the reference wrappers' argument lists and
pre-fills (projectPN/utils.py:84-102; pointnet2_utils.py:55-56,98,177,221,249).  Checked against the CPU oracle.
Also: the autograd Functions / modules of i2pnet_amd.pointnet2_utils that no other test reaches (B5)."""
import importlib
import sys
from pathlib import Path

import pytest
import torch

from helpers import cloud, range_image, stride_grid

ROOT = Path(__file__).resolve().parent.parent
DEV = "cuda:0"


@pytest.fixture(scope="module")
def dropin():
    sys.path.insert(0, str(ROOT / "dropin"))
    try:
        fcsk = importlib.import_module("fused_conv_select_k_cuda")
        p2 = importlib.import_module("pointnet2.pointnet2_cuda")
        yield fcsk, p2
    finally:
        sys.path.remove(str(ROOT / "dropin"))


@pytest.mark.gpu
def test_fused_conv_select_k_cuda_module(dropin, oracle_backend):
    fcsk, _ = dropin
    FLAG_COPY, FLAG_SHIFT = 1, 2
    B, H, W, oh, ow, sh, sw, kH, kW, K, dist = 2, 16, 90, 8, 45, 2, 2, 5, 9, 8, 6.0
    xyz = range_image(B, H, W, seed=3)
    idx_n2 = stride_grid(B, oh, ow, sh, sw)

    def call(fn, dev):                              # get_neighbor_copy, utils.py:76-102, verbatim allocation pattern
        x = xyz.to(dev); ix = idx_n2.to(dev)
        kt = kH * kW
        n = ix.shape[1]
        random_hw = torch.arange(0, kt, device=dev, dtype=torch.int)
        sb = torch.zeros(B, n, K, 1, device=dev, dtype=torch.long)
        shh = torch.zeros(B, n, K, 1, device=dev, dtype=torch.long)
        sww = torch.zeros(B, n, K, 1, device=dev, dtype=torch.long)
        v1 = torch.zeros(B, n, kt, 1, device=dev, dtype=torch.float)
        v2 = torch.zeros(B, n, kt, 1, device=dev, dtype=torch.float)
        m = torch.zeros(B, n, K, 1, device=dev, dtype=torch.float)
        ret = fn(x, x, ix, random_hw, H, W, n, kH, kW, K, FLAG_SHIFT | FLAG_COPY, dist, 1, 1, sb, shh, sww, v1, v2, m, H, W)
        return ret, [t.cpu() for t in (sb, shh, sww, m, v1, v2)]

    ret, got = call(fcsk.fused_conv_select_k, DEV)
    assert ret is None                              # the pybind function returns void (fused_conv_g.cpp:69-72)
    _, want = call(oracle_backend.fused_conv_select_k, "cpu")
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    assert got[3].sum() > 0
    with pytest.raises(RuntimeError):               # TORCH_CHECK(is_cuda) -> RuntimeError (fused_conv_g.cpp:11)
        call(fcsk.fused_conv_select_k, "cpu")


@pytest.mark.gpu
def test_pointnet2_cuda_module(dropin, oracle_backend):
    _, p2 = dropin
    cpu = oracle_backend
    B, N, M, C = 2, 1000, 128, 16
    xyz = cloud(B, N, seed=1, dup_frac=0.05)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, C, N, generator=g)

    # furthest_point_sampling_wrapper: output IntTensor(B, npoint), temp filled with 1e10 (pointnet2_utils.py:55-56)
    def fps(mod, dev):
        out = torch.empty(B, M, dtype=torch.int32, device=dev); temp = torch.full((B, N), 1e10, device=dev)
        mod.furthest_point_sampling_wrapper(B, N, M, xyz.to(dev), temp, out)
        return out.cpu(), temp.cpu()
    (i_g, t_g), (i_c, t_c) = fps(p2, DEV), fps(cpu, "cpu")
    assert torch.equal(i_g, i_c) and torch.equal(t_g, t_c)
    idx = i_c

    # gather_points_wrapper / _grad_ (grad buffer zeroed by the caller, :98)
    def gather(mod, dev):
        out = torch.empty(B, C, M, device=dev)
        mod.gather_points_wrapper(B, C, N, M, feats.to(dev), idx.to(dev), out)
        gin = torch.zeros(B, C, N, device=dev)
        mod.gather_points_grad_wrapper(B, C, N, M, out, idx.to(dev), gin)
        return out.cpu(), gin.cpu()
    (o_g, gi_g), (o_c, gi_c) = gather(p2, DEV), gather(cpu, "cpu")
    assert torch.equal(o_g, o_c) and torch.allclose(gi_g, gi_c, rtol=1e-5, atol=1e-6)
    new_xyz = o_c[:, :3].transpose(1, 2).contiguous() if C >= 3 else None
    new_xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()

    # ball_query_wrapper: idx zero-filled by the caller (:249)
    def ball(mod, dev):
        out = torch.zeros(B, M, 16, dtype=torch.int32, device=dev)
        mod.ball_query_wrapper(B, N, M, 6.0, 16, new_xyz.to(dev), xyz.to(dev), out)
        return out.cpu()
    bq = ball(cpu, "cpu")
    assert torch.equal(ball(p2, DEV), bq)

    # group_points_wrapper / _grad_ (:203,:221)
    def group(mod, dev):
        out = torch.empty(B, C, M, 16, device=dev)
        mod.group_points_wrapper(B, C, N, M, 16, feats.to(dev), bq.to(dev), out)
        gin = torch.zeros(B, C, N, device=dev)
        mod.group_points_grad_wrapper(B, C, N, M, 16, out, bq.to(dev), gin)
        return out.cpu(), gin.cpu()
    (o_g, gi_g), (o_c, gi_c) = group(p2, DEV), group(cpu, "cpu")
    assert torch.equal(o_g, o_c) and torch.allclose(gi_g, gi_c, rtol=1e-4, atol=1e-5)

    # three_nn_wrapper / three_interpolate_wrapper / _grad_ (:125-127,:159,:177)
    def three(mod, dev):
        d2 = torch.empty(B, N, 3, device=dev); ix = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
        mod.three_nn_wrapper(B, N, M, xyz.to(dev), new_xyz.to(dev), d2, ix)
        w = 1.0 / (torch.sqrt(d2) + 1e-8); w = (w / w.sum(2, keepdim=True)).contiguous()
        kf = feats[:, :, :M].contiguous().to(dev)
        out = torch.empty(B, C, N, device=dev)
        mod.three_interpolate_wrapper(B, C, M, N, kf, ix, w, out)
        gin = torch.zeros(B, C, M, device=dev)
        mod.three_interpolate_grad_wrapper(B, C, N, M, out, ix, w, gin)
        return d2.cpu(), ix.cpu(), out.cpu(), gin.cpu()
    g3, c3 = three(p2, DEV), three(cpu, "cpu")
    assert torch.equal(g3[0], c3[0]) and torch.equal(g3[1], c3[1])
    assert torch.allclose(g3[2], c3[2], rtol=1e-5, atol=1e-6) and torch.allclose(g3[3], c3[3], rtol=1e-4, atol=1e-4)

    # knn_wrapper: called by pointnet2_utils.KNN (:32) as (B, N, m, k, unknown, known, dist2, idx)
    k = 8
    d2 = torch.empty(B, M, k, device=DEV); ix = torch.empty(B, M, k, dtype=torch.int32, device=DEV)
    p2.knn_wrapper(B, M, N, k, new_xyz.to(DEV), xyz.to(DEV), d2, ix)
    full = ((new_xyz.unsqueeze(2) - xyz.unsqueeze(1)) ** 2).sum(-1)                  # [B,M,N]
    want_d, _ = torch.topk(full, k, dim=2, largest=False, sorted=True)
    assert torch.allclose(d2.cpu().sort(dim=2)[0], want_d, rtol=1e-5, atol=1e-5)
    assert torch.allclose(torch.gather(full, 2, ix.cpu().long()), d2.cpu(), rtol=1e-5, atol=1e-5)


def _b5_case(dev):
    from i2pnet_amd import pointnet2_utils as pu
    B, N, M, C = 2, 512, 64, 8
    xyz = cloud(B, N, seed=4).to(dev)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B, C, N, generator=g).to(dev).requires_grad_(True)
    idx = pu.furthest_point_sample(xyz, M)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    out = {"fps": idx}
    # QueryAndGroup (+ backward through GroupingOperation)
    qg = pu.QueryAndGroup(8.0, 16, use_xyz=True)(xyz, new_xyz, feats)
    assert qg.shape == (B, 3 + C, M, 16)
    out["qg"] = qg
    out["qg_nofeat"] = pu.QueryAndGroup(8.0, 16, use_xyz=True)(xyz, new_xyz, None)
    out["qg_noxyz"] = pu.QueryAndGroup(8.0, 16, use_xyz=False)(xyz, new_xyz, feats)
    ga = pu.GroupAll(use_xyz=True)(xyz, None, feats)
    assert ga.shape == (B, 3 + C, 1, N)
    out["ga"] = ga
    # GatherOperation backward
    gat = pu.gather_operation(feats, idx)
    # ThreeNN + ThreeInterpolate (+ backward)
    dist, i3 = pu.three_nn(xyz, new_xyz)
    w = 1.0 / (dist + 1e-8); w = (w / w.sum(2, keepdim=True)).contiguous()
    known_f = gat                                                                # [B,C,M]
    interp = pu.three_interpolate(known_f, i3, w)
    out["interp"] = interp
    # KNN
    kd, ki = pu.knn(8, new_xyz, xyz)
    out["knn_d"] = kd
    loss = (qg ** 2).sum() * 0.5 + (interp * torch.linspace(0.5, 1.5, N, device=dev)).sum() + (gat ** 3).sum() + ga.sum() * 0.1
    loss.backward()
    out["dfeats"] = feats.grad
    return {k: v.detach().cpu() for k, v in out.items()}, ki.cpu(), xyz.cpu(), new_xyz.cpu()


@pytest.mark.gpu
def test_b5_modules_and_backward_against_oracle(oracle_backend):
    from i2pnet_amd import ops
    got, ki_g, xyz, new_xyz = _b5_case(DEV)
    prev = ops.set_backend(oracle_backend)
    try:
        want, ki_c, _, _ = _b5_case("cpu")
    finally:
        ops.set_backend(prev)
    for k in ("fps", "qg_nofeat", "qg_noxyz", "qg", "ga"):
        assert torch.equal(got[k], want[k]), k                     # indices and pure copies: bit-exact
    for k in ("interp", "knn_d", "dfeats"):
        assert torch.allclose(got[k], want[k], rtol=1e-4, atol=1e-4), (k, float((got[k] - want[k]).abs().max()))
    # kNN returns the same neighbour SETS (order within equal distances is free, `topk(sorted=False)` upstream)
    assert torch.equal(ki_g.sort(dim=2)[0], ki_c.sort(dim=2)[0])
