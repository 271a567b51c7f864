"""GPU parity: every C-ABI entry of libi2p_ops.so against the CPU oracle on the same seeded
inputs.  Integer / index outputs must be bit-exact; float copies bit-exact; atomically
accumulated gradients within 1e-5 (summation order differs)."""
import pytest
import torch

from helpers import cloud, range_image, run_fcsk, stride_grid

pytestmark = pytest.mark.gpu
FLAG_COPY, FLAG_SHIFT = 1, 2
DEV = "cuda"


def _both_fcsk(ob, hb, xyz1, xyz2, idx, kH, kW, K, flag, dist, sh, sw, random_hw=None):
    ref = run_fcsk(ob, xyz1, xyz2, idx, kH, kW, K, flag, dist, sh, sw, random_hw)
    got = run_fcsk(hb, xyz1.to(DEV), xyz2.to(DEV), idx.to(DEV), kH, kW, K, flag, dist, sh, sw,
                   None if random_hw is None else random_hw.to(DEV))
    torch.cuda.synchronize()
    for name, r, g in zip(["b", "h", "w", "mask", "valid_idx", "valid_in_dis"], ref, got):
        assert torch.equal(r, g.cpu()), f"{name} differs: {(r != g.cpu()).sum().item()} of {r.numel()}"


# (kernel, K, flag, distance, stride) signatures of the model on reduced grids
SIGS = [
    ((9, 15), 32, 3, 0.75, (1, 1)),   # LiDAR_lv1
    ((9, 15), 16, 3, 3.0, (1, 1)),    # LiDAR_lv2
    ((5, 9), 16, 3, 6.0, (1, 1)),     # LiDAR_lv3/4, layer_idx
    ((5, 9), 8, 3, 9.0, (1, 2)),      # set_upconv (coarse image searched with stride)
    ((3, 5), 4, 2, 4.5, (1, 1)),      # cost volume pc-stage (FLAG_SHIFT only)
    ((1, 3), 5, 2, 200.0, (1, 2)),    # reference smoke-test signature
    ((10, 15), 7, 0, 5.0, (1, 1)),    # kt = 150 (max), no wrap, no copy
    ((2, 2), 9, 1, 5.0, (1, 1)),      # K > kt, copy without wrap
]


@pytest.mark.parametrize("sig", SIGS)
@pytest.mark.parametrize("lattice", [False, True])
def test_fcsk_parity_model_signatures(oracle_backend, hip_backend, sig, lattice):
    (kH, kW), K, flag, dist, (sh, sw) = sig
    B, H, W = 3, 16, 90
    x1 = range_image(B, H, W, seed=kH * 100 + K, empty_frac=0.25, lattice=lattice, scale=4.0)
    if (sh, sw) == (1, 1):
        x2 = x1
        idx = stride_grid(B, H // 2, W // 3, 2, 3)
    else:
        x2 = range_image(B, H // sh, (W + sw - 1) // sw, seed=7, empty_frac=0.25, lattice=lattice, scale=4.0)
        idx = stride_grid(B, H, W, 1, 1)
    _both_fcsk(oracle_backend, hip_backend, x1, x2, idx, kH, kW, K, flag, dist if not lattice else 3.5, sh, sw)


def test_fcsk_adversarial(oracle_backend, hip_backend):
    B, H, W = 2, 8, 24
    idx = stride_grid(B, H, W, 1, 1)
    # all-empty searched image: COPY broadcasts the sentinel (0,0) with mask 1
    x1 = range_image(B, H, W, 1, empty_frac=0.0)
    _both_fcsk(oracle_backend, hip_backend, x1, torch.zeros_like(x1), idx, 5, 9, 6, 3, 10.0, 1, 1)
    # every centre invalid: outputs stay the caller's zeros
    _both_fcsk(oracle_backend, hip_backend, torch.zeros_like(x1), x1, idx, 5, 9, 6, 3, 10.0, 1, 1)
    # identical points everywhere: all distances tie at the 1e-10 clamp
    same = torch.ones(B, H, W, 3)
    _both_fcsk(oracle_backend, hip_backend, same, same, idx, 3, 5, 8, 3, 1.0, 1, 1)
    # huge search radius: distance^2 >= 1e10 forces the verbatim serial path
    _both_fcsk(oracle_backend, hip_backend, x1, x1, idx, 3, 5, 8, 3, 2.0e5, 1, 1)
    # shuffled window visiting order (the reference's __main__ uses randperm)
    g = torch.Generator().manual_seed(3)
    rhw = torch.randperm(45, generator=g).int()
    xl = range_image(B, H, W, 9, lattice=True)
    _both_fcsk(oracle_backend, hip_backend, xl, xl, idx, 5, 9, 7, 3, 3.0, 1, 1, random_hw=rhw)
    _both_fcsk(oracle_backend, hip_backend, x1, x1, idx, 5, 9, 7, 2, 3.0, 1, 1, random_hw=rhw)
    # NaN / inf coordinates must not crash and must match
    xn = x1.clone(); xn[0, 2, 3] = float("nan"); xn[1, 4, 5, 0] = float("inf")
    _both_fcsk(oracle_backend, hip_backend, xn, xn, idx, 3, 5, 4, 3, 5.0, 1, 1)


def test_fcsk_fill_flag(oracle_backend, hip_backend):
    """FLAG_FILL (extension): garbage-initialised outputs come out exactly as zero-initialised ones
    without the flag — on the HIP kernel and on the oracle — for COPY and non-COPY searches."""
    B, H, W = 2, 16, 90
    x = range_image(B, H, W, seed=4, empty_frac=0.4, scale=4.0)
    idx = stride_grid(B, H, W, 1, 1)
    for flag in (2, 3):
        want = run_fcsk(oracle_backend, x, x, idx, 5, 9, 8, flag, 1.5, 1, 1)
        got_o = run_fcsk(oracle_backend, x, x, idx, 5, 9, 8, flag | 4, 1.5, 1, 1, init=77)
        got_h = run_fcsk(hip_backend, x.to(DEV), x.to(DEV), idx.to(DEV), 5, 9, 8, flag | 4, 1.5, 1, 1, init=77)
        for w_, o_, h_ in zip(want[:4], got_o[:4], got_h[:4]):
            assert torch.equal(w_, o_)
            assert torch.equal(w_, h_.cpu())


def test_fcsk_full_size_level1(oracle_backend, hip_backend):
    """BASELINE shape: 64x1800 image, 16x225 queries, 9x15 window, K=32."""
    B = 2
    x = range_image(B, 64, 1800, 11, empty_frac=0.15, scale=15.0)
    idx = stride_grid(B, 16, 225, 4, 8)
    _both_fcsk(oracle_backend, hip_backend, x, x, idx, 9, 15, 32, 3, 0.75, 1, 1)


def test_fcsk_errors(hip_backend):
    x = range_image(1, 4, 8, 0).to(DEV)
    idx = stride_grid(1, 2, 2, 1, 1).to(DEV)
    with pytest.raises(RuntimeError):
        run_fcsk(hip_backend, x, x, idx, 11, 15, 4, 3, 1.0, 1, 1)     # window > 150
    with pytest.raises(RuntimeError):
        run_fcsk(hip_backend, x.cpu(), x, idx, 3, 3, 4, 3, 1.0, 1, 1)  # host tensor


@pytest.mark.parametrize("n,m,dup,zero", [(64, 16, 0.0, 0.0), (100, 30, 0.5, 0.0), (1000, 128, 0.2, 0.1),
                                          (1024, 256, 0.0, 0.0), (8192, 2048, 0.05, 0.02),
                                          (9000, 64, 0.1, 0.0), (1, 1, 0.0, 0.0), (5, 5, 0.0, 0.0)])
def test_fps_parity(oracle_backend, hip_backend, n, m, dup, zero):
    B = 3
    pts = cloud(B, n, seed=n + m, dup_frac=dup, zero_frac=zero)
    ref = torch.zeros(B, m, dtype=torch.int32); rt = torch.full((B, n), 1e10)
    oracle_backend.furthest_point_sampling_wrapper(B, n, m, pts, rt, ref)
    got = torch.zeros(B, m, dtype=torch.int32, device=DEV); gt = torch.full((B, n), 1e10, device=DEV)
    hip_backend.furthest_point_sampling_wrapper(B, n, m, pts.to(DEV), gt, got)
    assert torch.equal(ref, got.cpu())
    assert torch.equal(rt, gt.cpu())          # running min-distances are part of the interface


def test_ball_query_parity(oracle_backend, hip_backend):
    for (N, M, ns, r) in [(500, 77, 16, 6.0), (64, 64, 4, 100.0), (300, 10, 32, 0.01), (130, 5, 200, 1e3)]:
        B = 2
        xyz = cloud(B, N, N); new_xyz = cloud(B, M, M + 1)
        new_xyz[:, : min(M, N) // 2] = xyz[:, : min(M, N) // 2]
        ref = torch.zeros(B, M, ns, dtype=torch.int32)
        oracle_backend.ball_query_wrapper(B, N, M, r, ns, new_xyz, xyz, ref)
        got = torch.zeros(B, M, ns, dtype=torch.int32, device=DEV)
        hip_backend.ball_query_wrapper(B, N, M, r, ns, new_xyz.to(DEV), xyz.to(DEV), got)
        assert torch.equal(ref, got.cpu())


def test_group_gather_parity(oracle_backend, hip_backend):
    B, C, N, P, S = 2, 37, 468, 57, 32
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, P, S), generator=g, dtype=torch.int32)
    ref = torch.empty(B, C, P, S); got = torch.empty(B, C, P, S, device=DEV)
    oracle_backend.group_points_wrapper(B, C, N, P, S, feats, idx, ref)
    hip_backend.group_points_wrapper(B, C, N, P, S, feats.to(DEV), idx.to(DEV), got)
    assert torch.equal(ref, got.cpu())
    go = torch.randn(B, C, P, S, generator=g)
    rg = torch.zeros(B, C, N); gg = torch.zeros(B, C, N, device=DEV)
    oracle_backend.group_points_grad_wrapper(B, C, N, P, S, go, idx, rg)
    hip_backend.group_points_grad_wrapper(B, C, N, P, S, go.to(DEV), idx.to(DEV), gg)
    assert torch.allclose(rg, gg.cpu(), rtol=1e-5, atol=1e-5)
    idx1 = idx[:, :, 0].contiguous()
    ref = torch.empty(B, C, P); got = torch.empty(B, C, P, device=DEV)
    oracle_backend.gather_points_wrapper(B, C, N, P, feats, idx1, ref)
    hip_backend.gather_points_wrapper(B, C, N, P, feats.to(DEV), idx1.to(DEV), got)
    assert torch.equal(ref, got.cpu())
    go = torch.randn(B, C, P, generator=g)
    rg = torch.zeros(B, C, N); gg = torch.zeros(B, C, N, device=DEV)
    oracle_backend.gather_points_grad_wrapper(B, C, N, P, go, idx1, rg)
    hip_backend.gather_points_grad_wrapper(B, C, N, P, go.to(DEV), idx1.to(DEV), gg)
    assert torch.allclose(rg, gg.cpu(), rtol=1e-5, atol=1e-5)


def test_three_nn_interpolate_parity(oracle_backend, hip_backend):
    B, N, M, C = 2, 700, 1500, 19
    unk, kn = cloud(B, N, 1), cloud(B, M, 2, dup_frac=0.2)
    rd = torch.empty(B, N, 3); ri = torch.empty(B, N, 3, dtype=torch.int32)
    oracle_backend.three_nn_wrapper(B, N, M, unk, kn, rd, ri)
    gd = torch.empty(B, N, 3, device=DEV); gi = torch.empty(B, N, 3, dtype=torch.int32, device=DEV)
    hip_backend.three_nn_wrapper(B, N, M, unk.to(DEV), kn.to(DEV), gd, gi)
    assert torch.equal(ri, gi.cpu()) and torch.equal(rd, gd.cpu())
    g = torch.Generator().manual_seed(0)
    w = torch.rand(B, N, 3, generator=g); feats = torch.randn(B, C, M, generator=g)
    ro = torch.empty(B, C, N); go = torch.empty(B, C, N, device=DEV)
    oracle_backend.three_interpolate_wrapper(B, C, M, N, feats, ri, w, ro)
    hip_backend.three_interpolate_wrapper(B, C, M, N, feats.to(DEV), gi, w.to(DEV), go)
    assert torch.equal(ro, go.cpu())
    gout = torch.randn(B, C, N, generator=g)
    rg = torch.zeros(B, C, M); gg = torch.zeros(B, C, M, device=DEV)
    oracle_backend.three_interpolate_grad_wrapper(B, C, N, M, gout, ri, w, rg)
    hip_backend.three_interpolate_grad_wrapper(B, C, N, M, gout.to(DEV), gi, w.to(DEV), gg)
    assert torch.allclose(rg, gg.cpu(), rtol=1e-5, atol=1e-5)


def test_gather_rows_parity(oracle_backend, hip_backend):
    B, H, W, C, Q = 2, 16, 225, 35, 904 * 16
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(B, H * W, C, generator=g)
    h = torch.randint(0, H, (B, Q), generator=g); w = torch.randint(0, W, (B, Q), generator=g)
    ro = torch.empty(B, Q, C); go = torch.empty(B, Q, C, device=DEV)
    oracle_backend.gather_rows(feat, h, w, W, ro)
    hip_backend.gather_rows(feat.to(DEV), h.to(DEV), w.to(DEV), W, go)
    assert torch.equal(ro, go.cpu())
    gout = torch.randn(B, Q, C, generator=g)
    rg = torch.zeros(B, H * W, C); gg = torch.zeros(B, H * W, C, device=DEV)
    oracle_backend.gather_rows_grad(gout, h, w, W, rg)
    hip_backend.gather_rows_grad(gout.to(DEV), h.to(DEV), w.to(DEV), W, gg)
    assert torch.allclose(rg, gg.cpu(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,Q,hw,hot,ld_off", [(32, 904 * 16, 3600, 0.93, None), (128, 228 * 32, 468, 0.0, None), (64, 228 * 4, 228, 0.5, None),
                                                (35, 1000, 77, 0.2, None), (3, 4099, 3600, 0.9, None), (200, 333, 50, 0.3, None),
                                                (64, 116 * 16, 228, 0.1, (72, 4)), (300, 64, 10, 0.0, None)])
def test_gather_rows_grad_hot_cell_scatter(oracle_backend, hip_backend, monkeypatch, C, Q, hw, hot, ld_off):
    """round 6: the fixed-point scatter keeps the hot cell (0,0) — where FLAG_COPY sends every empty slot and every empty query — in a
    per-block LDS accumulator row (scatter_fx_hot_kernel) instead of one global atomic per 8-row run.  Against the oracle's serial
    scatter (i2p_oracle.c gather_rows_grad, the backward of utils.py:36-60) to fp32 rounding, and BIT for BIT against the kernel of
    rounds 2-5 (I2P_SCATTER_V1=1): integer sums do not depend on the grouping.  Cases: the level-2 shape with 93 % of the rows on the
    hot cell, the kNN cost volume's 128-channel rows, windows of neighbouring queries (shared cells), odd channel counts, a strided
    source (pitch / offset), more channels than a block has threads."""
    B, W = 3, hw
    g = torch.Generator().manual_seed(C * 7 + Q)
    # neighbouring queries share cells: cell = a slowly drifting window + the hot cell with probability `hot`
    base = (torch.arange(Q) // 16 * 3) % max(hw - 20, 1)
    cell = (base.unsqueeze(0) + torch.randint(0, 20, (B, Q), generator=g)) % hw
    cell = torch.where(torch.rand(B, Q, generator=g) < hot, torch.zeros_like(cell), cell)
    h = torch.zeros(B, Q, dtype=torch.long); w = cell.long()
    ld, off = ld_off if ld_off else (C, 0)
    full = torch.randn(B, Q, ld, generator=g) * torch.logspace(-3, 2, Q).view(1, Q, 1)
    gout = full[:, :, off:off + C].contiguous()
    init = torch.randn(B, hw, C, generator=g)
    rg = init.clone()
    oracle_backend.gather_rows_grad(gout, h, w, W, rg)

    def run():
        gg = init.clone().to(DEV)
        if ld_off:
            hip_backend.gather_rows_grad_ld(full.to(DEV), ld, off, h.to(DEV), w.to(DEV), W, gg)
        else:
            hip_backend.gather_rows_grad(gout.to(DEV), h.to(DEV), w.to(DEV), W, gg)
        torch.cuda.synchronize()
        return gg.cpu()
    monkeypatch.delenv("I2P_SCATTER_V1", raising=False)
    new = run()
    monkeypatch.setenv("I2P_SCATTER_V1", "1")
    old = run()
    assert torch.equal(new, old)
    scale = float(rg.abs().max())
    assert float((new - rg).abs().max()) <= 2e-5 * scale


def test_knn_parity(oracle_backend, hip_backend):
    for (N, S, k) in [(468, 228, 32), (100, 7, 100), (2048, 64, 16)]:
        B = 2
        xyz = cloud(B, N, N + 3, dup_frac=0.1); q = cloud(B, S, S + 5)
        ri = torch.empty(B, S, k, dtype=torch.int32); gi = torch.empty(B, S, k, dtype=torch.int32, device=DEV)
        oracle_backend.knn(xyz, q, k, ri)
        hip_backend.knn(xyz.to(DEV), q.to(DEV), k, gi)
        assert torch.equal(ri, gi.cpu())


def test_knn_refill_and_nan(oracle_backend, hip_backend):
    """second-generation kNN: every near neighbour in ONE lane's slice (indices = 0 mod 64) forces the per-lane
    re-scan path; NaN points are never selected; large cloud."""
    g = torch.Generator().manual_seed(12)
    B, N, S, k = 2, 4096, 37, 32
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5) * 100
    q = (torch.rand(B, S, 3, generator=g) - 0.5) * 2
    xyz[:, ::64] = (torch.rand(B, N // 64, 3, generator=g) - 0.5) * 2           # the 64 closest points share a lane
    xyz[0, 5] = float("nan"); xyz[1, 64 * 3, 1] = float("nan")
    ri = torch.empty(B, S, k, dtype=torch.int32); gi = torch.empty(B, S, k, dtype=torch.int32, device=DEV)
    oracle_backend.knn(xyz, q, k, ri); hip_backend.knn(xyz.to(DEV), q.to(DEV), k, gi)
    assert torch.equal(ri, gi.cpu())
    assert int((ri % 64 == 0).sum()) > 0.9 * ri.numel()
    B, N, S, k = 1, 8192, 300, 32
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5) * 60; q = xyz[:, :S].clone()
    ri = torch.empty(B, S, k, dtype=torch.int32); gi = torch.empty(B, S, k, dtype=torch.int32, device=DEV)
    oracle_backend.knn(xyz, q, k, ri); hip_backend.knn(xyz.to(DEV), q.to(DEV), k, gi)
    assert torch.equal(ri, gi.cpu())


def test_pose_loss_parity(oracle_backend, hip_backend):
    g = torch.Generator().manual_seed(8)
    for B, l1 in [(8, True), (8, False), (100, True), (1, True)]:
        o3, o4 = torch.randn(B, 7, generator=g), torch.randn(B, 7, generator=g)
        qg, tg = torch.randn(B, 4, generator=g), torch.randn(B, 3, generator=g)
        wx, wq = torch.tensor([0.2]), torch.tensor([-2.5])
        r = oracle_backend.pose_loss(o3, o4, qg, tg, wx, wq, l1)
        h = hip_backend.pose_loss(*[t.to(DEV) for t in (o3, o4, qg, tg, wx, wq)], l1)
        for a, b in zip(r, h):
            assert torch.allclose(a, b.cpu(), rtol=1e-5, atol=1e-6)


def test_quat_mul_parity(oracle_backend, hip_backend):
    g = torch.Generator().manual_seed(5)
    for na, nb in [(1, 1), (1, 1440), (1440, 1), (1440, 1440)]:
        a = torch.randn(8, na, 4, generator=g); b = torch.randn(8, nb, 4, generator=g)
        for ca, cb in [(False, False), (True, False), (False, True)]:
            r = oracle_backend.quat_mul(a, b, conj_a=ca, conj_b=cb)
            h = hip_backend.quat_mul(a.to(DEV), b.to(DEV), conj_a=ca, conj_b=cb)
            assert torch.equal(r, h.cpu()), (na, nb, ca, cb)


def test_quat_unit_parity(oracle_backend, hip_backend):
    g = torch.Generator().manual_seed(6)
    for rows in (2, 8, 1000):
        q = torch.randn(rows, 4, generator=g) * torch.logspace(-3, 1, rows).unsqueeze(-1)
        q[0] = 0.0
        go = torch.randn(rows, 4, generator=g)
        for mode in (0, 1):
            r = oracle_backend.quat_unit_forward(mode, q)
            h = hip_backend.quat_unit_forward(mode, q.to(DEV))
            assert torch.allclose(r[1:], h.cpu()[1:], rtol=1e-6, atol=0), (rows, mode)
            assert torch.equal(r[0], h.cpu()[0])
            rb = oracle_backend.quat_unit_backward(mode, q, go)
            hb = hip_backend.quat_unit_backward(mode, q.to(DEV), go.to(DEV))
            assert torch.allclose(rb[1:], hb.cpu()[1:], rtol=1e-4, atol=1e-7 * float(rb[1:].abs().max())), (rows, mode)
            assert torch.isfinite(hb).all()
    with pytest.raises(Exception):
        hip_backend.quat_unit_forward(2, torch.zeros(1, 4, device=DEV))


def test_row_unitvar_parity(oracle_backend, hip_backend):
    g = torch.Generator().manual_seed(9)
    for rows, c in [(1824, 64), (3744, 128), (100, 37), (5, 256)]:
        x = torch.randn(rows, c, generator=g) * 2 + 0.5
        x[3] = -1.0
        gy = torch.randn(rows, c, generator=g)
        ry, rs = oracle_backend.row_unitvar_forward(x)
        hy, hs = hip_backend.row_unitvar_forward(x.to(DEV))
        assert torch.allclose(ry, hy.cpu(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(rs, hs.cpu(), rtol=1e-5, atol=0)
        rg = oracle_backend.row_unitvar_backward(gy, ry, rs)
        hg = hip_backend.row_unitvar_backward(gy.to(DEV), hy, hs)
        assert torch.allclose(rg, hg.cpu(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("stride,C,H,W", [(1, 16, 37, 53), (2, 16, 37, 53), (2, 32, 24, 40), (1, 64, 12, 20), (2, 128, 24, 78)])
def test_img_bn_pool_parity(oracle_backend, hip_backend, stride, C, H, W):
    """image-encoder block tail: HIP vs oracle (arg-max bit-exact away from ties, values 1e-5) and vs
    torch's batch_norm -> leaky_relu -> max_pool2d on the GPU (gradient of the conv output)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(C + stride)
    B = 3
    y = torch.randn(B, H, W, C, generator=g) * 2 + 0.3
    gam = torch.randn(C, generator=g); bet = torch.randn(C, generator=g) * 0.2; bias = torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_o, rv_o, rm_h, rv_h = rm.clone(), rv.clone(), rm.clone().to(DEV), rv.clone().to(DEV)
    ro, ra, rmi = oracle_backend.img_bn_pool_forward(y, gam, bet, 1e-5, 0.1, stride, 0.1, bias, rm_o, rv_o)
    ho, ha, hmi = hip_backend.img_bn_pool_forward(y.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5, 0.1, stride, 0.1, bias.to(DEV),
                                                  rm_h, rv_h)
    assert torch.allclose(rmi, hmi.cpu(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(ro, ho.cpu(), rtol=1e-5, atol=1e-5)
    assert (ra == ha.cpu()).float().mean() > 0.9999
    assert torch.allclose(rm_o, rm_h.cpu(), rtol=1e-5, atol=1e-6) and torch.allclose(rv_o, rv_h.cpu(), rtol=1e-5, atol=1e-6)
    gout = torch.randn(ro.shape, generator=g)
    rdy, rdg, rdb = oracle_backend.img_bn_pool_backward(gout, ra, y, rmi, gam, bet, 0.1, stride)
    hdy, hdg, hdb = hip_backend.img_bn_pool_backward(gout.to(DEV), ra.to(DEV), y.to(DEV), rmi.to(DEV), gam.to(DEV),
                                                     bet.to(DEV), 0.1, stride)
    sc = float(rdy.abs().max())
    assert torch.allclose(rdy, hdy.cpu(), rtol=1e-4, atol=1e-5 * sc)
    assert torch.allclose(rdg, hdg.cpu(), rtol=1e-4, atol=1e-4 * float(rdg.abs().max()))
    assert torch.allclose(rdb, hdb.cpu(), rtol=1e-4, atol=1e-4 * float(rdb.abs().max()))
    # torch on the GPU
    yt = y.to(DEV).permute(0, 3, 1, 2).clone().requires_grad_()
    gt, bt = gam.to(DEV).requires_grad_(), bet.to(DEV).requires_grad_()
    out_t = F.max_pool2d(F.leaky_relu(F.batch_norm(yt, None, None, gt, bt, True, 0.1, 1e-5), 0.1), 3, stride, 1)
    assert torch.allclose(out_t.permute(0, 2, 3, 1), ho, rtol=1e-4, atol=1e-4)
    (out_t * gout.to(DEV).permute(0, 3, 1, 2)).sum().backward()
    assert torch.allclose(yt.grad.permute(0, 2, 3, 1), hdy, rtol=1e-3, atol=1e-4 * sc)
    assert torch.allclose(gt.grad, hdg, rtol=1e-3, atol=1e-3 * float(rdg.abs().max()))


@pytest.mark.parametrize("ybf,obf", [(False, False), (True, True), (True, False)])
@pytest.mark.parametrize("stride,C,H,W", [(1, 16, 37, 53), (2, 16, 37, 53), (2, 32, 24, 40), (1, 64, 12, 20), (2, 128, 24, 78)])
def test_img_block_gen2_parity(oracle_backend, hip_backend, stride, C, H, W, ybf, obf):
    """second generation of the block tail (i2p_img_block_fwd / _bwd: coefficients formed in the consumers' prologues; fp32 or bf16
    storage of the conv output / the pooled output) vs the oracle on the values the kernels find in memory: fp32 storage to the
    limits of test_img_bn_pool_parity; bf16 storage — arg-max and fp32 results to the same limits, bf16 results to one bf16 ulp
    (2^-8 relative)."""
    g = torch.Generator().manual_seed(C + stride + 7)
    B = 3
    bf = torch.bfloat16
    y = torch.randn(B, H, W, C, generator=g) * 2 + 0.3
    if ybf:
        y = y.to(bf).float()
    gam = torch.randn(C, generator=g); bet = torch.randn(C, generator=g) * 0.2; bias = torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_o, rv_o, rm_h, rv_h = rm.clone(), rv.clone(), rm.clone().to(DEV), rv.clone().to(DEV)
    ro, ra, rmi = oracle_backend.img_bn_pool_forward(y, gam, bet, 1e-5, 0.1, stride, 0.1, bias, rm_o, rv_o)
    yd = y.to(DEV).to(bf) if ybf else y.to(DEV)
    ho, ha, hmi = hip_backend.img_block_forward(yd, gam.to(DEV), bet.to(DEV), 1e-5, 0.1, stride, 0.1, bias.to(DEV), rm_h, rv_h,
                                                out_bf16=obf)
    assert ho.dtype == (bf if obf else torch.float32)
    assert torch.allclose(rmi, hmi.cpu(), rtol=1e-5, atol=1e-6)
    assert (ra == ha.cpu()).float().mean() > 0.9999
    if obf:
        assert torch.allclose(ro, ho.float().cpu(), rtol=2.0 ** -8, atol=1e-5)
    else:
        assert torch.allclose(ro, ho.cpu(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(rm_o, rm_h.cpu(), rtol=1e-5, atol=1e-6) and torch.allclose(rv_o, rv_h.cpu(), rtol=1e-5, atol=1e-6)
    gout = torch.randn(ro.shape, generator=g)
    if obf:
        gout = gout.to(bf).float()
    rdy, rdg, rdb = oracle_backend.img_bn_pool_backward(gout, ra, y, rmi, gam, bet, 0.1, stride)
    gd = gout.to(DEV).to(bf) if obf else gout.to(DEV)
    hdy, hdg, hdb = hip_backend.img_block_backward(gd, ra.to(DEV), yd, rmi.to(DEV), gam.to(DEV), bet.to(DEV), 0.1, stride)
    assert hdy.dtype == yd.dtype
    sc = float(rdy.abs().max())
    if ybf:
        assert torch.allclose(rdy, hdy.float().cpu(), rtol=2.0 ** -8, atol=1e-5 * sc)
    else:
        assert torch.allclose(rdy, hdy.cpu(), rtol=1e-4, atol=1e-5 * sc)
    assert torch.allclose(rdg, hdg.cpu(), rtol=1e-4, atol=1e-4 * float(rdg.abs().max()))
    assert torch.allclose(rdb, hdb.cpu(), rtol=1e-4, atol=1e-4 * float(rdb.abs().max()))


def test_image_encoder_bf16_storage_matches_torch_bf16(hip_backend, monkeypatch):
    """the 15-block image encoder in bf16 storage mode (MIOpen bf16 convolutions + the bf16 block tails of csrc/image_block.hip,
    one multi-tensor weight cast) against plain torch ops with the same storage points (conv output and pooled output in bf16;
    BN / activation / pooling math in fp32; RF3 fp32).
    (a) block by block on identical input bits: the pooled outputs agree to 1e-4 of rms with < 1 % of the elements one bf16 ulp
        apart (fp32 rounding of the BN coefficients deciding a bf16 tie);
    (b) the whole encoder: every such flip re-rounds the 9*C outputs it feeds, so two bf16 runs drift apart to the bf16 rounding
        floor (measured 3.4e-2 of rms at RF3; bf16 against fp32 storage measures 0.15 on this random-init encoder) — limit 0.1;
        parameter gradients fp32 and finite.  The network-level limits of test_bf16_gpu.py are the contract for the bf16 mode."""
    import torch.nn.functional as F
    from i2pnet_amd import modules, ops
    from i2pnet_amd.config import I2PNetConfig as cfg
    bf = torch.bfloat16
    monkeypatch.setenv("I2P_IMG_FP32_BLOCKS", "0")          # every block in bf16 storage (the default keeps the first one fp32)
    torch.manual_seed(3)
    nets = torch.nn.Sequential(*[modules.createCNNs(*c) for c in cfg.rgb_encoder_channels]).to(DEV).to(memory_format=torch.channels_last)
    for i, net in enumerate(nets):
        net.encoder_index = i
    nets.train()
    x = torch.rand(2, 3, 96, 320, device=DEV).contiguous(memory_format=torch.channels_last)
    blocks = [m for net in nets for m in net]
    nb = len(blocks) // 4
    with torch.no_grad():
        h = x.to(bf)
        for j in range(nb):
            conv, bn, act, pool = blocks[4 * j:4 * j + 4]
            y = F.conv2d(h, conv.weight.to(bf), None, 1, 1)
            z = F.batch_norm(y.float(), None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
            ref = pool(F.leaky_relu(z, 0.1))
            ref = ref if j == nb - 1 else ref.to(bf)
            got, _, _ = hip_backend.img_block_forward(y.permute(0, 2, 3, 1).contiguous(), bn.weight, bn.bias, bn.eps, 0.1, pool.stride,
                                                      0.1, conv.bias, None, None, out_bf16=j < nb - 1)
            got = got.permute(0, 3, 1, 2)
            assert got.dtype == ref.dtype
            d = got.float() - ref.float()
            assert float(d.pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt()) < 1e-4, j
            assert j == nb - 1 or float((got != ref).float().mean()) < 1e-2, j      # (the last block's output is fp32)
            h = ref
        ref_out = h
    prev = ops.set_precision("bf16")
    try:
        out = nets(x)
    finally:
        ops.set_precision(prev)
    assert out.dtype == torch.float32
    assert float((out.detach() - ref_out).pow(2).mean().sqrt() / ref_out.pow(2).mean().sqrt()) < 0.1
    w = torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)
    (out * w).sum().backward()
    n = 0
    for k, p in nets.named_parameters():
        if p.grad is not None:
            assert p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
            n += 1
    assert n == 3 * nb                                     # conv weight, BN weight, BN bias of every block
    # the default tier: the FIRST block stays fp32 (fp32 convolution, fp32 conv output, bf16 pooled output for the second block)
    monkeypatch.setenv("I2P_IMG_FP32_BLOCKS", "1")
    with torch.no_grad():
        conv, bn, act, pool = blocks[0:4]
        z = F.batch_norm(F.conv2d(x, conv.weight, None, 1, 1), None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
        h = pool(F.leaky_relu(z, 0.1)).to(bf)
        for j in range(1, nb):
            conv, bn, act, pool = blocks[4 * j:4 * j + 4]
            z = F.batch_norm(F.conv2d(h, conv.weight.to(bf), None, 1, 1).float(), None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
            h = pool(F.leaky_relu(z, 0.1))
            h = h if j == nb - 1 else h.to(bf)
    prev = ops.set_precision("bf16")
    try:
        out1 = nets(x)
    finally:
        ops.set_precision(prev)
    assert out1.dtype == torch.float32
    assert float((out1.detach() - h).pow(2).mean().sqrt() / h.pow(2).mean().sqrt()) < 0.1


@pytest.mark.parametrize("rows,cin,cout,slope_out", [(3000, 64, 128, 0.0), (5000, 32, 32, 0.1), (2000, 36, 32, 0.0),
                                                     (4000, 128, 64, 0.1), (1500, 68, 64, 0.0)])
def test_lin_bwd_last_layer_slope_out(oracle_backend, hip_backend, rows, cin, cout, slope_out):
    """last layer of a stack: lin_backward(dL/da, slope_out) == lin_backward(dL/dy from bn_act_backward) —
    oracle vs oracle (the identity), HIP vs oracle (gen-2 kernels for power-of-two widths, gen-1 otherwise)."""
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g) * 2 + 0.5
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    gin = 1 + 0.1 * torch.randn(cin, generator=g); bin_ = 0.2 * torch.randn(cin, generator=g)
    gout = torch.randn(cout, generator=g); bout = 0.3 * torch.randn(cout, generator=g)        # negative gammas too
    ga = torch.randn(rows, cout, generator=g)

    def run(be, dev, fused):
        X, W, GA = x.to(dev), w.to(dev), ga.to(dev)
        in_coef, in_mi = be.bn_finalize(rows, be.bn_stats(X), gin.to(dev), bin_.to(dev), 1e-5)
        Y, ys = be.lin_forward(X, in_coef, 0.1, W)
        out_coef, out_mi = be.bn_finalize(rows, ys, gout.to(dev), bout.to(dev), 1e-5)
        if fused:
            ods = be.bn_act_backward_stats(GA, Y, out_mi, gout.to(dev), bout.to(dev), slope_out)
            return be.lin_backward(GA, Y, out_coef, out_mi, ods, X, in_coef, in_mi, 0.1, W, slope_out=slope_out)
        gy, _, _ = be.bn_act_backward(GA, Y, out_mi, gout.to(dev), bout.to(dev), slope_out)
        return be.lin_backward(gy, None, None, None, None, X, in_coef, in_mi, 0.1, W)
    r0 = run(oracle_backend, "cpu", False)
    r1 = run(oracle_backend, "cpu", True)
    h1 = run(hip_backend, DEV, True)
    for a, b, c, tol in ((r0[0], r1[0], h1[0], 3e-5), (r0[2], r1[2], h1[2], 2e-4)):
        sc = float(a.abs().max())
        assert float((a - b).abs().max()) <= tol * sc + 1e-6
        assert float((a - c.cpu()).abs().max()) <= tol * sc + 1e-6
    s0 = r0[1].view(32, 2, cin).sum(0); s1 = r1[1].view(32, 2, cin).sum(0); sh = h1[1].cpu().view(32, 2, cin).sum(0)
    assert torch.allclose(s0, s1, rtol=1e-4, atol=1e-3 * rows ** 0.5)
    assert torch.allclose(s0, sh, rtol=1e-4, atol=1e-3 * rows ** 0.5)


@pytest.mark.parametrize("groups,K,c", [(28800, 32, 32), (7232, 16, 64), (1824, 16, 128), (100, 8, 16), (5, 255, 64)])
def test_bn_act_maxk_parity(oracle_backend, hip_backend, groups, K, c):
    g = torch.Generator().manual_seed(groups + K)
    y = torch.randn(groups * K, c, generator=g) * 2 + 0.3
    y.view(groups, K, c)[:, K // 2:] = y.view(groups, K, c)[:, :1]                 # duplicated neighbours: ties
    coef = torch.stack([0.3 * torch.randn(c, generator=g), torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)]).contiguous()
    for slope in (0.0, 0.1):
        ro, ra = oracle_backend.bn_act_maxk_forward(y, coef, slope, K)
        ho, ha = hip_backend.bn_act_maxk_forward(y.to(DEV), coef.to(DEV), slope, K)
        assert torch.equal(ro, ho.cpu()) and torch.equal(ra, ha.cpu())
        z = (y - coef[0]) * coef[1] + coef[2]
        want = torch.where(z > 0, z, z * slope).view(groups, K, c).max(1)[0]
        assert torch.allclose(ro, want, rtol=1e-6, atol=1e-6)
    gg = torch.randn(groups, c, generator=g)
    rd = oracle_backend.unpool_k(gg, ra, K); hd = hip_backend.unpool_k(gg.to(DEV), ra.to(DEV), K)
    assert torch.equal(rd, hd.cpu())
    assert torch.equal(rd.view(groups, K, c).sum(1), gg)


def test_project_seq_parity(oracle_backend, hip_backend):
    """Cells are bit-exact for points away from a bin edge (device OCML vs host libm differ in
    the last ulp of atan2/asin); images are compared on cells whose winner agrees."""
    B, N, H, W = 2, 20000, 64, 1800
    g = torch.Generator().manual_seed(0)
    az = (torch.rand(B, N, generator=g) * 2 - 1) * 3.14159
    el = torch.deg2rad(torch.rand(B, N, generator=g) * 26.8 - 24.8)
    r = 3 + 57 * torch.rand(B, N, generator=g)
    xyz = torch.stack([r * torch.cos(el) * torch.cos(az), r * torch.cos(el) * torch.sin(az), r * torch.sin(el)], -1)
    xyz[:, N - 500:] = 0.0                        # zero padding rows like the reference loaders
    xyz = xyz.contiguous()
    f1 = torch.rand(B, N, 1, generator=g); f2 = torch.randn(B, N, 3, generator=g)
    rx, rf, rw = oracle_backend.project_seq(xyz, [f1, f2], H, W, 2.0, -24.8)
    gx, gf, gw = hip_backend.project_seq(xyz.to(DEV), [f1.to(DEV), f2.to(DEV)], H, W, 2.0, -24.8)
    same = rw == gw.cpu()
    assert same.float().mean() > 0.9995, same.float().mean()
    m = same.view(B, H, W, 1)
    assert torch.equal(rx * m, gx.cpu() * m)
    for a, b2 in zip(rf, gf):
        assert torch.equal(a * m, b2.cpu() * m)
    # self-consistency on the device: every filled cell holds its winner's rows
    wn = gw.long().clamp(min=0)
    exp = torch.gather(xyz.to(DEV), 1, wn.unsqueeze(-1).expand(-1, -1, 3)) * (gw >= 0).unsqueeze(-1)
    assert torch.equal(exp.view(B, H, W, 3), gx)


@pytest.mark.parametrize("rows,c,slope", [(1000, 16, 0.0), (12345, 64, 0.1), (853, 128, 0.1), (70000, 256, 0.0),
                                           (999, 10, 1.0), (4096, 32, 1.0)])
def test_bn_act_parity(oracle_backend, hip_backend, rows, c, slope):
    """fused BN(batch stats)+activation: HIP vs oracle (1e-5) and vs a plain torch fp32/fp64 reference."""
    g = torch.Generator().manual_seed(rows + c)
    y = torch.randn(rows, c, generator=g) * (1 + torch.arange(c).float() / c) + 50.0 * torch.randn(c, generator=g)
    gamma = 1 + 0.1 * torch.randn(c, generator=g); beta = 0.1 * torch.randn(c, generator=g)
    go = torch.randn(rows, c, generator=g)
    ro, rmi = oracle_backend.bn_act_forward(y, gamma, beta, 1e-5, slope)
    ho, hmi = hip_backend.bn_act_forward(y.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, slope)
    assert torch.allclose(rmi, hmi.cpu(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(ro, ho.cpu(), rtol=1e-5, atol=1e-5)
    rdy, rdg, rdb = oracle_backend.bn_act_backward(go, y, rmi, gamma, beta, slope)
    hdy, hdg, hdb = hip_backend.bn_act_backward(go.to(DEV), y.to(DEV), hmi, gamma.to(DEV), beta.to(DEV), slope)
    assert torch.allclose(rdy, hdy.cpu(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rdg, hdg.cpu(), rtol=1e-4, atol=1e-3) and torch.allclose(rdb, hdb.cpu(), rtol=1e-4, atol=1e-3)
    # independent reference: torch autograd in float64
    yd = y.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    var, mean = torch.var_mean(yd, dim=0, unbiased=False)
    z = (yd - mean) * torch.rsqrt(var + 1e-5) * gd + bd
    a = torch.where(z > 0, z, z * slope)
    a.backward(go.double())
    assert torch.allclose(ho.cpu().double(), a.detach(), rtol=1e-4, atol=1e-4)
    # an element whose pre-activation is within fp32 rounding of 0 takes the other activation branch
    # in fp64: allow a 1e-5 fraction of such elements
    bad = ~torch.isclose(hdy.cpu().double(), yd.grad, rtol=1e-3, atol=1e-4)
    assert bad.float().mean() < 1e-5, bad.float().mean()
    # (one activation-branch flip moves a channel's dgamma/dbeta by |go * xhat| ~ a few units)
    assert torch.allclose(hdg.cpu().double(), gd.grad, rtol=1e-3, atol=1e-2 + 1e-4 * rows)
    assert torch.allclose(hdb.cpu().double(), bd.grad, rtol=1e-3, atol=1e-2 + 1e-4 * rows)


@pytest.mark.parametrize("rows,cin,cout,with_bn", [(1000, 128, 128, True), (4133, 64, 64, True), (777, 10, 16, False),
                                                   (5000, 35, 32, False), (2048, 128, 64, True), (300, 67, 128, False),
                                                   (12800, 16, 32, True), (999, 131, 128, False), (640, 128, 256, True),
                                                   (5000, 36, 32, False), (3000, 68, 64, False), (2000, 12, 16, False),
                                                   (1000, 256, 128, True), (928, 320, 128, True), (1824, 64, 192, True),
                                                   (3000, 128, 320, False), (70, 192, 64, True), (14848, 128, 256, True),
                                                   (14848, 132, 128, False), (58368, 136, 128, False), (1000, 160, 64, True)])
def test_lin_fwd_parity(oracle_backend, hip_backend, rows, cin, cout, with_bn):
    """fused (BN+act on load) x W^T + output statistics: HIP MFMA kernel vs oracle (k-ordered fmaf
    chain) and vs torch fp64.  Asymmetric W catches transposed fragments."""
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g) * 2 + 1
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5 + torch.arange(cout).view(-1, 1) * 0.01
    coef = None
    if with_bn:
        coef = torch.stack([torch.randn(cin, generator=g), 0.5 + torch.rand(cin, generator=g),
                            0.1 * torch.randn(cin, generator=g)]).contiguous()
    ry, rs = oracle_backend.lin_forward(x, coef, 0.1, w)
    hy, hs = hip_backend.lin_forward(x.to(DEV), None if coef is None else coef.to(DEV), 0.1, w.to(DEV))
    scale = float(ry.abs().max())
    assert float((ry - hy.cpu()).abs().max()) <= 2e-6 * scale + 1e-6
    rsum = rs.view(32, 2, cout).sum(0); hsum = hs.cpu().view(32, 2, cout).sum(0)
    assert torch.allclose(rsum, hsum, rtol=1e-6, atol=1e-4 * rows ** 0.5)
    xd = x.double()
    if with_bn:
        z = (xd - coef[0].double()) * coef[1].double() + coef[2].double()
        xd = torch.where(z > 0, z, z * 0.1)
    yd = xd @ w.double().t()
    assert float((yd - hy.cpu().double()).abs().max()) <= 1e-5 * float(yd.abs().max())
    gam = 1 + 0.1 * torch.randn(cout, generator=g); bet = 0.1 * torch.randn(cout, generator=g)
    rc, rmi = oracle_backend.bn_finalize(rows, rs, gam, bet, 1e-5)
    hc, hmi = hip_backend.bn_finalize(rows, hs, gam.to(DEV), bet.to(DEV), 1e-5)
    assert torch.allclose(rc, hc.cpu(), rtol=1e-5, atol=1e-6) and torch.allclose(rmi, hmi.cpu(), rtol=1e-5, atol=1e-6)
    var, mean = torch.var_mean(yd, dim=0, unbiased=False)
    assert torch.allclose(hc.cpu()[0].double(), mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(hc.cpu()[1].double(), torch.rsqrt(var + 1e-5) * gam.double(), rtol=1e-4)


@pytest.mark.parametrize("rows,cin,cout,in_bn,out_bn", [(1000, 128, 128, True, True), (4133, 64, 64, True, True),
                                                        (777, 12, 16, False, True), (2048, 128, 64, True, False),
                                                        (300, 68, 128, False, True), (12800, 16, 32, True, True),
                                                        (999, 132, 128, False, True), (5000, 64, 128, True, True),
                                                        (1000, 128, 256, True, True), (928, 320, 128, True, True),
                                                        (1824, 64, 192, True, True), (2000, 256, 128, True, True),
                                                        (777, 192, 64, False, True), (500, 256, 256, True, False),
                                                        (14848, 132, 128, False, True), (58368, 136, 128, False, True),
                                                        (1000, 160, 64, True, True)])
def test_lin_bwd_parity(oracle_backend, hip_backend, rows, cin, cout, in_bn, out_bn):
    """fused layer backward (BN-backward on load, wgrad + dgrad on MFMA, activation derivative and
    statistics in the epilogue): HIP vs oracle, and the whole layer vs torch autograd in fp64."""
    g = torch.Generator().manual_seed(rows + cin + cout)
    x = torch.randn(rows, cin, generator=g) * 2 + 0.5
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5 + torch.arange(cout).view(-1, 1) * 0.003
    gin = 1 + 0.1 * torch.randn(cin, generator=g); bin_ = 0.2 * torch.randn(cin, generator=g)
    gout = 1 + 0.1 * torch.randn(cout, generator=g); bout = 0.2 * torch.randn(cout, generator=g)
    slope = 0.1
    # ---- reference in fp64 via autograd ------------------------------------------------------------
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    if in_bn:
        v, m = torch.var_mean(xd, 0, unbiased=False)
        zin = (xd - m) * torch.rsqrt(v + 1e-5) * gin.double() + bin_.double()
        xa = torch.where(zin > 0, zin, zin * slope)
    else:
        xa = xd
    yd = xa @ wd.t()
    if out_bn:
        v2, m2 = torch.var_mean(yd, 0, unbiased=False)
        zo = (yd - m2) * torch.rsqrt(v2 + 1e-5) * gout.double() + bout.double()
    else:
        zo = yd
    gz = torch.randn(rows, cout, generator=g)
    zo.backward(gz.double())
    # ---- operator inputs (statistics from the forward kernels of each backend) ----------------------
    def run(be, dev):
        X, W, GZ = x.to(dev), w.to(dev), gz.to(dev)
        in_coef = in_mi = None
        if in_bn:
            s = torch.zeros(32 * 2 * cin, dtype=torch.float64, device=dev)
            be._call("i2p_bn_stats", rows, cin, be._p(X, torch.float32, "x"), be._p(s, torch.float64, "s"), stream=be._stream())
            in_coef, in_mi = be.bn_finalize(rows, s, gin.to(dev), bin_.to(dev), 1e-5)
        Y, ys = be.lin_forward(X, in_coef, slope, W)
        out_coef = out_mi = ods = None
        if out_bn:
            out_coef, out_mi = be.bn_finalize(rows, ys, gout.to(dev), bout.to(dev), 1e-5)
            xh = (Y - out_mi[:cout]) * out_mi[cout:]
            ods = torch.zeros(32, 2, cout, dtype=torch.float64, device=dev)
            ods[0, 0] = GZ.double().sum(0); ods[0, 1] = (GZ.double() * xh.double()).sum(0)
            ods = ods.reshape(-1).contiguous()
        gzin, ids, dw = be.lin_backward(GZ, Y if out_bn else None, out_coef, out_mi, ods, X, in_coef, in_mi, slope, W)
        return gzin, ids, dw, in_coef, in_mi
    rg, rids, rdw, _, _ = run(oracle_backend, "cpu")
    hg, hids, hdw, in_coef, in_mi = run(hip_backend, DEV)
    sc = float(rdw.abs().max())
    assert float((rdw - hdw.cpu()).abs().max()) <= 2e-4 * sc + 1e-5
    assert float((rg - hg.cpu()).abs().max()) <= 2e-5 * float(rg.abs().max()) + 1e-6
    if in_bn:
        a = rids.view(32, 2, cin).sum(0); b = hids.cpu().view(32, 2, cin).sum(0)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * rows ** 0.5)
    # ---- against fp64 autograd: dW directly; dX through the remaining BN-backward of the input BN ----
    assert float((hdw.cpu().double() - wd.grad).abs().max()) <= 1e-4 * float(wd.grad.abs().max()) + 1e-6
    if in_bn:
        dsum = hids.cpu().view(32, 2, cin).sum(0)
        m1 = dsum[0] / rows; m2_ = dsum[1] / rows
        xh = (x.double() - in_mi.cpu()[:cin].double()) * in_mi.cpu()[cin:].double()
        gx = in_coef.cpu()[1].double() * (hg.cpu().double() - m1 - xh * m2_)
    else:
        gx = hg.cpu().double()
    bad = ~torch.isclose(gx, xd.grad, rtol=2e-3, atol=2e-4 * float(xd.grad.abs().max()))
    assert bad.float().mean() < 1e-4, bad.float().mean()


@pytest.mark.parametrize("B,N,M,C,Co", [(2, 57, 80, 128, 128), (1, 228, 468, 128, 128), (3, 10, 70, 64, 32)])
def test_pair_lin_parity(oracle_backend, hip_backend, B, N, M, C, Co):
    """factored first cost-volume layer (bilinear point x pixel product on MFMA): HIP vs oracle and
    vs the materialised torch formulation in fp64 (forward and all five gradients)."""
    g_ = torch.Generator().manual_seed(B * N + M)
    f = torch.randn(B, N, C, generator=g_); g = torch.randn(B, M, C, generator=g_)
    bn = torch.randn(B, N, Co, generator=g_); bk = torch.randn(B, M, Co, generator=g_)
    w = torch.randn(Co, C, generator=g_) / C ** 0.5 + torch.arange(Co).view(-1, 1) * 0.002
    ry, rs = oracle_backend.pair_lin_forward(f, g, bn, bk, w)
    hy, hs = hip_backend.pair_lin_forward(f.to(DEV), g.to(DEV), bn.to(DEV), bk.to(DEV), w.to(DEV))
    assert float((ry - hy.cpu()).abs().max()) <= 3e-6 * float(ry.abs().max()) + 1e-6
    assert torch.allclose(rs.view(32, 2, Co).sum(0), hs.cpu().view(32, 2, Co).sum(0), rtol=1e-6, atol=1e-3)
    fd, gd, bnd, bkd, wd = [t.double().requires_grad_(True) for t in (f, g, bn, bk, w)]
    yd = torch.einsum("bnc,bkc,oc->bnko", fd, gd, wd) + bnd[:, :, None] + bkd[:, None]
    assert float((yd.reshape(-1, Co) - hy.cpu().double()).abs().max()) <= 1e-5 * float(yd.abs().max())
    gy = torch.randn(B * N * M, Co, generator=g_)
    yd.backward(gy.view(B, N, M, Co).double())
    r = oracle_backend.pair_lin_backward(gy, f, g, w)
    h = hip_backend.pair_lin_backward(gy.to(DEV), f.to(DEV), g.to(DEV), w.to(DEV))
    for name, rr, hh, ref in zip(["d_f", "d_g", "d_bn", "d_bk", "dw"], r, h, [fd.grad, gd.grad, bnd.grad, bkd.grad, wd.grad]):
        sc = float(ref.abs().max())
        assert float((rr - hh.cpu()).abs().max()) <= 3e-4 * sc, name
        assert float((hh.cpu().double() - ref).abs().max()) <= 3e-4 * sc, name


def _pair_bn_case(B, N, M, C, Co, seed):
    g_ = torch.Generator().manual_seed(seed)
    f = torch.randn(B, N, C, generator=g_); g = torch.randn(B, M, C, generator=g_)
    bn = torch.randn(B, N, Co, generator=g_); bk = torch.randn(B, M, Co, generator=g_)
    w = torch.randn(Co, C, generator=g_) / C ** 0.5
    gam = torch.randn(Co, generator=g_); bet = 0.2 * torch.randn(Co, generator=g_)
    gz = torch.randn(B * N * M, Co, generator=g_)
    return f, g, bn, bk, w, gam, bet, gz


@pytest.mark.parametrize("B,N,M,C,Co", [(2, 23, 70, 128, 128), (1, 9, 130, 64, 64), (3, 17, 64, 128, 64)])
def test_pair_lin_bwd_bn_on_load(oracle_backend, hip_backend, B, N, M, C, Co):
    """pair layer backward with the BN behind it formed on load: HIP vs oracle, and both vs the two-step
    path (bn_act_backward with slope 1 -> dL/dy, then the plain pair backward)."""
    f, g, bn, bk, w, gam, bet, gz = _pair_bn_case(B, N, M, C, Co, 31)
    rows = B * N * M

    def run(be, dev, on_load):
        t = lambda v: v.to(dev)
        y, st = be.pair_lin_forward(t(f), t(g), t(bn), t(bk), t(w))
        coef, mi = be.bn_finalize(rows, st, t(gam), t(bet), 1e-5)
        ds = be.bn_act_backward_stats(t(gz), y, mi, t(gam), t(bet), 1.0)
        if on_load:
            return be.pair_lin_backward(t(gz), t(f), t(g), t(w), y=y, out_coef=coef, out_mi=mi, out_dsums=ds)
        dy, _, _ = be.bn_act_backward(t(gz), y, mi, t(gam), t(bet), 1.0)
        return be.pair_lin_backward(dy, t(f), t(g), t(w))
    r0 = run(oracle_backend, "cpu", False); r1 = run(oracle_backend, "cpu", True); h1 = run(hip_backend, DEV, True)
    for name, a, b_, c_ in zip(["d_f", "d_g", "d_bn", "d_bk", "dw"], r0, r1, h1):
        sc = float(a.abs().max())
        assert float((a - b_).abs().max()) <= 3e-4 * sc, name
        assert float((a - c_.cpu()).abs().max()) <= 3e-4 * sc, name


@pytest.mark.parametrize("B,N,M,C", [(2, 23, 70, 64), (8, 228, 468, 64), (1, 5, 9, 128)])
def test_pair_bias_bn_bwd_parity(oracle_backend, hip_backend, B, N, M, C):
    """gradient of the position-encoding factors (closed-form BN backward of an outer sum): HIP vs the literal
    oracle evaluation; small case also vs torch autograd in fp64."""
    g_ = torch.Generator().manual_seed(B + N)
    en = torch.randn(B, N, C, generator=g_) * 2; ek = torch.randn(B, M, C, generator=g_) + 0.5
    gam = torch.randn(C, generator=g_); bet = 0.1 * torch.randn(C, generator=g_)
    rows = B * N * M
    gz = torch.randn(rows, C, generator=g_)
    ye = (en[:, :, None] + ek[:, None]).reshape(rows, C).contiguous()
    if rows > 100000:       # the literal oracle loop is slow: compare with the same closed form in fp64 torch
        coef, mi = hip_backend.bn_finalize(rows, hip_backend.bn_stats(ye.to(DEV)), gam.to(DEV), bet.to(DEV), 1e-5)
        ds = hip_backend.bn_act_backward_stats(gz.to(DEV), ye.to(DEV), mi, gam.to(DEV), bet.to(DEV), 1.0)
        hn, hk = hip_backend.pair_bias_bn_backward(B, N, M, gz.to(DEV), en.to(DEV), ek.to(DEV), ds, coef, mi)
        yd = ye.double(); m = yd.mean(0); v = yd.var(0, unbiased=False); is_ = (v + 1e-5).rsqrt()
        xh = (yd - m) * is_; gd = gz.double()
        dy = (is_ * gam.double()) * (gd - gd.mean(0) - xh * (gd * xh).mean(0))
        dy = dy.view(B, N, M, C)
        assert torch.allclose(hn.cpu().double(), dy.sum(2), rtol=1e-3, atol=2e-3 * float(dy.sum(2).abs().max()))
        assert torch.allclose(hk.cpu().double(), dy.sum(1), rtol=1e-3, atol=2e-3 * float(dy.sum(1).abs().max()))
        return

    def run(be, dev):
        t = lambda v_: v_.to(dev)
        coef, mi = be.bn_finalize(rows, be.bn_stats(t(ye)), t(gam), t(bet), 1e-5)
        ds = be.bn_act_backward_stats(t(gz), t(ye), mi, t(gam), t(bet), 1.0)
        return be.pair_bias_bn_backward(B, N, M, t(gz), t(en), t(ek), ds, coef, mi)
    rn, rk = run(oracle_backend, "cpu"); hn, hk = run(hip_backend, DEV)
    assert float((rn - hn.cpu()).abs().max()) <= 1e-3 * float(rn.abs().max())
    assert float((rk - hk.cpu()).abs().max()) <= 1e-3 * float(rk.abs().max())
    end, ekd = en.double().requires_grad_(), ek.double().requires_grad_()
    yed = (end[:, :, None] + ekd[:, None]).reshape(rows, C)
    z = torch.nn.functional.batch_norm(yed, None, None, gam.double(), bet.double(), True, 0.0, 1e-5)
    z.backward(gz.double())
    assert float((rn.double() - end.grad).abs().max()) <= 1e-3 * float(end.grad.abs().max())
    assert float((rk.double() - ekd.grad).abs().max()) <= 1e-3 * float(ekd.grad.abs().max())


def test_two_source_backward_repeated(hip_backend):
    """regression: the two-source wgrad kernel used to read 3*cin coefficients from the 3*split_c-long array of
    the first source (a memory fault whenever that small tensor ended a mapped segment).  Fresh allocations in
    a loop make the layout vary."""
    be = hip_backend
    B, N, M, Ca, Cb, Co = 2, 23, 70, 64, 64, 128
    rows = B * N * M
    for it in range(12):
        xa = torch.randn(rows, Ca, device=DEV); xb = torch.randn(rows, Cb, device=DEV)
        w = torch.randn(Co, Ca + Cb, device=DEV) / 11
        one = lambda c: (torch.ones(c, device=DEV), torch.zeros(c, device=DEV))
        ca, ma = be.bn_finalize(rows, be.bn_stats(xa), *one(Ca), 1e-5)
        cb, mb = be.bn_finalize(rows, be.bn_stats(xb), *one(Cb), 1e-5)
        y, st = be.lin_forward_2src(xa, ca, 0.1, xb, cb, 0.1, w)
        co, mo = be.bn_finalize(rows, st, *one(Co), 1e-5)
        gz = torch.randn(rows, Co, device=DEV)
        ods = torch.zeros(32 * 2 * Co, dtype=torch.float64, device=DEV)
        out = be.lin_backward_2src(gz, y, co, mo, ods, xa, ca, ma, 0.1, xb, cb, mb, 0.1, torch.randn(rows, Cb, device=DEV), w)
        torch.cuda.synchronize()
        assert all(torch.isfinite(t).all() for t in out)


def test_cv_tail_ops_parity(oracle_backend, hip_backend):
    """two-source fused layer (fwd/bwd) and softmax-weighted sum (fwd/bwd): HIP vs oracle."""
    B, N, M, Ca, Cb, Co = 2, 23, 70, 64, 64, 128
    rows = B * N * M
    g = torch.Generator().manual_seed(5)
    xa = torch.randn(rows, Ca, generator=g); xb = torch.randn(rows, Cb, generator=g) * 2 + 0.3
    w = torch.randn(Co, Ca + Cb, generator=g) / 11 + torch.arange(Co).view(-1, 1) * 0.002
    mk = lambda c: (1 + 0.1 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g))
    (ga, ba), (gb, bb), (go, bo) = mk(Ca), mk(Cb), mk(Co)

    def run(be, dev):
        t = lambda v: v.to(dev)
        ca, ma = be.bn_finalize(rows, be.bn_stats(t(xa)), t(ga), t(ba), 1e-5)
        cb, mb = be.bn_finalize(rows, be.bn_stats(t(xb)), t(gb), t(bb), 1e-5)
        y, st = be.lin_forward_2src(t(xa), ca, 0.1, t(xb), cb, 0.1, t(w))
        co, mo = be.bn_finalize(rows, st, t(go), t(bo), 1e-5)
        gz = torch.randn(rows, Co, generator=torch.Generator().manual_seed(9)).to(dev)
        xh = (y - mo[:Co]) * mo[Co:]
        ods = torch.zeros(32, 2, Co, dtype=torch.float64, device=dev)
        ods[0, 0] = gz.double().sum(0); ods[0, 1] = (gz.double() * xh.double()).sum(0)
        e_add = torch.randn(rows, Cb, generator=torch.Generator().manual_seed(10)).to(dev)
        bw = be.lin_backward_2src(gz, y, co, mo, ods.reshape(-1).contiguous(), t(xa), ca, ma, 0.1, t(xb), cb, mb, 0.1, e_add, t(w))
        # softmax-weighted sum on (y[:, :64] as logits, xb as values)
        y5 = y[:, :64].contiguous(); c5 = torch.stack([co[0, :64], co[1, :64], co[2, :64]]).contiguous()
        m5 = torch.cat([mo[:64], mo[Co:Co + 64]]).contiguous()
        out, ms = be.cv_softmax_wsum_forward(B, N, M, y5, c5, 0.1, t(xb), cb, 0.1)
        gout = torch.randn(B, N, 64, generator=torch.Generator().manual_seed(11)).to(dev)
        sb = be.cv_softmax_wsum_backward(B, N, M, gout, out, ms, y5, c5, m5, 0.1, t(xb), cb, 0.1)
        return [y, *bw, out, *sb]

    ref = run(oracle_backend, "cpu"); got = run(hip_backend, DEV)
    names = ["y", "gz_a", "ds_a", "gz_b", "ds_b", "dw", "out", "gz5", "ds5", "ga3"]
    for n_, r, h in zip(names, ref, got):
        h = h.cpu()
        if r.dtype == torch.float64:
            c = r.numel() // 64
            r = r.view(32, 2, -1).sum(0); h = h.view(32, 2, -1).sum(0)
            assert torch.allclose(r, h, rtol=1e-4, atol=1e-2), n_
        else:
            assert float((r - h).abs().max()) <= 3e-4 * float(r.abs().max()) + 1e-6, n_
    # softmax-weighted sum against plain torch
    y5 = ref[0][:, :64]
    h5 = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(y5, None, None, go[:64], bo[:64], True, 0.0, 1e-5), 0.1)
    h3 = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(xb, None, None, gb, bb, True, 0.0, 1e-5), 0.1)
    want = (torch.softmax(h5.view(B, N, M, 64), 2) * h3.view(B, N, M, 64)).sum(2)
    assert torch.allclose(got[6].cpu(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("lattice", [False, True])
def test_sa_l1_group_matches_unfused_path(hip_backend, lattice):
    """Fused level-1 grouping (selection + gather + feature build, window strip in LDS) against the unfused operator
    chain (fused_conv_select_k -> gather_rows -> torch feature build), which is itself bit-exact against the oracle:
    identical neighbour choices incl. ties (lattice image), empty centres, empty windows and column wrap."""
    from i2pnet_amd import modules, projectpn as P
    from helpers import range_image
    B, H, W = 2, 64, 1800
    raw = range_image(B, H, W, seed=11, empty_frac=0.5, lattice=lattice).cuda()
    rot = torch.tensor([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]], device="cuda")
    sel = (raw @ rot.t() + torch.tensor([0.3, -0.2, 0.1], device="cuda")) * (raw != 0).any(-1, keepdim=True)
    dist = 3.0 if lattice else 0.75
    net = modules.ProjectPointNet(H, W, 16, 225, 4, 8, [9, 15], 32, dist, 12, [16, 16, 32], use_trans=True).cuda()
    outs = {}
    for fused in (True, False):
        modules.USE_FUSED_GROUP = fused
        try:
            captured = {}
            orig = net._mlp_max
            net._mlp_max = lambda x, B_, _c=captured: _c.setdefault("feat", x.clone()) if False else (_c.__setitem__("feat", x.clone()) or orig(x, B_))
            net.forward_center(raw, sel, None, raw_feat_point=True)
            outs[fused] = captured["feat"]
        finally:
            modules.USE_FUSED_GROUP = True
            net._mlp_max = orig
    a, b = outs[True].reshape(B, 3600, 32, 12), outs[False].reshape(B, 3600, 32, 12)
    assert torch.equal(a[..., :9], b[..., :9])                       # differences, centres, neighbours: pure copies / one subtraction
    assert torch.allclose(a[..., 9], b[..., 9], rtol=1e-6, atol=0)    # |d|: torch.norm's reduction may round differently
    assert float(a[..., 10:].abs().max()) == 0.0
    assert float(a[..., :3].abs().max()) > 0


def _sa_l1_oracle_chain(ob, sel, raw, out_h, out_w, sh, sw, kH, kW, K, dist):
    """what i2p_sa_l1_group fuses, on the CPU oracle: fused_conv_select_k_cpu (FLAG_SHIFT|FLAG_COPY, zero-initialised
    outputs) -> gather_rows_cpu of the raw image -> the 10-channel feature rows of PPBackbone_center.py:177-187"""
    B, H, W, _ = sel.shape
    N = out_h * out_w
    idx = stride_grid(B, out_h, out_w, sh, sw)
    sb, shh, sww, m, _, _ = run_fcsk(ob, sel, sel, idx, kH, kW, K, 3, dist, 1, 1)
    nb = torch.empty(B, N * K, 3)
    ob.gather_rows(raw.reshape(B, H * W, 3).contiguous(), shh.reshape(B, N * K).contiguous(), sww.reshape(B, N * K).contiguous(), W, nb)
    nb = nb.view(B, N, K, 3)
    c_sel = sel[:, ::sh, ::sw][:, :out_h, :out_w].reshape(B, N, 1, 3)
    c_raw = raw[:, ::sh, ::sw][:, :out_h, :out_w].reshape(B, N, 1, 3)
    d = nb - c_raw
    dn = torch.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]).unsqueeze(-1)
    return torch.cat([d, c_sel.expand(-1, -1, K, -1), nb, dn, torch.zeros(B, N, K, 2)], -1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["scan", "dense", "lattice", "nuscenes", "scalar_stage"])
def test_sa_l1_group_full_size_vs_oracle_chain(hip_backend, oracle_backend, case, monkeypatch):
    """i2p_sa_l1_group at the level-1 size of the network (64 x 1800 -> 16 x 225 queries, 9 x 15 window, K = 32) against the
    ORACLE chain it replaces (fused_conv_go.cu:49-238 + utils.py:36-60 + PPBackbone_center.py:177-187), not against another
    HIP path: sparse scan (7 % of the cells occupied: most centres empty), dense image, lattice (ties everywhere -> the
    serial redo), the nuScenes level-1 shape (21 rows, row stride 2) and the scalar staging path (no 16-byte alignment)."""
    B, H, W, out_h, sh = 2, 64, 1800, 16, 4
    dist = 0.75
    if case == "nuscenes":
        H, out_h, sh = 21, 11, 2
    empty = {"scan": 0.93, "dense": 0.1, "lattice": 0.5, "nuscenes": 0.5, "scalar_stage": 0.6}[case]
    raw = range_image(B, H, W, seed=21, empty_frac=empty, lattice=(case == "lattice"))
    if case == "lattice":
        dist = 3.0
    rot = torch.tensor([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    sel = ((raw @ rot.t() + torch.tensor([0.3, -0.2, 0.1])) * (raw != 0).any(-1, keepdim=True)).contiguous()
    if case == "scalar_stage":
        monkeypatch.setenv("I2P_SA_SCALAR_STAGE", "1")
    got = hip_backend.sa_l1_group(sel.to(DEV), raw.to(DEV), out_h, 225, sh, 8, 9, 15, 32, dist).cpu().view(B, out_h * 225, 32, 12)
    want = _sa_l1_oracle_chain(oracle_backend, sel, raw, out_h, 225, sh, 8, 9, 15, 32, dist)
    assert torch.equal(got[..., :9], want[..., :9])
    assert torch.allclose(got[..., 9], want[..., 9], rtol=2e-7, atol=0)
    assert float(got[..., 10:].abs().max()) == 0.0
    live = (sel[:, ::sh, ::8][:, :out_h, :225] != 0).any(-1).float().mean()
    assert (0.02 < live < 0.2) if case == "scan" else live > 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("xyz_first", [True, False])
def test_sa_rows_matches_gather_cat(hip_backend, xyz_first):
    """i2p_sa_rows (grouped MLP input rows in one launch, levels 2-4 / up-convolutions) against the chain it replaces:
    gather_torch of the coordinate image - centre, gather_torch of the feature image, zero-padded cat
    (PPBackbone_center.py:94-129, :236-262): identical values; feature gradient from the strided fixed-point scatter."""
    from i2pnet_amd import modules, projectpn as P
    B, H, W, C, N, K = 2, 16, 57, 35, 40, 8
    g = torch.Generator().manual_seed(3)
    xyz = torch.randn(B, H, W, 3, generator=g).to(DEV); centre = torch.randn(B, N, 3, generator=g).to(DEV)
    feat = torch.randn(B, H, W, C, generator=g).to(DEV).requires_grad_()
    h = torch.randint(0, H, (B, N, K), generator=g).to(DEV); w = torch.randint(0, W, (B, N, K), generator=g).to(DEV)
    h[:, :, -2:] = 0; w[:, :, -2:] = 0                                     # the reference's empty slots: cell (0, 0), many times
    rows = P.sa_rows(xyz, centre, feat, h, w, K, W, xyz_first=xyz_first)
    gout = torch.randn(rows.shape, generator=torch.Generator().manual_seed(4)).to(DEV)
    (ga,) = torch.autograd.grad(rows, feat, gout)
    d = P.gather_torch(xyz, None, h, w, B, H, W) - centre.view(B, N, 1, 3)
    f2 = P.gather_torch(feat, None, h, w, B, H, W)
    ref = modules.cat_padded([d, f2] if xyz_first else [f2, d], pow2=True)
    assert rows.shape == ref.shape and torch.equal(rows, ref)
    (gb,) = torch.autograd.grad(ref, feat, gout)
    assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-6 * float(gb.abs().max()))
    (ga2,) = torch.autograd.grad(P.sa_rows(xyz, centre, feat, h, w, K, W, xyz_first=xyz_first), feat, gout)
    assert torch.equal(ga, ga2)                                            # order-independent accumulation


@pytest.mark.gpu
@pytest.mark.parametrize("rows,m,n", [(14848, 256, 128), (3744, 3, 64), (1824, 64, 192), (1000, 5, 7), (33, 70, 130), (928, 128, 320)])
def test_gemm_tn_matches_fp64(hip_backend, rows, m, n):
    """i2p_gemm_tn (weight gradient of the plain linear layers: a^T b with the rows cut over the grid) against fp64, ragged
    tiles and a row count that is not a multiple of the stage; two runs agree bit for bit (fixed summation order)"""
    g = torch.Generator().manual_seed(rows + m)
    a = torch.randn(rows, m, generator=g).cuda(); b = torch.randn(rows, n, generator=g).cuda()
    out = hip_backend.gemm_tn(a, b)
    want = a.double().t() @ b.double()
    assert out.shape == (m, n)
    assert float((out.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-4
    assert torch.equal(out, hip_backend.gemm_tn(a, b))


@pytest.mark.gpu
def test_linear_weight_grad_on_gemm_tn(hip_backend):
    """fused.linear == F.linear in value, input gradient and weight gradient (the latter on i2p_gemm_tn), strided weight view"""
    from i2pnet_amd import fused
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 936, 200, generator=g).cuda().requires_grad_(True)
    Wfull = torch.randn(96, 203, generator=g).cuda().requires_grad_(True)
    go = torch.randn(4, 936, 96, generator=g).cuda()
    outs = []
    for fn in (fused.linear, F.linear):
        w = torch.split(Wfull, [3, 200], dim=1)[1]
        y = fn(x, w)
        gx, gw = torch.autograd.grad(y, (x, Wfull), go)
        outs.append((y, gx, gw))
    (y0, gx0, gw0), (y1, gx1, gw1) = outs
    assert torch.allclose(y0, y1, rtol=1e-5, atol=1e-5) and torch.allclose(gx0, gx1, rtol=1e-5, atol=1e-5)
    assert float((gw0 - gw1).abs().max()) <= 2e-5 * float(gw1.abs().max()) + 1e-4
    assert float(gw0[:, :3].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,C", [(2, 23, 70, 64), (1, 228, 468, 64), (3, 5, 9, 128)])
def test_outer_sum_matches_broadcast_add(hip_backend, B, N, M, C):
    """i2p_outer_sum (position encoding of all point x pixel pairs + its BN statistics in one pass) against the broadcast add
    and fp64 sums it replaces: values bit-exact, sums to fp64 rounding"""
    g = torch.Generator().manual_seed(B * N + M)
    en = torch.randn(B, N, C, generator=g).cuda(); ek = torch.randn(B, M, C, generator=g).cuda()
    ye, sums = hip_backend.outer_sum(en, ek)
    want = (en.unsqueeze(2) + ek.unsqueeze(1)).reshape(B * N * M, C)
    assert torch.equal(ye, want)
    s = sums.view(-1, 2, C).sum(0)
    assert torch.allclose(s[0], want.double().sum(0), rtol=1e-9, atol=1e-6)
    assert torch.allclose(s[1], want.double().square().sum(0), rtol=1e-9, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,K,C", [(2, 57, 4, 64), (8, 228, 4, 64), (1, 9, 16, 128)])
def test_softmax_wsum_k_matches_torch(hip_backend, B, N, K, C):
    """pc-stage tail of the cost volume (softmax over the K neighbours, weighted sum of their features) on the
    cv_softmax_wsum kernels against softmax / mul / sum, values and both gradients; fully masked groups (-1e10 logits) included"""
    from i2pnet_amd import fused
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N + K)
    logit = torch.randn(B, N, K, C, generator=g).cuda()
    logit[0, 0] = -1e10                                        # a point without valid neighbours: uniform weights
    logit[0, 1, 1:] = -1e10
    value = torch.randn(B, N, K, C, generator=g).cuda()
    go = torch.randn(B, N, C, generator=g).cuda()
    res = []
    for fn in (fused.softmax_wsum_k, lambda l, v: torch.sum(F.softmax(l, dim=2) * v, dim=2)):
        l, v = logit.clone().requires_grad_(True), value.clone().requires_grad_(True)
        out = fn(l, v)
        gl, gv = torch.autograd.grad(out, (l, v), go)
        res.append((out, gl, gv))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,K,C", [(2, 57, 80, 32, 128), (1, 9, 30, 4, 64), (3, 20, 16, 8, 12)])
def test_knn_rows_matches_gather_mul_cat(hip_backend, B, N, M, K, C):
    """i2p_knn_rows_fwd/bwd (kNN pi-stage rows of the fine cost volume in one launch) against the chain it replaces — two
    index_points_group gathers, the product and the zero-padded cat: rows bit-exact, gradients w.r.t. the point coordinates /
    features and the pixel features to fp32 rounding; repeated indices (several neighbours = the same pixel) included"""
    from i2pnet_amd import projectpn as P
    from i2pnet_amd.modules import cat_padded
    g = torch.Generator().manual_seed(N * K + C)
    xyz = torch.randn(B, N, 3, generator=g).cuda(); pix_xyz = torch.randn(B, M, 3, generator=g).cuda()
    pts = torch.randn(B, N, C, generator=g).cuda(); pix = torch.randn(B, M, C, generator=g).cuda()
    idx = torch.randint(0, M, (B, N, K), generator=g).cuda()
    cpad = (6 + C + 3) // 4 * 4
    go = torch.randn(B, N, K, cpad, generator=g).cuda()
    res = []
    for fused in (True, False):
        a, b, c = xyz.clone().requires_grad_(True), pts.clone().requires_grad_(True), pix.clone().requires_grad_(True)
        if fused:
            rows = P.knn_rows(a, pix_xyz, b, c, idx, cpad)
        else:
            own = a.unsqueeze(2).expand(-1, -1, K, -1)
            rows = cat_padded([own, P.index_points_group(pix_xyz, idx), b.unsqueeze(2) * P.index_points_group(c, idx)])
        res.append((rows,) + torch.autograd.grad(rows, (a, b, c), go))
    assert torch.equal(res[0][0], res[1][0])
    for got, want in zip(res[0][1:], res[1][1:]):
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6


def _on_oracle(oracle_backend, fn):
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        return fn()
    finally:
        ops.set_backend(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("xyz_first", [True, False])
def test_sa_rows_full_size_vs_oracle_chain(hip_backend, oracle_backend, xyz_first):
    """i2p_sa_rows at the level-2 size (16 x 225 image, 904 centres x 16 neighbours, 32 channels) against the oracle chain:
    fused_conv_select_k_cpu -> gather_rows_cpu x2 -> subtraction -> zero-padded cat (PPBackbone_center.py:94-129); forward rows
    bit-exact, feature gradient (fixed-point scatter on the device, gather_rows_grad_cpu in the oracle) to fp32 rounding."""
    from i2pnet_amd import modules, projectpn as P
    B, H, W, C, oh, ow, K = 2, 16, 225, 32, 8, 113, 16
    img = range_image(B, H, W, seed=5, empty_frac=0.4, scale=8.0)
    feat = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(6))
    idx = stride_grid(B, oh, ow, 2, 2)
    centre = img[:, ::2, ::2][:, :oh, :ow].reshape(B, oh * ow, 3).contiguous()
    gout = torch.randn(B, oh * ow, K, 128, generator=torch.Generator().manual_seed(7))

    def chain():
        f = feat.clone().requires_grad_(True)
        sb, sh, sw, m, _, _ = run_fcsk(oracle_backend, img, img, idx, 9, 15, K, 3, 3.0, 1, 1)
        d = P.gather_torch(img, None, sh.squeeze(-1), sw.squeeze(-1), B, H, W) - centre.view(B, -1, 1, 3)
        f2 = P.gather_torch(f, None, sh.squeeze(-1), sw.squeeze(-1), B, H, W)
        rows = modules.cat_padded([d, f2] if xyz_first else [f2, d], pow2=True)
        (gf,) = torch.autograd.grad(rows, f, gout[..., :rows.shape[-1]])
        return rows, gf, sh.squeeze(-1), sw.squeeze(-1)
    want, want_g, sh, sw = _on_oracle(oracle_backend, chain)
    f = feat.to(DEV).requires_grad_(True)
    rows = P.sa_rows(img.to(DEV), centre.to(DEV), f, sh.to(DEV), sw.to(DEV), K, W, xyz_first=xyz_first)
    assert rows.shape == want.shape and torch.equal(rows.cpu(), want)
    (gf,) = torch.autograd.grad(rows, f, gout[..., :rows.shape[-1]].to(DEV))
    assert float((gf.cpu() - want_g).abs().max()) <= 2e-6 * float(want_g.abs().max()) + 1e-6


@pytest.mark.gpu
def test_knn_rows_full_size_vs_oracle_chain(hip_backend, oracle_backend):
    """i2p_knn_rows_fwd/bwd at the fine cost volume's size (228 points x 32 nearest of 468 pixels, 128 channels) against the
    oracle chain: knn_cpu -> group_points_cpu gathers -> product -> cat (PPBackbone_center.py:367-395)."""
    from i2pnet_amd import projectpn as P
    from i2pnet_amd.modules import cat_padded
    B, N, M, K, C = 2, 228, 468, 32, 128
    g = torch.Generator().manual_seed(31)
    xyz = torch.randn(B, N, 3, generator=g); pix_xyz = torch.randn(B, M, 3, generator=g)
    pts = torch.randn(B, N, C, generator=g); pix = torch.randn(B, M, C, generator=g)
    cpad = (6 + C + 3) // 4 * 4
    go = torch.randn(B, N, K, cpad, generator=g)

    def chain():
        a, b, c = xyz.clone().requires_grad_(True), pts.clone().requires_grad_(True), pix.clone().requires_grad_(True)
        idx = P.knn_point(K, pix_xyz, xyz)
        own = a.unsqueeze(2).expand(-1, -1, K, -1)
        rows = cat_padded([own, P.index_points_group(pix_xyz, idx), b.unsqueeze(2) * P.index_points_group(c, idx)])
        return (rows, idx) + torch.autograd.grad(rows, (a, b, c), go)
    want, idx, wa, wb, wc = _on_oracle(oracle_backend, chain)
    a, b, c = xyz.to(DEV).requires_grad_(True), pts.to(DEV).requires_grad_(True), pix.to(DEV).requires_grad_(True)
    idx_d = P.knn_point(K, pix_xyz.to(DEV), xyz.to(DEV))
    assert torch.equal(idx_d.cpu(), idx)
    rows = P.knn_rows(a, pix_xyz.to(DEV), b, c, idx_d, cpad)
    assert torch.equal(rows.cpu(), want)
    for got, ref in zip(torch.autograd.grad(rows, (a, b, c), go.to(DEV)), (wa, wb, wc)):
        assert float((got.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6


@pytest.mark.gpu
def test_cv_tail_kernels_vs_cpu_float64(hip_backend):
    """i2p_outer_sum and the softmax-weighted-sum kernels against plain CPU restatements in fp64 (PPBackbone_center.py:423-433,
    :481-487): position-encoding rows bit-exact (one fp32 addition), softmax tail to 1e-6."""
    from i2pnet_amd import fused
    g = torch.Generator().manual_seed(77)
    B, N, M, C = 2, 57, 468, 64
    en, ek = torch.randn(B, N, C, generator=g), torch.randn(B, M, C, generator=g)
    ye, sums = hip_backend.outer_sum(en.to(DEV), ek.to(DEV))
    want = (en.unsqueeze(2) + ek.unsqueeze(1)).reshape(B * N * M, C)
    assert torch.equal(ye.cpu(), want)
    s = sums.cpu().view(-1, 2, C).sum(0)
    assert torch.allclose(s[0], want.double().sum(0), rtol=1e-9, atol=1e-6) and torch.allclose(s[1], want.double().square().sum(0), rtol=1e-9, atol=1e-6)
    K = 4
    logit = torch.randn(B, N, K, C, generator=g); logit[0, 0] = -1e10; logit[0, 1, 1:] = -1e10
    value = torch.randn(B, N, K, C, generator=g); go = torch.randn(B, N, C, generator=g)
    ld, vd = logit.double().requires_grad_(True), value.double().requires_grad_(True)
    ref = torch.sum(torch.softmax(ld, dim=2) * vd, dim=2)
    rl, rv = torch.autograd.grad(ref, (ld, vd), go.double())
    l, v = logit.to(DEV).requires_grad_(True), value.to(DEV).requires_grad_(True)
    out = fused.softmax_wsum_k(l, v)
    gl, gv = torch.autograd.grad(out, (l, v), go.to(DEV))
    for a, b in ((out, ref), (gl, rl), (gv, rv)):
        assert float((a.cpu().double() - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-7


@pytest.mark.gpu
def test_glue_kernels_vs_cpu(hip_backend):
    """csrc/glue.hip against the torch expressions of the reference they replace, evaluated on the CPU: check_valid
    (utils.py:106-108), the -1e10 mask fill and its gradient (modellearn_proj_center.py:318), zero-padded weights, strided
    centre picks (PPBackbone_center.py:94-95)."""
    from i2pnet_amd import modules, projectpn as P
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 37, 3, generator=g); x[0, :5] = 0.0; x[1, 7, :2] = 0.0; x[2, 9] = torch.tensor([0.0, -0.0, 0.0])
    assert torch.equal(P.check_valid(x.to(DEV)).cpu(), torch.any(torch.ne(x, 0), dim=-1, keepdim=True).float())
    for c in (64, 6):
        a = torch.randn(4, 29, c, generator=g); valid = (torch.rand(4, 29, 1, generator=g) > 0.4).float(); go = torch.randn(4, 29, c, generator=g)
        ad = a.to(DEV).requires_grad_(True)
        out = modules.mask_fill(ad, valid.to(DEV))
        (ga,) = torch.autograd.grad(out, ad, go.to(DEV))
        ar = a.clone().requires_grad_(True)
        ref = ar * valid + (-1e10) * (1.0 - valid)                                          # the reference's formulation
        (gr,) = torch.autograd.grad(ref, ar, go)
        assert torch.equal(out.cpu(), ref.detach()) and torch.equal(ga.cpu(), gr)
    w = torch.randn(128, 131, generator=g)
    assert torch.equal(hip_backend.pad_cols(w.to(DEV), 132).cpu(), torch.nn.functional.pad(w, (0, 1)))
    ia, ib = torch.randn(2, 21, 1800, 3, generator=g), torch.randn(2, 21, 1800, 3, generator=g)
    oa, ob = modules._centres(ia.to(DEV), ib.to(DEV), 2, 8, 11, 225)
    assert torch.equal(oa.cpu(), ia[:, ::2, ::8][:, :11, :225]) and torch.equal(ob.cpu(), ib[:, ::2, ::8][:, :11, :225])


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,C,empty", [(8, 228, 468, 128, False), (2, 57, 80, 64, True), (1, 9, 33, 20, False)])
def test_max_response_fused_vs_torch_cpu(hip_backend, B, N, M, C, empty):
    """i2p_max_response_fwd/bwd against the torch formulation of modules._MaxResponse evaluated on the CPU (itself the closed form of
    PPBackbone_center.py:408-414): values bit-exact (one product per element), gradients to fp32 summation order; a sample without
    any valid point (-1e10 everywhere, zero gradients) included."""
    from i2pnet_amd import modules
    g = torch.Generator().manual_seed(B * N + C)
    pts, pix = torch.randn(B, N, C, generator=g), torch.randn(B, M, C, generator=g)
    valid = (torch.rand(B, N, 1, generator=g) > 0.3).float()
    if empty:
        valid[0] = 0.0
    go = torch.randn(B, M, C, generator=g)
    a, b = pts.clone().requires_grad_(True), pix.clone().requires_grad_(True)
    ref = modules._MaxResponse.apply(a, b, valid)
    ra, rb = torch.autograd.grad(ref, (a, b), go)
    ad, bd = pts.to(DEV).requires_grad_(True), pix.to(DEV).requires_grad_(True)
    out = modules.max_response(ad, bd, valid.to(DEV))
    ga, gb = torch.autograd.grad(out, (ad, bd), go.to(DEV))
    assert torch.equal(out.cpu(), ref.detach())
    assert torch.equal(gb.cpu(), rb)
    assert float((ga.cpu() - ra).abs().max()) <= 1e-5 * float(ra.abs().max()) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("need_xyz", [True, False])
def test_pc_rows_vs_oracle_chain(hip_backend, oracle_backend, need_xyz):
    """i2p_pc_rows_fwd/bwd (pc-stage front end of the cost volumes in one launch each way) at the network's size (4 x 57 cells,
    K = 4 of a 3 x 5 window, 128 + 64 channels) against the ORACLE chain it replaces: fused_conv_select_k_cpu -> gather_rows_cpu x2,
    expand, subtraction, sqrt(sum + 1e-20), cats (PPBackbone_center.py:443-476) with gather_rows_grad_cpu in its backward: forward
    rows bit-exact, the three input gradients to fp32 rounding."""
    from i2pnet_amd import projectpn as P
    B, H, W, K, C, c = 2, 4, 57, 4, 128, 64
    N = H * W
    img = range_image(B, H, W, seed=9, empty_frac=0.3, scale=6.0)
    g = torch.Generator().manual_seed(10)
    pts, feat = torch.randn(B, N, C, generator=g), torch.randn(B, N, c, generator=g)
    idx = stride_grid(B, H, W, 1, 1)
    g_geo, g_part, g_nbf = torch.randn(B, N, K, 12, generator=g), torch.randn(B, N, K, C + c, generator=g), torch.randn(B, N, K, c, generator=g)
    g_geo[..., 10:] = 0.0

    def chain():
        x = img.reshape(B, N, 3).clone().requires_grad_(need_xyz); p_, f_ = pts.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        sb, sh, sw, m, _, _ = run_fcsk(oracle_backend, img, img, idx, 3, 5, K, 2, 4.5, 1, 1)
        hh, ww = sh.squeeze(-1), sw.squeeze(-1)
        nb_xyz = P.gather_torch(x.view(B, H, W, 3), None, hh, ww, B, H, W)
        nb_feat = P.gather_torch(f_, None, hh, ww, B, H, W)
        own = x.unsqueeze(2).expand(-1, -1, K, -1)
        diff = nb_xyz - own
        euc = torch.sqrt(torch.sum(diff * diff, dim=3, keepdim=True) + 1e-20)
        geo = torch.cat([own, nb_xyz, diff, euc, torch.zeros(B, N, K, 2)], dim=3)
        part = torch.cat([p_.unsqueeze(2).expand(-1, -1, K, -1), nb_feat], dim=-1)
        ins = ([x] if need_xyz else []) + [p_, f_]
        outs, gs = ([geo, part, nb_feat], [g_geo, g_part, g_nbf]) if need_xyz else ([part, nb_feat], [g_part, g_nbf])
        grads = torch.autograd.grad(outs, ins, gs)
        return geo.detach(), part.detach(), nb_feat.detach(), grads, hh, ww
    geo, part, nbf, grads, hh, ww = _on_oracle(oracle_backend, chain)
    x = img.reshape(B, N, 3).to(DEV).requires_grad_(need_xyz); p_, f_ = pts.to(DEV).requires_grad_(True), feat.to(DEV).requires_grad_(True)
    got = P.pc_rows(x, p_, f_, hh.to(DEV), ww.to(DEV), K, W)
    assert torch.equal(got[0][..., :9].cpu(), geo[..., :9]) and torch.equal(got[0][..., 10:].cpu(), geo[..., 10:])
    assert torch.allclose(got[0][..., 9].cpu(), geo[..., 9], rtol=2e-7, atol=0)       # torch.sum's order over the 3 squares may differ by an ulp
    assert torch.equal(got[1].cpu(), part) and torch.equal(got[2].cpu(), nbf)
    ins = ([x] if need_xyz else []) + [p_, f_]
    hg = torch.autograd.grad(list(got), ins, [g_geo.to(DEV), g_part.to(DEV), g_nbf.to(DEV)])
    for a, b in zip(hg, grads):
        assert float((a.cpu() - b).abs().max()) <= 3e-6 * float(b.abs().max()) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cin,cout", [(853632, 128, 64), (65584, 64, 64), (14848, 64, 128), (3000, 32, 16), (58368, 136, 128)])
def test_last_block_finalize_is_stable_over_many_launches(hip_backend, rows, cin, cout):
    """The BN coefficients a forward launch finalises in its LAST block (ticket atomic after the statistics atomics, csrc/mlp.hip
    finalize_by_last_block and the wreg / big variants) against the separate i2p_bn_finalize on the same sums, bit for bit, over 150
    back-to-back launches per shape: a lost or late partial sum (a missing release/acquire across the XCDs' L2s) would show as a
    differing mean / invstd."""
    g = torch.Generator(device=DEV).manual_seed(rows % 1000)
    x = torch.randn(rows, cin, generator=g, device=DEV) * 1.5 + 0.2
    w = torch.randn(cout, cin, generator=g, device=DEV) / cin ** 0.5
    gam = 1 + 0.1 * torch.randn(cout, generator=g, device=DEV); bet = 0.1 * torch.randn(cout, generator=g, device=DEV)
    in_coef, _ = hip_backend.bn_finalize(rows, hip_backend.bn_stats(x), torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV), 1e-5)
    bad = 0
    for it in range(150):
        y, sums, coef, mi = hip_backend.lin_forward_fin(x, in_coef, 0.1, w, gam, bet, 1e-5)
        c2, m2 = hip_backend.bn_finalize(rows, sums, gam, bet, 1e-5)
        bad += int(not (torch.equal(coef, c2) and torch.equal(mi, m2)))
    assert bad == 0, bad


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,p", [(8, 64, 0.5), (2, 64, 0.0), (16, 64, 0.5), (5, 128, 0.3)])
def test_pose_head_mlp_fused_vs_torch_cpu(hip_backend, oracle_backend, B, C, p):
    """modules.PoseHead's regression MLP (hidden Conv1d -> Dropout -> quaternion / translation Conv1d -> normalisation,
    PPBackbone_center.py:553-562) on the fused kernels against the same module evaluated by plain torch on the CPU with the SAME
    dropout mask: outputs and every gradient to fp32 summation order."""
    from i2pnet_amd import modules
    import torch.nn.functional as F
    torch.manual_seed(B + C)
    head = modules.PoseHead([C, C], [], [], 256, 4, 3, dropout_rate=p)
    head.train()
    pooled = torch.randn(B, 1, C)
    mask = (torch.rand(B, 256) > p).float() / (1.0 - p) if p > 0 else None
    gq, gt = torch.randn(B, 4), torch.randn(B, 3)
    # reference: the module's torch formulation with the mask applied by hand
    conv = lambda h: h.composed_module[0]
    ref_in = pooled.clone().requires_grad_(True)
    hid = F.linear(ref_in, conv(head.hidden_layer).weight.squeeze(-1), conv(head.hidden_layer).bias)
    if mask is not None:
        hid = hid * mask.unsqueeze(1)
    q = F.linear(hid, conv(head.quat_head).weight.squeeze(-1), conv(head.quat_head).bias).squeeze(1)
    t = F.linear(hid, conv(head.trans_head).weight.squeeze(-1), conv(head.trans_head).bias).squeeze(1)
    q = q / (torch.sqrt(torch.sum(q * q, dim=-1, keepdim=True) + 1e-10) + 1e-10)
    params = [conv(h).weight for h in (head.hidden_layer, head.quat_head, head.trans_head)] + [conv(h).bias for h in (head.hidden_layer, head.quat_head, head.trans_head)]
    ref_g = torch.autograd.grad([q, t], [ref_in] + params, [gq, gt])
    # fused
    hd = modules.PoseHead([C, C], [], [], 256, 4, 3, dropout_rate=p).to(DEV)
    hd.load_state_dict(head.state_dict())
    x = pooled.to(DEV).requires_grad_(True)
    dparams = [conv(h).weight for h in (hd.hidden_layer, hd.quat_head, hd.trans_head)] + [conv(h).bias for h in (hd.hidden_layer, hd.quat_head, hd.trans_head)]
    qd, td = modules._PoseHeadMlp.apply(x.reshape(B, C), dparams[0].squeeze(-1), dparams[3], dparams[1].squeeze(-1), dparams[4], dparams[2].squeeze(-1),
                                        dparams[5], None if mask is None else mask.to(DEV))
    got_g = torch.autograd.grad([qd, td], [x] + dparams, [gq.to(DEV), gt.to(DEV)])
    assert torch.allclose(qd.cpu(), q.detach(), rtol=1e-5, atol=1e-6) and torch.allclose(td.cpu(), t.detach(), rtol=1e-5, atol=1e-5)
    for a, b in zip(got_g, ref_g):
        assert float((a.cpu() - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6
    # and through the module (training mode draws its own mask: compare the p = 0 / eval case only)
    if p == 0.0:
        valid = torch.ones(B, 57, C)
        pred = torch.randn(B, 57, C)
        q1, t1, _ = hd(pred.to(DEV), valid.to(DEV), None, None, None)
        modules.USE_FUSED_MLP = False                      # (plain torch + oracle operators on the CPU for the reference module)
        try:
            q0, t0, _ = _on_oracle(oracle_backend, lambda: head(pred, valid, None, None, None))
        finally:
            modules.USE_FUSED_MLP = True
        assert torch.allclose(q1.cpu(), q0, rtol=1e-4, atol=1e-5) and torch.allclose(t1.cpu(), t0, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(8, 228), (2, 600), (1, 5)])
def test_warp_split_vs_reference_chain_on_cpu(hip_backend, oracle_backend, B, N):
    """i2p_warp_split_fwd/bwd against the chain it replaces evaluated on the CPU with the oracle's quaternion operators
    (warp_utils.py:78-94 warp_quat_xyz, `* valid`, z = P[:, :, 2:], uv = P / (z + 1e-10), xyz = uv * z): forward bit-exact, the pose
    gradients to fp32 summation order."""
    from i2pnet_amd import warp as warp_utils
    g = torch.Generator().manual_seed(B * 1000 + N)
    p = torch.randn(B, N, 3, generator=g) * 10 + torch.tensor([0.0, 0.0, 20.0])
    q = torch.randn(B, 4, generator=g); q = q / q.norm(dim=1, keepdim=True) * (1 + 0.01 * torch.randn(B, 1, generator=g))
    t = torch.cat([torch.zeros(B, 1), torch.randn(B, 3, generator=g)], 1)
    valid = (torch.rand(B, N, 1, generator=g) > 0.2).float()
    gu, gz, gx = torch.randn(B, N, 3, generator=g), torch.randn(B, N, 1, generator=g), torch.randn(B, N, 3, generator=g)

    def chain():
        qq, tt = q.clone().requires_grad_(True), t.clone().requires_grad_(True)
        P3 = warp_utils.warp_quat_xyz(p, qq, tt) * valid
        z = P3[:, :, 2:]
        uv = P3 / (z + 1e-10)
        xyz = uv.mul(z)
        return (uv.detach(), z.detach(), xyz.detach()) + torch.autograd.grad([uv, z, xyz], [qq, tt], [gu, gz, gx])
    uv, z, xyz, dq, dt = _on_oracle(oracle_backend, chain)
    qd, td = q.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
    got = warp_utils.warp_split(p.to(DEV), qd, td, valid.to(DEV))
    for a, b in zip(got, (uv, z, xyz)):
        assert torch.equal(a.cpu(), b)
    gq, gt = torch.autograd.grad(list(got), [qd, td], [gu.to(DEV), gz.to(DEV), gx.to(DEV)])
    assert float((gq.cpu() - dq).abs().max()) <= 2e-5 * float(dq.abs().max()) + 1e-5
    assert float((gt.cpu() - dt).abs().max()) <= 2e-5 * float(dt.abs().max()) + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 16, 1, 70])
def test_pose_compose_vs_reference_chain_on_cpu(hip_backend, oracle_backend, B):
    """i2p_pose_compose_fwd/bwd against the chain it replaces evaluated on the CPU with the oracle's quaternion operators
    (modellearn_proj_center.py:388-404: q = q3 * q_prev, t = (q3 [0,t_prev] q3^-1)[1:4] + t3): forward bit-exact, gradients to fp32
    rounding of the same products."""
    import os
    from i2pnet_amd import warp as warp_utils
    g = torch.Generator().manual_seed(B)
    unit = lambda: torch.nn.functional.normalize(torch.randn(B, 4, generator=g), dim=1) * (1 + 0.01 * torch.randn(B, 1, generator=g))
    q3, qp = unit(), unit()
    t3 = torch.randn(B, 3, generator=g)
    tpq = torch.cat([torch.zeros(B, 1), torch.randn(B, 3, generator=g)], 1)
    go = torch.randn(B, 7, generator=g)

    def chain():
        a, b, c, d = [t.clone().requires_grad_(True) for t in (q3, t3, qp, tpq)]
        out = warp_utils.compose_pose(a, b, c, d)
        return (out.detach(),) + torch.autograd.grad(out, [a, b, c, d], go)
    ref = _on_oracle(oracle_backend, chain)
    dev = [t.to(DEV).requires_grad_(True) for t in (q3, t3, qp, tpq)]
    out = warp_utils.compose_pose(*dev)
    assert out.grad_fn is not None and type(out.grad_fn).__name__ == "_PoseComposeBackward"
    assert torch.equal(out.detach().cpu(), ref[0])
    grads = torch.autograd.grad(out, dev, go.to(DEV))
    for k, (a, b) in enumerate(zip(grads, ref[1:])):
        b = b if k != 3 else torch.cat([torch.zeros(B, 1), b[:, 1:]], 1)        # w of [0, t_prev] is a constant: its gradient is discarded
        a = a.cpu() if k != 3 else torch.cat([torch.zeros(B, 1), a.cpu()[:, 1:]], 1)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6, k
    # and against the unfused device chain
    os.environ["I2P_NO_POSE_COMPOSE"] = "1"
    try:
        dev2 = [t.to(DEV).requires_grad_(True) for t in (q3, t3, qp, tpq)]
        out2 = warp_utils.compose_pose(*dev2)
    finally:
        os.environ.pop("I2P_NO_POSE_COMPOSE", None)
    assert torch.equal(out2, out)


@pytest.mark.gpu
@pytest.mark.parametrize("groups,K,c", [(28800, 32, 32), (7232, 16, 64), (928, 16, 128), (5, 3, 16)])
def test_unpool_k_stats_matches_two_pass(hip_backend, oracle_backend, groups, K, c):
    """i2p_unpool_k_stats (dense max-pool gradient + the BN-backward statistics of the layer underneath in one pass) against the
    oracle's unpool_k_cpu -> bn_act_bwd_stats_cpu pair: gradient rows bit-exact, statistics to fp64 summation order."""
    g = torch.Generator().manual_seed(groups + K)
    y = torch.randn(groups * K, c, generator=g) * 1.3 + 0.2
    gam = 1 + 0.1 * torch.randn(c, generator=g); bet = 0.1 * torch.randn(c, generator=g)
    var, mean = torch.var_mean(y.double(), 0, unbiased=False)
    mi = torch.cat([mean, torch.rsqrt(var + 1e-5)]).float()
    gp = torch.randn(groups, c, generator=g)
    arg = torch.randint(0, K, (groups, c), generator=g).to(torch.uint8)
    want_gd = oracle_backend.unpool_k(gp, arg, K)
    want_ds = oracle_backend.bn_act_backward_stats(want_gd, y, mi, gam, bet, 0.0)
    gd, ds = hip_backend.unpool_k_stats(gp.to(DEV), arg.to(DEV), K, y.to(DEV), mi.to(DEV), gam.to(DEV), bet.to(DEV), 0.0)
    assert torch.equal(gd.cpu(), want_gd)
    a, b = ds.cpu().view(32, 2, c).sum(0), want_ds.view(32, 2, c).sum(0)
    assert torch.allclose(a, b, rtol=1e-9, atol=1e-7 * groups ** 0.5)

@pytest.mark.gpu
def test_intrinsic_inverse_matches_the_torch_form(hip_backend):
    """i2p_intrinsic_inverse = change_intrinsic + inverse_3x3 of model.py (modellearn_proj_center.py:457-463, :282) in one launch: the same
    arithmetic, so bit-identical to the torch expression; and an inverse (K_scaled @ K_inv = I to fp32 rounding, fp64 torch.inverse 1e-6)."""
    from i2pnet_amd import model
    g = torch.Generator().manual_seed(3)
    B = 16
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = 700 + 50 * torch.rand(B, generator=g); K[:, 1, 1] = 700 + 50 * torch.rand(B, generator=g)
    K[:, 0, 2] = 600 + 20 * torch.rand(B, generator=g); K[:, 1, 2] = 180 + 10 * torch.rand(B, generator=g); K[:, 2, 2] = 1.0
    K[:, 0, 1] = 0.01 * torch.randn(B, generator=g)                      # a little skew: the general adjugate, not only the triangular case
    Kd = K.to(DEV)
    RF, img = torch.empty(1, 1, 24, 78), torch.empty(1, 1, 375, 1242)
    want = model.inverse_3x3(model.change_intrinsic(Kd, RF, img))
    got = hip_backend.intrinsic_inverse(Kd, 78 / 1242, 24 / 375)
    assert torch.equal(got, want)
    ref = torch.linalg.inv((K.double() * torch.tensor([[78 / 1242, 1.0, 78 / 1242], [1.0, 24 / 375, 24 / 375], [1.0, 1.0, 1.0]], dtype=torch.float64)))
    assert float((got.cpu().double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
