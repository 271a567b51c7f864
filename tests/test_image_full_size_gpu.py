"""The image encoder's hand-written kernels at the sizes bench.py runs them (configs[1]/[2]/[4]: batch 8, 375x1242 -> 188x621),
against the reference's own arithmetic (src/modules/basicConv.py:6-20: Conv2d(3x3, padding 1) + BatchNorm2d(train) + LeakyReLU(0.1)
+ MaxPool2d(3, stride, 1)) evaluated in fp64 with plain torch ops on the device (unfold + matmul for the convolution: MIOpen has no
fp64 convolution).  The small-shape tests (test_image_first_gpu.py, test_image_conv16_gpu.py, test_ops_gpu.py) cover the edges;
these cover the grids, strides and sums of the timed sizes.  Tolerances are the small-shape ones."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, H1, W1 = 8, 375, 1242
H2, W2 = 188, 621


def conv64(x, w):
    """Conv2d(3x3, padding 1) of x [B,C,H,W] with w [K,C,3,3], both fp64, as unfold + matmul (differentiable)"""
    Bn, C, H, W = x.shape
    win = F.unfold(x, 3, padding=1)                                   # [B, C*9, H*W] in (ci, kh, kw) order
    return (w.reshape(w.shape[0], -1) @ win).reshape(Bn, w.shape[0], H, W)


def block64(y, gam, bet, stride, slope=0.1, eps=1e-5):
    a = F.leaky_relu(F.batch_norm(y, None, None, gam, bet, True, 0.1, eps), slope)
    return F.max_pool2d(a, 3, stride, 1), a


def picked_by(arg, act, stride, shape):
    """reference activation at the window position each arg byte [B,Ho,Wo,C] names (+ the check that none names the padding)"""
    Bn, C, H, W = act.shape
    Ho, Wo = shape[2:]
    a = arg.long().permute(0, 3, 1, 2)
    h = torch.arange(Ho, device=DEV).view(1, 1, Ho, 1) * stride - 1 + a // 3
    w = torch.arange(Wo, device=DEV).view(1, 1, 1, Wo) * stride - 1 + a % 3
    assert int(h.min()) >= 0 and int(h.max()) < H and int(w.min()) >= 0 and int(w.max()) < W, "arg-max points into the padding"
    return act.reshape(Bn, C, H * W).gather(2, (h * W + w).reshape(Bn, C, -1)).reshape(shape)


def test_first_block_at_bench_size(hip_backend):
    """i2p_img_first_fwd / _bwd (image_first.hip) on 8 x 375 x 1242, stride 2"""
    g = torch.Generator().manual_seed(20)
    x = torch.randn(B, 3, H1, W1, generator=g) * 1.1 + torch.tensor([0.4, -0.2, 0.1]).view(1, 3, 1, 1)
    w = torch.randn(16, 3, 3, 3, generator=g) * 0.3
    gam, bet = torch.randn(16, generator=g), torch.randn(16, generator=g) * 0.2
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last); wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    out, arg, mi, gram = hip_backend.img_first_forward(xd, wd, gam.to(DEV), bet.to(DEV), 1e-5, 0.1, 2)
    x64 = x.to(DEV).double()
    w64, g64, b64 = w.to(DEV).double().requires_grad_(), gam.to(DEV).double().requires_grad_(), bet.to(DEV).double().requires_grad_()
    y = conv64(x64, w64)
    ref, act = block64(y, g64, b64, 2)
    n = B * H1 * W1
    assert gram.view(32, 32)[27, 27].item() == n
    mean, var = y.detach().mean((0, 2, 3)), y.detach().var((0, 2, 3), unbiased=False)
    assert torch.allclose(mi[:16].double(), mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(mi[16:].double(), (var + 1e-5).rsqrt(), rtol=2e-5)
    got = out.permute(0, 3, 1, 2).double()
    assert (got - ref.detach()).abs().max().item() <= 2e-5 * max(ref.detach().abs().max().item(), 1.0)
    picked = picked_by(arg, act, 2, ref.shape)
    assert (ref.detach() - picked.detach()).abs().max().item() <= 1e-6 * ref.detach().abs().max().item(), "arg-max off a near-tie"
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    (picked * gout.double()).sum().backward()                        # the kernel's own arg-max in the fp64 reference
    dW, dg, db = hip_backend.img_first_backward(gout.permute(0, 2, 3, 1).contiguous(), arg, xd, wd, gam.to(DEV), bet.to(DEV), 0.1, 2, mi, gram)
    for name, r, got in (("dW", w64.grad, dW), ("dgamma", g64.grad, dg), ("dbeta", b64.grad, db)):
        err = (got.double() - r).abs().max().item()
        assert err <= 2e-4 * max(r.abs().max().item(), 1e-3), f"{name}: {err} vs scale {r.abs().max().item()}"


@pytest.mark.parametrize("cout", [16, 32])
def test_conv16_at_bench_size(hip_backend, cout):
    """i2p_img_conv_fwd (+ BatchNorm sums) / _bwd_data / _wgrad (image_conv16.hip) on 8 x 188 x 621 x 16"""
    g = torch.Generator().manual_seed(30 + cout)
    x = torch.randn(B, H2, W2, 16, generator=g)
    w = torch.randn(cout, 16, 3, 3, generator=g) * 0.2
    dy = torch.randn(B, H2, W2, cout, generator=g)
    xd, dyd = x.to(DEV), dy.to(DEV)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    x64 = xd.permute(0, 3, 1, 2).double().requires_grad_()
    w64 = w.to(DEV).double().requires_grad_()
    ref = conv64(x64, w64)
    (ref * dyd.permute(0, 3, 1, 2).double()).sum().backward()
    ref = ref.detach().permute(0, 2, 3, 1)
    y, sums = hip_backend.img_conv16(xd, wd, with_sums=True)
    sc = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= 2e-6 * sc
    s = sums.view(-1, 2 * cout).sum(0)
    n = B * H2 * W2
    assert torch.allclose(s[:cout], ref.sum((0, 1, 2)), rtol=1e-5, atol=2e-6 * sc * n)
    assert torch.allclose(s[cout:], (ref * ref).sum((0, 1, 2)), rtol=2e-6, atol=1e-9)
    dx = hip_backend.img_conv16(dyd, wd, input_grad=True)
    dref = x64.grad.permute(0, 2, 3, 1)
    assert (dx.double() - dref).abs().max().item() <= 2e-6 * dref.abs().max().item()
    dW = hip_backend.img_conv16_wgrad(xd, dyd, wd)
    assert (dW.double() - w64.grad).abs().max().item() <= 5e-6 * w64.grad.abs().max().item()


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("stride", [1, 2])
def test_block_tail_at_bench_size(hip_backend, stride, storage):
    """i2p_img_block_fwd / _bwd (image_block.hip: BatchNorm statistics, LeakyReLU, 3x3 max-pool with arg bytes and their backward)
    on 8 x 188 x 621 x 16, in fp32 storage and in configs[2]/[4]'s bf16 storage of the conv output and the pooled output"""
    bf = storage == "bf16"
    g = torch.Generator().manual_seed(40 + stride)
    y = (torch.randn(B, H2, W2, 16, generator=g) * 2 + 0.3).to(DEV)
    yd = y.to(torch.bfloat16) if bf else y
    gam, bet = torch.randn(16, generator=g).to(DEV), (torch.randn(16, generator=g) * 0.2).to(DEV)
    out, arg, mi = hip_backend.img_block_forward(yd, gam, bet, 1e-5, 0.1, stride, out_bf16=bf)
    y64 = yd.double().permute(0, 3, 1, 2).requires_grad_()            # the values the kernel finds in memory
    g64, b64 = gam.double().requires_grad_(), bet.double().requires_grad_()
    ref, act = block64(y64, g64, b64, stride)
    mean, var = y64.detach().mean((0, 2, 3)), y64.detach().var((0, 2, 3), unbiased=False)
    assert torch.allclose(mi[:16].double(), mean, rtol=1e-5, atol=1e-6) and torch.allclose(mi[16:].double(), (var + 1e-5).rsqrt(), rtol=2e-5)
    got = out.double().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref.detach(), rtol=2.0 ** -8 if bf else 1e-5, atol=1e-5)
    picked = picked_by(arg, act, stride, ref.shape)
    assert (ref.detach() - picked.detach()).abs().max().item() <= 1e-6 * ref.detach().abs().max().item(), "arg-max off a near-tie"
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(6)).to(DEV)
    gd = gout.to(torch.bfloat16) if bf else gout
    (picked * gd.double().permute(0, 3, 1, 2)).sum().backward()
    dy, dg, db = hip_backend.img_block_backward(gd, arg, yd, mi, gam, bet, 0.1, stride)
    assert dy.dtype == yd.dtype
    rdy = y64.grad.permute(0, 2, 3, 1)
    sc = rdy.abs().max().item()
    assert torch.allclose(dy.double(), rdy, rtol=2.0 ** -8 if bf else 1e-4, atol=1e-5 * sc)
    assert torch.allclose(dg.double(), g64.grad, rtol=1e-4, atol=1e-4 * g64.grad.abs().max().item())
    assert torch.allclose(db.double(), b64.grad, rtol=1e-4, atol=1e-4 * b64.grad.abs().max().item())
