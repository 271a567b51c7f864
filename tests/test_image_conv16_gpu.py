"""GPU parity of csrc/image_conv16.hip — Conv2d(16, 16, 3, padding=1) on NHWC fp32 tensors (the image encoder's blocks 2-4,
src/modules/basicConv.py:6-20), its fused BatchNorm sums and its input gradient — against torch's convolution in fp64 on the CPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("layout", ["contiguous", "channels_last"])
@pytest.mark.parametrize("cout", [16, 32])
@pytest.mark.parametrize("B,H,W", [(2, 37, 83), (1, 5, 14), (3, 1, 1), (1, 64, 29), (2, 3, 200)])
def test_conv16_forward_sums_and_input_gradient(hip_backend, layout, cout, B, H, W):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + cout)
    x = torch.randn(B, H, W, 16, generator=g)
    w = torch.randn(cout, 16, 3, 3, generator=g) * 0.2
    wd = w.to(DEV)
    if layout == "channels_last":
        wd = wd.contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    y, sums = hip_backend.img_conv16(x.to(DEV), wd, with_sums=True)
    sc = ref.abs().max().item()
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-6 * sc
    y2 = hip_backend.img_conv16(x.to(DEV), wd)
    assert torch.equal(y, y2)
    s = sums.view(-1, 2 * cout).sum(0).cpu()
    n = B * H * W
    assert torch.allclose(s[:cout], ref.sum((0, 1, 2)), rtol=1e-5, atol=2e-6 * sc * n)
    assert torch.allclose(s[cout:], (ref * ref).sum((0, 1, 2)), rtol=2e-6, atol=1e-9)
    dy = torch.randn(B, H, W, cout, generator=g)
    dref = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    dx = hip_backend.img_conv16(dy.to(DEV), wd, input_grad=True)
    assert (dx.cpu().double() - dref).abs().max().item() <= 2e-6 * dref.abs().max().item()
    xr = x.permute(0, 3, 1, 2).double()
    wref = torch.nn.grad.conv2d_weight(xr, (cout, 16, 3, 3), dy.permute(0, 3, 1, 2).double(), 1, 1)
    dW = hip_backend.img_conv16_wgrad(x.to(DEV), dy.to(DEV), wd)
    assert dW.stride() == wd.stride()
    assert (dW.cpu().double() - wref).abs().max().item() <= 5e-6 * wref.abs().max().item()
    assert torch.equal(dW, hip_backend.img_conv16_wgrad(x.to(DEV), dy.to(DEV), wd))              # fixed summation order


def test_conv16_rejects_other_shapes(hip_backend):
    with pytest.raises(RuntimeError):
        hip_backend.img_conv16(torch.zeros(1, 4, 4, 32, device=DEV), torch.zeros(16, 16, 3, 3, device=DEV))
    with pytest.raises(RuntimeError):
        hip_backend.img_conv16(torch.zeros(1, 4, 4, 16, device=DEV), torch.zeros(16, 16, 1, 1, device=DEV))
    with pytest.raises(RuntimeError):
        hip_backend.img_conv16(torch.zeros(1, 4, 4, 16, device=DEV), torch.zeros(64, 16, 3, 3, device=DEV))
    assert hip_backend.img_conv16(torch.zeros(0, 4, 4, 16, device=DEV), torch.zeros(16, 16, 3, 3, device=DEV)).shape == (0, 4, 4, 16)


def test_encoder_stack_with_and_without_conv16(hip_backend, monkeypatch):
    """RGB_net1 forward + backward through `_ImageCNN` with blocks 2-4 on image_conv16.hip against MIOpen's convolutions there
    (I2P_NO_CONV16=1), same weights"""
    from i2pnet_amd import ops
    from i2pnet_amd.modules import createCNNs
    prev = ops.set_backend(None)
    try:
        torch.manual_seed(4)
        net = createCNNs(3, [16, 16, 16, 16, 32], [2, 1, 1, 1, 2]).to(DEV).to(memory_format=torch.channels_last).train()
        x = torch.randn(2, 3, 75, 122, device=DEV)
        res = {}
        for tag, env in (("conv16", "0"), ("miopen", "1")):
            monkeypatch.setenv("I2P_NO_CONV16", env)
            state = {k: v.clone() for k, v in net.state_dict().items()}
            net.zero_grad(set_to_none=True)
            out = net(x)
            (out * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
            res[tag] = (out.detach().clone(), [p.grad.clone() if p.grad is not None else None for p in net.parameters()],
                        {k: v.clone() for k, v in net.state_dict().items()})
            net.load_state_dict(state)
        (o1, g1, s1), (o2, g2, s2) = res["conv16"], res["miopen"]
        assert (o1 - o2).abs().max().item() <= 2e-4 * o2.abs().max().item()
        for (n, _), a, b in zip(net.named_parameters(), g1, g2):
            assert (a is None) == (b is None), n
            if a is not None:       # (near-tie arg-max flips of the pooling windows behind re-route gradient: see test_image_first_gpu.py)
                assert (a - b).abs().max().item() <= 5e-2 * max(b.abs().max().item(), 1e-4), n
        for k in s1:
            assert torch.allclose(s1[k].float(), s2[k].float(), rtol=1e-4, atol=1e-5), k
    finally:
        ops.set_backend(prev)


@pytest.mark.parametrize("B,H,W", [(2, 41, 90), (1, 3, 12), (1, 7, 5), (3, 20, 131)])
def test_tail_backward_and_input_gradient_in_one_kernel(hip_backend, B, H, W):
    """i2p_img_block_bwd_stats + i2p_img_conv_tail_bwd against i2p_img_block_bwd + i2p_img_conv_bwd_data"""
    g = torch.Generator().manual_seed(B * 100 + W)
    yk = (torch.randn(B, H, W, 16, generator=g) * 1.5 + 0.2).to(DEV)
    gam, bet = torch.randn(16, generator=g).to(DEV), (torch.randn(16, generator=g) * 0.2).to(DEV)
    out, arg, mi = hip_backend.img_block_forward(yk, gam, bet, 1e-5, 0.1, 1)
    w = (torch.randn(16, 16, 3, 3, generator=g) * 0.2).to(DEV).contiguous(memory_format=torch.channels_last)
    gin = torch.randn(B, H, W, 16, generator=g).to(DEV)
    dy0, dg0, db0 = hip_backend.img_block_backward(gin, arg, yk, mi, gam, bet, 0.1, 1)
    dx0 = hip_backend.img_conv16(dy0, w, input_grad=True)
    dy1, dx1, dg1, db1 = hip_backend.img_conv16_tail_backward(gin, arg, yk, mi, gam, bet, 0.1, w)
    assert torch.equal(dy0, dy1)                                 # the same arithmetic in the same order
    assert (dx0 - dx1).abs().max().item() <= 1e-6 * max(dx0.abs().max().item(), 1e-6)
    assert torch.allclose(dg0, dg1, rtol=1e-6, atol=1e-6) and torch.allclose(db0, db1, rtol=1e-6, atol=1e-6)


