"""Host side of the two-stream step (i2pnet_amd/model.py; the device side: tests/test_two_streams_gpu.py): what can be checked without a GPU."""
import copy

import pytest
import torch


def test_chains_off_nests_and_unwinds():
    from i2pnet_amd import ops
    assert ops._CHAINS_OFF[0] == 0
    with ops.chains_off():
        assert ops._CHAINS_OFF[0] == 1
        with pytest.raises(RuntimeError):
            with ops.chains_off():
                assert ops._CHAINS_OFF[0] == 2
                raise RuntimeError("inside")
        assert ops._CHAINS_OFF[0] == 1
    assert ops._CHAINS_OFF[0] == 0


def test_branch_join_is_the_identity_with_gradients():
    from i2pnet_amd.model import _BranchJoin
    a = torch.randn(2, 5, 3, requires_grad=True)
    b = torch.randn(2, 5, 4, requires_grad=True)
    ap = a.permute(0, 2, 1)                       # (the image features arrive as a permuted view)
    x, y = _BranchJoin.apply(ap, b)
    assert torch.equal(x, ap) and torch.equal(y, b) and x.stride() == ap.stride()
    (x.sum() * 2 + (y * y).sum()).backward()
    assert torch.equal(a.grad, torch.full_like(a, 2.0)) and torch.equal(b.grad, 2 * b.detach())
    # only one of the two gets a gradient: the other's arrives as zeros (autograd materialises it) and passes through
    a.grad = b.grad = None
    x, y = _BranchJoin.apply(a, b)
    y.sum().backward()
    assert (a.grad is None or not a.grad.any()) and torch.equal(b.grad, torch.ones_like(b))


def test_the_cpu_model_has_no_second_stream_and_copies_without_stream_state(monkeypatch):
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.model import RegNet_v2
    net = RegNet_v2(cfg=cfg)
    assert net._branch_stream(torch.device("cpu")) is None
    monkeypatch.setenv("I2P_ONE_STREAM", "1")
    assert net._branch_stream(torch.device("cuda", 0)) is None       # (no device call is made for the answer)
    net.__dict__["_side_stream"] = object()
    net.__dict__["_lidar_event"] = object()
    twin = copy.deepcopy(net)
    assert "_side_stream" not in twin.__dict__ and "_lidar_event" not in twin.__dict__
    assert [k for k, _ in twin.named_parameters()] == [k for k, _ in net.named_parameters()]
