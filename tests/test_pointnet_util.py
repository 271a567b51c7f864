"""i2pnet_amd.pointnet_util (sample_and_group / PointNetSetAbstraction / index_points /
knn_point) against a golden vector produced by the REFERENCE pointnet_util.PointNetSetAbstraction
(tools/gen_golden.py sa; reference imported in the build container with the CPU oracle as its extension)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import synthetic_state

GOLD = Path(__file__).resolve().parent / "golden" / "set_abstraction.npz"


def _inputs(B, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(B, 3, N, generator=g) - 0.5) * 20.0
    pts = torch.randn(B, D, N, generator=g)
    return xyz, pts


def _run(device):
    from i2pnet_amd.pointnet_util import PointNetSetAbstraction
    gold = np.load(GOLD)
    B, N, D, S, K, seed = [int(v) for v in gold["meta"]]
    sa = PointNetSetAbstraction(npoint=S, radius=None, nsample=K, in_channel=3 + D, mlp=[16, 32], group_all=False)
    ours = {k: tuple(v.shape) for k, v in sa.state_dict().items()}
    theirs = {k: tuple(int(x) for x in s.split(",") if x) for k, s in zip(gold["state_keys"].tolist(), gold["state_shapes"].tolist())}
    assert ours == theirs                                            # a reference state_dict loads
    sa.load_state_dict(synthetic_state(list(theirs.items()), seed=seed))
    sa.to(device).train()
    xyz, pts = _inputs(B, N, D, seed)
    xyz, pts = xyz.to(device), pts.to(device).requires_grad_()
    new_xyz, new_points, grouped_xyz, fps_idx, _ = sa(xyz, pts)
    w = torch.randn(new_points.shape, generator=torch.Generator().manual_seed(seed + 1)).to(device)
    (new_points * w).sum().backward()
    assert torch.equal(fps_idx.cpu().long(), torch.from_numpy(gold["fps_idx"]))                   # FPS: bit-exact
    assert torch.equal(new_xyz.detach().cpu(), torch.from_numpy(gold["new_xyz"]))
    got_sorted = np.sort(grouped_xyz.detach().cpu().numpy().reshape(B, S, K * 3), axis=-1)        # neighbour SETS
    assert np.mean(got_sorted == gold["grouped_xyz_sorted"]) > 0.999
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
    assert rel(new_points.detach().cpu().numpy(), gold["new_points"]) < 1e-4
    assert rel(pts.grad.cpu().numpy(), gold["pts_grad"]) < 2e-3
    assert rel(sa.mlp_convs[0].weight.grad.cpu().numpy(), gold["w0_grad"]) < 2e-3
    assert rel(sa.mlp_bns[1].weight.grad.cpu().numpy(), gold["bn1_gamma_grad"]) < 2e-3
    assert rel(sa.mlp_bns[1].running_mean.cpu().numpy(), gold["running_mean1"]) < 1e-4
    assert rel(sa.mlp_bns[1].running_var.cpu().numpy(), gold["running_var1"]) < 1e-4
    sa.eval()
    with torch.no_grad():
        ev = sa(xyz, pts.detach())[1]
    assert rel(ev.cpu().numpy(), gold["new_points_eval"]) < 1e-4


def test_set_abstraction_vs_reference_cpu_oracle(oracle_backend):
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        _run("cpu")
    finally:
        ops.set_backend(prev)


@pytest.mark.gpu
def test_set_abstraction_vs_reference_gpu():
    _run("cuda")


@pytest.mark.gpu
def test_set_abstraction_small_range_sizes(oracle_backend, hip_backend):
    """the small-range model's first level at its real size (8192 -> 2048 centres, 32 neighbours): FPS and the
    kNN neighbour sets of the HIP kernels against the oracle (no 2048 x 8192 distance matrix is formed)."""
    from i2pnet_amd.pointnet_util import sample_and_group
    g = torch.Generator().manual_seed(3)
    B, N, S, K = 1, 8192, 2048, 32
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5) * 60.0
    feats = torch.randn(B, N, 4, generator=g)
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        r = sample_and_group(S, None, K, xyz, feats, returnfps=True)
    finally:
        ops.set_backend(prev)
    h = sample_and_group(S, None, K, xyz.cuda(), feats.cuda(), returnfps=True)
    assert torch.equal(r[3], h[3].cpu())                                        # fps_idx
    assert torch.equal(r[0], h[0].cpu())                                        # new_xyz
    assert torch.equal(r[1], h[1].cpu())                                        # grouped features incl. order
