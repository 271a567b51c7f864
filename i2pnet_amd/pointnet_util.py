"""Point-set abstraction for the small-range model on the HIP operator layer.

Mirror of the reference's `pointnet_util.py` (`square_distance` :36-58, `index_points` :60-78, `knn_point`
:112-123, `sample_and_group` :165-233, `sample_and_group_all` :236-254, `PointNetSetAbstraction` :257-314):
same names, arguments, return tuples and parameter names (`mlp_convs.{i}`, `mlp_bns.{i}` => a reference
`state_dict` loads).  What runs underneath:

* furthest point sampling: `i2p_furthest_point_sampling` (the call the reference makes at :183,
  `FurthestPointSampling.forward(None, xyz.contiguous(), npoint)`), bit-exact tie rule;
* kNN: `i2p_knn` — one wave per query scanning the cloud, no [B,S,N] distance matrix (67 MB per sample at
  2048 x 8192 in the reference's matmul + topk); neighbours come back ordered by (distance, index) where the
  reference's `topk(sorted=False)` order is unspecified (every consumer is order-invariant: max over the group);
* gathers: the channel-last row gather `i2p_gather_rows` with scatter-add backward;
* 1x1 conv + BatchNorm2d + ReLU on the channel-last `[B,S,K,C]` view: one GEMM per layer and the fused
  BN/activation kernels (fp64 batch statistics); the BatchNorm2d running buffers are updated as torch does
  (momentum, unbiased variance, conv bias included in the mean) and used in eval mode.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import modules, ops
from . import projectpn as P
from .fused import linear
from .modules import bn_act_running
from .pointnet2_utils import FurthestPointSampling


def square_distance(src, dst):
    """[B,N,C] x [B,M,C] -> [B,N,M] squared distances in the expanded form (pointnet_util.py:36-58)."""
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).unsqueeze(-1)
    dist += torch.sum(dst ** 2, -1).unsqueeze(1)
    return dist


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] (i64) -> [B,S,C] / [B,S,K,C] (pointnet_util.py:60-78)."""
    B, N, C = points.shape
    flat = idx.reshape(B, -1).long().contiguous()
    h = P._cached(("zero_rows", B, flat.shape[1], str(points.device)),
                  lambda: torch.zeros(B, flat.shape[1], dtype=torch.long, device=points.device))
    pts = points if points.dtype == torch.float32 else points.float()
    out = P._GatherRows.apply(pts.contiguous(), h, flat, N)
    return out.view(*idx.shape, C)


def knn_point(nsample, xyz, new_xyz):
    """indices [B,S,nsample] (i64) of the nsample nearest `xyz` points of every `new_xyz` point
    (pointnet_util.py:112-123), ordered by (distance, index)."""
    return P.knn_point(nsample, xyz, new_xyz)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, sample_idx=None, raw_feat_point=False,
                     raw_xyz=None, feat_mode=None):
    """xyz [B,N,3], points [B,N,D] or None -> new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D]
    (+ grouped_xyz, fps_idx, new_raw_xyz with returnfps) — pointnet_util.py:165-233."""
    B, N, C = xyz.shape
    S = npoint
    if sample_idx is not None:
        fps_idx = sample_idx
    else:
        fps_idx = FurthestPointSampling.forward(None, xyz.contiguous(), npoint).long()     # :183
    new_xyz = index_points(xyz, fps_idx)
    new_raw_xyz = index_points(raw_xyz, fps_idx) if raw_feat_point else None
    idx = knn_point(nsample, xyz, new_xyz)
    if raw_feat_point:
        grouped_xyz = index_points(raw_xyz, idx)
        centre = new_raw_xyz
    else:
        grouped_xyz = index_points(xyz, idx)
        centre = new_xyz
    grouped_xyz_norm = grouped_xyz - centre.view(B, S, 1, C)
    if feat_mode == "dim10feat":
        dist = torch.norm(grouped_xyz_norm, p=2, dim=3, keepdim=True)
        new_points = torch.cat([grouped_xyz_norm, centre.view(B, S, 1, C).expand(-1, -1, nsample, -1), grouped_xyz, dist], -1)
    elif feat_mode == "dist":
        new_points = torch.norm(grouped_xyz_norm, p=2, dim=3, keepdim=True)
    elif points is not None:
        new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], -1)
    else:
        new_points = grouped_xyz_norm
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx, new_raw_xyz
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """one group holding the whole cloud (pointnet_util.py:236-254)"""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, C)
    new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1) if points is not None else grouped_xyz
    return new_xyz, new_points


class _ConvBnView:
    """what `fused.mlp_stack` reads of a layer, for the reference's separate `mlp_convs[i]` / `mlp_bns[i]` modules"""

    def __init__(self, conv, bn):
        self.conv, self.bn_linear = conv, bn
        self.bn, self.activation_fn, self.negative_slope = True, True, 0.0          # BatchNorm2d + ReLU (:299)
        self.in_channels, self.out_channels = conv.in_channels, conv.out_channels

    def weight2d(self):
        return self.conv.weight.view(self.out_channels, self.in_channels)

    def __call__(self, x):                                   # layer outside the fused kernels' shape limits
        y = linear(x, self.weight2d())
        return bn_act_running(y, self.conv.bias, self.bn_linear, 0.0)


class PointNetSetAbstraction(nn.Module):
    """FPS -> kNN grouping -> (1x1 conv + BatchNorm2d + ReLU) x len(mlp) -> max over the group
    (pointnet_util.py:257-314).  Inputs/outputs channel-major like the reference: xyz [B,3,N], points [B,D,N]
    -> new_xyz [B,3,S], new_points [B,D',S], grouped_xyz [B,S,K,3], fps_idx [B,S], new_raw_xyz."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def _layer(self, x, conv, bn):
        """x [..., Cin] channel-last -> relu(bn(conv(x))) [..., Cout]"""
        W = conv.weight.view(conv.out_channels, conv.in_channels)
        if self.training or not bn.track_running_stats:
            y = linear(x, W)                            # the bias cancels in the batch-statistics BN
            return bn_act_running(y, conv.bias, bn, 0.0)
        y = F.linear(x, W, conv.bias)
        y = (y - bn.running_mean) * (torch.rsqrt(bn.running_var + bn.eps) * bn.weight) + bn.bias
        return F.relu(y)

    def forward(self, xyz, points, sample_idx=None, feat_mode=None, raw_feat_point=False, raw_xyz=None):
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        grouped_xyz, fps_idx, new_raw_xyz = [], [], None
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)
        else:
            new_xyz, new_points, grouped_xyz, fps_idx, new_raw_xyz = sample_and_group(
                self.npoint, self.radius, self.nsample, xyz, points, returnfps=True, sample_idx=sample_idx,
                raw_feat_point=raw_feat_point, raw_xyz=raw_xyz, feat_mode=feat_mode)
        x = new_points                                   # [B,S,K,C] stays channel-last (reference: permute to [B,C,K,S])
        if self.training and modules.USE_FUSED_MLP:
            # fused layer kernels (BN + ReLU of the previous layer on load, statistics in the epilogue, max over the
            # group fused into the last BN/activation pass); running buffers updated from the chain's statistics
            stack = [_ConvBnView(conv, bn) for conv, bn in zip(self.mlp_convs, self.mlp_bns)]
            pooled = modules.run_stack(modules.cat_padded([x]), stack, pool_k=x.shape[2])     # [B,S,D']
            new_points = pooled.permute(0, 2, 1)
        else:
            for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                x = self._layer(x, conv, bn)
            new_points = torch.max(x, 2)[0].permute(0, 2, 1)    # [B,D',S]
        return new_xyz.permute(0, 2, 1), new_points, grouped_xyz, fps_idx, (new_raw_xyz if raw_feat_point else None)
