"""PointNet++ operator layer on libi2p_ops.so.

Same public names, argument order and autograd behaviour as the reference's
`pointnet2/pointnet2_utils.py:11-321` (autograd Functions whose index-producing members have
no gradient and whose gather-type members scatter-add their gradient), so code written against
that module runs unchanged.  Differences: outputs are allocated on the input's device instead
of `torch.cuda.*Tensor`, and `KNN` works (upstream calls an unbound `knn_wrapper`,
pointnet2_utils.py:32 vs pointnet2_api.cpp:10-24).
"""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops


def _need_contiguous(*tensors):
    for t in tensors:
        assert t.is_contiguous()          # pointnet2_utils.py:24-25,52,80-81,...


class KNN(Function):
    @staticmethod
    def forward(ctx, k: int, unknown: torch.Tensor, known: torch.Tensor):
        """unknown [B,N,3], known [B,M,3] -> (dist [B,N,k] L2, idx [B,N,k] i32), ascending."""
        _need_contiguous(unknown, known)
        B, N, _ = unknown.size()
        idx = torch.empty(B, N, k, dtype=torch.int32, device=unknown.device)
        ops.get_backend().knn(known, unknown, k, idx)
        nb = torch.gather(known, 1, idx.long().reshape(B, N * k, 1).expand(-1, -1, 3)).reshape(B, N, k, 3)
        dist = torch.sqrt(torch.sum((nb - unknown.unsqueeze(2)) ** 2, dim=-1))
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz [B,N,3] -> [B,npoint] i32, idx[:,0] = 0 (pointnet2_utils.py:41-60).
        Also called statically with ctx=None (pointnet_util.py:183)."""
        _need_contiguous(xyz)
        B, N, _ = xyz.size()
        output = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        ops.get_backend().furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features [B,C,N], idx [B,npoint] i32 -> [B,C,npoint]"""
        _need_contiguous(features, idx)
        B, npoint = idx.size()
        _, Cn, N = features.size()
        output = torch.empty(B, Cn, npoint, dtype=torch.float32, device=features.device)
        ops.get_backend().gather_points_wrapper(B, Cn, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, Cn, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, Cn, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = ops.zeros((B, Cn, N), torch.float32, grad_out.device)
        ops.get_backend().gather_points_grad_wrapper(B, Cn, N, npoint, grad_out.contiguous(), idx,
                                                     grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown [B,N,3], known [B,M,3] -> (dist [B,N,3] L2, idx [B,N,3] i32)"""
        _need_contiguous(unknown, known)
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)
        ops.get_backend().three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        return torch.sqrt(dist2), idx                      # pointnet2_utils.py:129

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features [B,C,M], idx [B,n,3] i32, weight [B,n,3] -> [B,C,n]"""
        _need_contiguous(features, idx, weight)
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        ops.get_backend().three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = ops.zeros((B, c, m), torch.float32, grad_out.device)
        ops.get_backend().three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight,
                                                         grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features [B,C,N], idx [B,npoint,nsample] i32 -> [B,C,npoint,nsample]"""
        _need_contiguous(features, idx)
        B, nfeatures, nsample = idx.size()
        _, Cn, N = features.size()
        output = torch.empty(B, Cn, nfeatures, nsample, dtype=torch.float32, device=features.device)
        ops.get_backend().group_points_wrapper(B, Cn, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, Cn, npoint, nsample = grad_out.size()
        grad_features = ops.zeros((B, Cn, N), torch.float32, grad_out.device)
        ops.get_backend().group_points_grad_wrapper(B, Cn, N, npoint, nsample, grad_out.contiguous(), idx,
                                                    grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz [B,N,3], new_xyz [B,npoint,3] -> idx [B,npoint,nsample] i32"""
        _need_contiguous(new_xyz, xyz)
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros(B, npoint, nsample, dtype=torch.int32, device=xyz.device)
        ops.get_backend().ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball query + grouping, optionally prefixed by the centred xyz (pointnet2_utils.py:262-295)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features


class GroupAll(nn.Module):
    """whole cloud as one group (pointnet2_utils.py:298-321)."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features
