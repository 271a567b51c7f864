"""Drop-in for the reference's compiled extension `pointnet2.pointnet2_cuda`
(pointnet2/src/pointnet2_api.cpp:10-24): copy into the reference's `pointnet2/` directory next
to `pointnet2_utils.py`, which then imports it unchanged (`import pointnet2.pointnet2_cuda as
pointnet2`, pointnet2_utils.py:9)."""
from i2pnet_amd import ops as _ops

_be = _ops.hip_backend()
ball_query_wrapper = _be.ball_query_wrapper
group_points_wrapper = _be.group_points_wrapper
group_points_grad_wrapper = _be.group_points_grad_wrapper
gather_points_wrapper = _be.gather_points_wrapper
gather_points_grad_wrapper = _be.gather_points_grad_wrapper
furthest_point_sampling_wrapper = _be.furthest_point_sampling_wrapper
three_nn_wrapper = _be.three_nn_wrapper
three_interpolate_wrapper = _be.three_interpolate_wrapper
three_interpolate_grad_wrapper = _be.three_interpolate_grad_wrapper


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    """called by pointnet2_utils.KNN (:32) but never bound upstream; provided here."""
    _be.knn(known, unknown, k, idx)
    import torch
    nb = torch.gather(known, 1, idx.long().reshape(b, n * k, 1).expand(-1, -1, 3)).reshape(b, n, k, 3)
    dist2.copy_(((nb - unknown.unsqueeze(2)) ** 2).sum(-1))
