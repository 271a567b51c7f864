"""Operator layer of the projection point-cloud branch.

Mirrors the function names and contracts of the reference's `src/projectPN/utils.py`
(`get_idx_cuda`, `get_sample_idx`, `get_stride_idx_cuda`, `gather_torch`, `get_neighbor_copy`,
`get_neighbor_att`, `check_valid`, `project_seq`, `grouping`, `square_distance`, `knn_point`,
`index_points_group`) on top of libi2p_ops.so.  What changes is how they execute:

* `gather_torch` is one HIP gather (and one scatter-add in backward) instead of an int64
  index expansion + `torch.gather` + two permutes (utils.py:36-60);
* `get_neighbor_*` no longer zero-fills the two `[B,N,kt,1]` float tensors the reference
  allocates and never writes (utils.py:91-92; 3.9 MB per call at level 1);
* `project_seq` is three small kernels with a deterministic duplicate-cell rule instead of a
  Python loop of `index_put_` (utils.py:173-177);
* `knn_point` never materialises the `[B,S,N]` distance matrix (utils.py:362-379).
"""
import numpy as np
import os

import torch
from torch.autograd import Function

from . import ops
from .fused_conv_select_k import FLAG_COPY, FLAG_FILL, FLAG_SHIFT, fused_conv_select_k
from .pointnet2_utils import grouping_operation

# ---------------------------------------------------------------------------------------------
# index grids (utils.py:8-33)
# ---------------------------------------------------------------------------------------------


def get_idx_cuda(B, H, W, device):
    """[B, H*W, 2] i32 of (h, w) for every cell."""
    return get_stride_idx_cuda(B, H, W, 1, 1, device)


_grid_cache = {}


def _cached(key, make):
    """constant index grids are built once per (shape, device): read-only for every consumer, and
    a training step re-creating them costs ~60 tiny launches"""
    t = _grid_cache.get(key)
    if t is None:
        t = _grid_cache[key] = make()
    return t


def get_stride_idx_cuda(B, out_h, out_w, stride_h, stride_w, device):
    """[B, out_h*out_w, 2] i32 of (h*stride_h, w*stride_w) (read-only: cached per shape and device)."""
    return _cached(("stride", B, out_h, out_w, stride_h, stride_w, str(device)),
                   lambda: _make_stride_idx(B, out_h, out_w, stride_h, stride_w, device))


def _make_stride_idx(B, out_h, out_w, stride_h, stride_w, device):
    h = torch.arange(0, out_h * stride_h, stride_h, device=device, dtype=torch.int32)
    w = torch.arange(0, out_w * stride_w, stride_w, device=device, dtype=torch.int32)
    grid = torch.stack(torch.meshgrid(h, w, indexing="ij"), dim=-1).reshape(1, out_h * out_w, 2)
    return grid.expand(B, -1, -1).contiguous()


def get_sample_idx(batch, out_h, out_w, stride_H, stride_W, device):
    """three [batch, out_h, out_w] i64 grids (b, h*stride_H, w*stride_W) (read-only: cached)."""
    return _cached(("sample", batch, out_h, out_w, stride_H, stride_W, str(device)),
                   lambda: _make_sample_idx(batch, out_h, out_w, stride_H, stride_W, device))


def _make_sample_idx(batch, out_h, out_w, stride_H, stride_W, device):
    h = torch.arange(0, out_h * stride_H, stride_H, device=device, dtype=torch.int64)
    w = torch.arange(0, out_w * stride_W, stride_W, device=device, dtype=torch.int64)
    b = torch.arange(batch, device=device, dtype=torch.int64)
    return (b.view(-1, 1, 1).expand(batch, out_h, out_w).contiguous(),
            h.view(1, -1, 1).expand(batch, out_h, out_w).contiguous(),
            w.view(1, 1, -1).expand(batch, out_h, out_w).contiguous())


# ---------------------------------------------------------------------------------------------
# gather on channel-last images (utils.py:36-60)
# ---------------------------------------------------------------------------------------------


class _GatherRows(Function):
    @staticmethod
    def forward(ctx, feat, h_idx, w_idx, width):
        # feat [B, HW, C] contiguous f32; h_idx/w_idx [B, Q] i64
        B, HW, C = feat.shape
        out = torch.empty(B, h_idx.shape[1], C, dtype=torch.float32, device=feat.device)
        ops.get_backend().gather_rows(feat, h_idx, w_idx, width, out)
        ctx.save_for_backward(h_idx, w_idx)
        ctx.shape = (B, HW, C, width)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h_idx, w_idx = ctx.saved_tensors
        B, HW, C, width = ctx.shape
        grad_feat = ops.zeros((B, HW, C), torch.float32, grad_out.device)
        ops.get_backend().gather_rows_grad(grad_out.contiguous(), h_idx, w_idx, width, grad_feat)
        return grad_feat, None, None, None


class _SaRows(Function):
    """grouped rows [xyz[cell] - centre, feat[cell], zero padding] of a set-abstraction / up-convolution MLP in one launch
    (reference: gather_torch x2 + subtraction + cat, PPBackbone_center.py:94-129, :236-262); gradient only w.r.t. the
    features (the coordinate images are data), scattered straight from the row gradient's feature columns."""

    @staticmethod
    def forward(ctx, xyz, centre, feat, h_idx, w_idx, K, width, cpad, xyz_col, feat_col):
        out = ops.get_backend().sa_rows(xyz, centre, feat, h_idx, w_idx, K, width, cpad, xyz_col, feat_col)
        ctx.save_for_backward(h_idx, w_idx)
        ctx.meta = (feat.shape, width, cpad, feat_col)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h_idx, w_idx = ctx.saved_tensors
        (B, HW, C), width, cpad, feat_col = ctx.meta
        grad_feat = ops.zeros((B, HW, C), torch.float32, grad_out.device)
        ops.get_backend().gather_rows_grad_ld(grad_out.contiguous(), cpad, feat_col, h_idx, w_idx, width, grad_feat)
        return None, None, grad_feat, None, None, None, None, None, None, None


class _KnnRows(Function):
    """rows [xyz, pix_xyz[idx], pts * pix[idx], padding] of the fine cost volume's kNN pi-stage in one launch (reference: two
    index_points_group gathers, a product and a cat, PPBackbone_center.py:369-395).  Gradients: point coordinates and features
    (sums over the K neighbours), pixel features (the per-neighbour products scattered with the fixed-point row scatter); the
    pixel rays are data."""

    @staticmethod
    def forward(ctx, xyz, pix_xyz, pts, pix, idx, K, cpad):
        out = ops.get_backend().knn_rows(xyz, pix_xyz, pts, pix, idx, K, cpad)
        ctx.save_for_backward(pts, pix, idx)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, g):
        pts, pix, idx = ctx.saved_tensors
        be = ops.get_backend()
        d_xyz, d_pts, gq = be.knn_rows_backward(g.contiguous(), pts, pix, idx, ctx.K, ctx.needs_input_grad[0])
        d_pix = None
        if ctx.needs_input_grad[3]:
            B, M, C = pix.shape
            d_pix = ops.zeros((B, M, C), torch.float32, g.device)
            h = _cached(("zero_rows", B, idx.shape[1], str(g.device)), lambda: torch.zeros(B, idx.shape[1], dtype=torch.long, device=g.device))
            be.gather_rows_grad(gq, h, idx, M, d_pix)
        return d_xyz, None, d_pts, d_pix, None, None, None


class _PcRows(Function):
    """the pc-stage front end of a cost volume in one launch each way (csrc/sa_group.hip i2p_pc_rows_fwd/bwd; reference:
    PPBackbone_center.py:443-476) -> (geo [B,N,K,12], part [B,N,K,C+c], nb_feat [B,N,K,c])"""

    @staticmethod
    def forward(ctx, xyz, pts, feat, h_idx, w_idx, K, W):
        geo, part, nbf = ops.get_backend().pc_rows_forward(xyz, pts, feat, h_idx, w_idx, K, W)
        ctx.save_for_backward(xyz, h_idx, w_idx)
        ctx.dims = (K, W, pts.shape[2], feat.shape[2])
        B, HW = xyz.shape[0], xyz.shape[1]
        return geo.view(B, HW, K, 12), part.view(B, HW, K, -1), nbf.view(B, HW, K, -1)

    @staticmethod
    def backward(ctx, g_geo, g_part, g_nbf):
        xyz, h_idx, w_idx = ctx.saved_tensors
        K, W, Cp, cf = ctx.dims
        B, HW = xyz.shape[0], xyz.shape[1]
        c = lambda t, w_: None if t is None else t.reshape(B, HW * K, w_).contiguous()
        if g_part is None:
            g_part = torch.zeros(B, HW * K, Cp + cf, dtype=torch.float32, device=xyz.device)
        d_pts, comb = ops.get_backend().pc_rows_backward(xyz, h_idx, w_idx, K, W, Cp, cf, c(g_geo if ctx.needs_input_grad[0] else None, 12),
                                                         c(g_part, Cp + cf), c(g_nbf, cf))
        d_xyz = comb[:, :, cf:cf + 3] if ctx.needs_input_grad[0] else None
        return d_xyz, d_pts, comb[:, :, :cf], None, None, None, None


def pc_rows_fusable(xyz, pts, feat):
    be = ops.get_backend()
    return (be.device_type == "cuda" and be.name == "hip" and xyz.dtype == torch.float32
            and pts.dtype == torch.float32 and feat.dtype == torch.float32 and pts.shape[-1] % 4 == 0 and feat.shape[-1] % 4 == 0)


def pc_rows(xyz, pts, feat, h_idx, w_idx, K, W):
    """xyz [B,HW,3], pts [B,HW,C], feat [B,HW,c], h_idx / w_idx [B,HW,K(,1)] -> (geo, part, nb_feat), see _PcRows"""
    B = xyz.shape[0]
    return _PcRows.apply(xyz.contiguous(), pts.contiguous(), feat.contiguous(), h_idx.reshape(B, -1).contiguous(), w_idx.reshape(B, -1).contiguous(), K, W)


def knn_rows_fusable(xyz, pix_xyz, pts, pix):
    be = ops.get_backend()
    return (be.device_type == "cuda" and be.name == "hip" and not pix_xyz.requires_grad
            and pts.dtype == torch.float32 and pix.dtype == torch.float32 and xyz.dtype == torch.float32)


def knn_rows(xyz, pix_xyz, pts, pix, idx, cpad):
    """xyz [B,N,3], pix_xyz [B,M,3], pts [B,N,C], pix [B,M,C], idx [B,N,K] -> [B,N,K,cpad]"""
    B, N, K = idx.shape
    out = _KnnRows.apply(xyz.contiguous(), pix_xyz.contiguous(), pts.contiguous(), pix.contiguous(),
                         idx.reshape(B, N * K).long().contiguous(), K, cpad)
    return out.view(B, N, K, cpad)


def padded_width(c, pow2):
    """channel count after modules.cat_padded: multiple of 4, or 16/32/64/128 for gradient-carrying inputs <= 128"""
    if pow2 and c <= 128:
        return next(w for w in (16, 32, 64, 128) if w >= c)
    return c + (-c) % 4


def sa_rows_fusable(xyz_img, centre, feature):
    be = ops.get_backend()
    return (be.device_type == "cuda" and be.name == "hip" and feature is not None and feature.dtype == torch.float32
            and not xyz_img.requires_grad and not centre.requires_grad)


def sa_rows(xyz_img, centre, feature, h_idx, w_idx, K, width, xyz_first=True, pow2=True):
    """xyz_img [B,H,W,3], centre [B,N,3] (or [B,h,w,3]), feature [B,H,W,C], h_idx / w_idx [B,N,K(,1)] ->
    [B, N, K, cpad] = cat([xyz[cell] - centre, feature[cell]]) (xyz_first) or cat([feature[cell], xyz[cell] - centre]), zero padded"""
    B = xyz_img.shape[0]
    C = feature.shape[-1]
    cpad = padded_width(3 + C, pow2)
    N = h_idx.reshape(B, -1).shape[1] // K
    out = _SaRows.apply(xyz_img.reshape(B, -1, 3).contiguous(), centre.reshape(B, N, 3).contiguous(),
                        feature.reshape(B, -1, C).contiguous(), h_idx.reshape(B, -1).contiguous(), w_idx.reshape(B, -1).contiguous(),
                        K, width, cpad, 0 if xyz_first else C, 3 if xyz_first else 0)
    return out.view(B, N, K, cpad)


def gather_torch(feature, neigh_b_idx, neigh_h_idx, neigh_w_idx, batch, height, width):
    """feature [B,H,W,C] (any shape that reshapes to it), neigh_{h,w}_idx [B,H',W'] i64 ->
    [B,H',W',C] = feature[b, h, w, :].  `neigh_b_idx` is ignored, as in the reference
    (utils.py:48 uses only h*width+w)."""
    nei_h, nei_w = neigh_h_idx.shape[1:3]
    feat = feature.reshape(batch, height * width, -1)
    if feat.dtype != torch.float32:
        feat = feat.float()
    h = neigh_h_idx.reshape(batch, -1).contiguous()
    w = neigh_w_idx.reshape(batch, -1).contiguous()
    out = _GatherRows.apply(feat.contiguous(), h, w, width)
    return out.reshape(batch, nei_h, nei_w, -1)


# ---------------------------------------------------------------------------------------------
# neighbour selection (utils.py:63-103, :253-293)
# ---------------------------------------------------------------------------------------------

_arange_cache = {}


def _window_order(kt, device):
    key = (kt, str(device))
    t = _arange_cache.get(key)
    if t is None:
        t = torch.arange(0, kt, device=device, dtype=torch.int32)     # utils.py:84
        _arange_cache[key] = t
    return t


def _get_neighbor(xyz1_proj, xyz2_proj, idx_n2, kernel_shape, knn_points, stride_h, stride_w,
                  distance, flag):
    batch, height, width, _ = xyz1_proj.shape
    small_h, small_w = xyz2_proj.shape[1], xyz2_proj.shape[2]
    kt = kernel_shape[0] * kernel_shape[1]
    n_points = idx_n2.shape[1]
    dev = xyz1_proj.device
    random_hw = _window_order(kt, dev)
    # FLAG_FILL: the operator writes every slot, so no zero-fill of the outputs (reference: torch.zeros, utils.py:77-82)
    sel = torch.empty(3, batch, n_points, knn_points, 1, device=dev, dtype=torch.long)
    mask = torch.empty(batch, n_points, knn_points, 1, device=dev, dtype=torch.float32)
    # never written by the operator (reference allocates [B,N,kt,1] zeros for both)
    unused = _cached(("unused", str(dev)), lambda: torch.zeros(1, device=dev, dtype=torch.float32))
    fused_conv_select_k(xyz1_proj.contiguous(), xyz2_proj.contiguous(), idx_n2.contiguous(), random_hw,
                        height, width, n_points, kernel_shape[0], kernel_shape[1], knn_points, flag | FLAG_FILL,
                        distance, stride_h, stride_w, sel[0], sel[1], sel[2], unused, unused, mask,
                        small_h, small_w)
    return sel[0].squeeze(-1), sel[1].squeeze(-1), sel[2].squeeze(-1), mask


def get_neighbor_copy(xyz1_proj, xyz2_proj, idx_n2, kernel_shape, knn_points, stride_h=1, stride_w=1,
                      distance=10):
    """K nearest window cells of `xyz2_proj` for the query cells `idx_n2` of `xyz1_proj`;
    empty slots repeat the nearest hit (FLAG_SHIFT|FLAG_COPY).  -> (b,h,w idx [B,N,K] i64, mask [B,N,K,1])."""
    return _get_neighbor(xyz1_proj, xyz2_proj, idx_n2, kernel_shape, knn_points, stride_h, stride_w,
                         distance, FLAG_SHIFT | FLAG_COPY)


def get_neighbor_att(xyz1_proj, xyz2_proj, idx_n2, kernel_shape, knn_points, stride_h=1, stride_w=1,
                     distance=10):
    """Same search, empty slots stay (0,0) with mask 0 (FLAG_SHIFT only)."""
    return _get_neighbor(xyz1_proj, xyz2_proj, idx_n2, kernel_shape, knn_points, stride_h, stride_w,
                         distance, FLAG_SHIFT)


def check_valid(xyz):
    """1.0 where the point is not the all-zero "empty cell" marker (utils.py:106-108)."""
    be = ops.get_backend()
    if be.name == "hip" and xyz.is_cuda and xyz.dtype == torch.float32:
        return be.row_valid(xyz.detach())                  # one launch instead of ne + any + cast
    return torch.any(torch.ne(xyz, 0), dim=-1, keepdim=True).float()


# ---------------------------------------------------------------------------------------------
# spherical projection (utils.py:111-187)
# ---------------------------------------------------------------------------------------------


class _ProjectSeq(torch.autograd.Function):
    """The scatter of `project_seq` with the gradient of the reference's `index_put_` formulation (utils.py:173-177):
    the cells are computed without gradient (utils.py:140), the scattered VALUES are differentiable — a cell's
    gradient goes back to the point that won it (every point wins at most one cell, so the backward is a plain
    gather along the winner map, no accumulation)."""

    @staticmethod
    def forward(ctx, H, W, fup, fdown, xyz, *feats):
        out_xyz, outs, winner = ops.get_backend().project_seq(xyz.detach(), [f.detach() for f in feats], H, W, fup, fdown)
        ctx.save_for_backward(winner)
        ctx.n = xyz.shape[1]
        ctx.mark_non_differentiable(winner)
        return (out_xyz, winner, *outs)

    @staticmethod
    def backward(ctx, g_xyz, _g_winner, *g_feats):
        winner, = ctx.saved_tensors                          # [B, H*W] i32, -1 = empty cell
        B, HW = winner.shape
        valid = winner >= 0
        idx = winner.clamp_min(0).long()

        def back(g):
            if g is None:
                return None
            g = g.reshape(B, HW, -1) * valid.unsqueeze(-1)
            out = g.new_zeros(B, ctx.n, g.shape[-1])
            return out.scatter_add_(1, idx.unsqueeze(-1).expand(-1, -1, g.shape[-1]), g)
        return (None, None, None, None, back(g_xyz), *[back(g) for g in g_feats])


def project_seq(xyz, features, H, W, use_rank=True, fup=2.0, fdown=-24.8):
    """xyz [B,N,3], features list of [B,N,D] -> (xyz_proj [B,H,W,3], [feature_proj [B,H,W,D]]).

    Cell of a point: col = trunc((pi - atan2(y,x)) / (2*pi/W)), row = H - trunc(asin(z/r)/dv + off),
    clamped.  When several points share a cell the highest point index wins (with `use_rank`
    the cloud is first ordered by decreasing range, so the nearest point wins)."""
    xyz = xyz.float().contiguous()
    feats = [f.float().contiguous() for f in features]
    if use_rank:
        with torch.no_grad():
            rank = torch.argsort(torch.norm(xyz, p=2, dim=2), dim=1, descending=True)  # utils.py:159
        xyz = torch.gather(xyz, 1, rank[:, :, None].expand(-1, -1, 3)).contiguous()
        feats = [torch.gather(f, 1, rank[:, :, None].expand(-1, -1, f.shape[-1])).contiguous() for f in feats]
    if torch.is_grad_enabled() and (xyz.requires_grad or any(f.requires_grad for f in feats)):
        res = _ProjectSeq.apply(H, W, fup, fdown, xyz, *feats)      # learned inputs: differentiable scatter
        return res[0], list(res[2:])
    with torch.no_grad():
        out_xyz, outs, _ = ops.get_backend().project_seq(xyz, feats, H, W, fup, fdown)
    return out_xyz, outs


# ---------------------------------------------------------------------------------------------
# kNN grouping in the normalised image plane (utils.py:313-393)
# ---------------------------------------------------------------------------------------------


def square_distance(src, dst):
    """[B,N,C] x [B,M,C] -> [B,N,M] squared distances (expanded form, utils.py:343-364)."""
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).unsqueeze(-1)
    dist += torch.sum(dst ** 2, -1).unsqueeze(1)
    return dist


def knn_point(nsample, xyz, new_xyz):
    """indices [B,S,nsample] (i64) of the nsample nearest `xyz` points of every `new_xyz` point.
    Ordered by (distance, index); the reference's topk(sorted=False) order is unspecified."""
    B, S, _ = new_xyz.shape
    idx = torch.empty(B, S, nsample, dtype=torch.int32, device=xyz.device)
    with torch.no_grad():
        ops.get_backend().knn(xyz.detach().float().contiguous(), new_xyz.detach().float().contiguous(),
                              nsample, idx)
    return idx.long()


def index_points_group(points, knn_idx):
    """points [B,N,C], knn_idx [B,S,K] -> [B,S,K,C] (utils.py:382-393).  The reference transposes to [B,C,N] for
    `group_points` and back; here the channel-last row gather (and its run-merging scatter-add backward, which
    adds C contiguous floats per row instead of one strided float per (channel, sample)) does it in place."""
    B, N, C = points.shape
    S, K = knn_idx.shape[1], knn_idx.shape[2]
    w = knn_idx.reshape(B, S * K).long().contiguous()
    h = _cached(("zero_rows", B, S * K, str(points.device)), lambda: torch.zeros(B, S * K, dtype=torch.long, device=points.device))
    pts = points if points.dtype == torch.float32 else points.float()
    return _GatherRows.apply(pts.contiguous(), h, w, N).view(B, S, K, C)


def grouping(feature, K, src_xyz, q_xyz, use_xyz=False):
    """-> grouped_xyz [B,S,K,3], xyz_diff [B,S,K,3], new_points [B,S,K,C(+3)], idx [B,S,K]"""
    q_xyz = q_xyz.contiguous()
    src_xyz = src_xyz.contiguous()
    point_indices = knn_point(K, src_xyz, q_xyz)
    grouped_xyz = index_points_group(src_xyz, point_indices)
    xyz_diff = grouped_xyz - q_xyz.unsqueeze(2)
    grouped_feature = index_points_group(feature, point_indices)
    new_points = torch.cat([xyz_diff, grouped_feature], dim=-1) if use_xyz else grouped_feature
    return grouped_xyz, xyz_diff, new_points, point_indices
