"""Quaternion algebra used by the coarse-to-fine registration (reference:
src/modules/warp_utils.py:10-55, :78-94).  Device-agnostic (the reference hard-codes .cuda())."""
import os

import torch

from . import ops


class _QuatMul(torch.autograd.Function):
    """Hamilton product as one HIP launch (csrc/projection_ops.hip `quat_mul_kernel`); the product is
    bilinear, so both gradients are the same kernel with one operand conjugated."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.get_backend().quat_mul(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        be = ops.get_backend()
        g = g.contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = be.quat_mul(g, b, conj_b=True)                 # dL/da = g (x) conj(b)
            if a.shape[1] == 1 and ga.shape[1] != 1:
                ga = ga.sum(1, keepdim=True)
        if ctx.needs_input_grad[1]:
            gb = be.quat_mul(a, g, conj_a=True)                 # dL/db = conj(a) (x) g
            if b.shape[1] == 1 and gb.shape[1] != 1:
                gb = gb.sum(1, keepdim=True)
        return ga, gb


class _QuatUnit(torch.autograd.Function):
    """mode 0: conj(q) / (|q|^2 + 1e-10); mode 1: q / (sqrt(|q|^2 + 1e-10) + 1e-10) — one launch forward, one
    backward (csrc/projection_ops.hip `quat_unit_*_kernel`) instead of 5-6 and 10-12 elementwise kernels."""

    @staticmethod
    def forward(ctx, q, mode):
        q = q.contiguous()
        ctx.save_for_backward(q)
        ctx.mode = mode
        return ops.get_backend().quat_unit_forward(mode, q)

    @staticmethod
    def backward(ctx, g):
        (q,) = ctx.saved_tensors
        return ops.get_backend().quat_unit_backward(ctx.mode, q, g.contiguous()), None


def normalise_q(q):
    """q [...,4] -> q / (sqrt(|q|^2 + 1e-10) + 1e-10) (PPBackbone_center.py:562)"""
    if q.dtype != torch.float32:
        return q / (torch.sqrt(torch.sum(q * q, dim=-1, keepdim=True) + 1e-10) + 1e-10)
    return _QuatUnit.apply(q, 1)


_CONJ_SIGN = {}


def _conj_sign(device):
    """(1,-1,-1,-1) on `device`, created once (the first, eager, call — never inside a hipGraph capture)."""
    s = _CONJ_SIGN.get(device)
    if s is None:
        s = _CONJ_SIGN[device] = torch.tensor([1.0, -1.0, -1.0, -1.0], device=device)
    return s


def inv_q(q):
    """q [B,4] or [B,1,4] (w,x,y,z) -> conj(q) / (|q|^2 + 1e-10), [B,4]"""
    B = q.shape[0]
    q = q.reshape(B, 4)
    if q.dtype == torch.float32:
        return _QuatUnit.apply(q, 0)
    n2 = torch.sum(q * q, dim=-1, keepdim=True) + 1e-10
    return (q * _conj_sign(q.device)) / n2


def mul_q(a, b):
    """Hamilton product, a/b [B,N|1,4] or [B,4] -> [B,N,4]"""
    if a.ndim == 2:
        a = a.unsqueeze(1)
    if b.ndim == 2:
        b = b.unsqueeze(1)
    return _QuatMul.apply(a, b)


def warp_quat_xyz(lidar_xyz, Hi_quat, H_trans):
    """p' = q [0,p] q^-1 + t for p [B,N,3], q [B,4], t [B,4] = [0,tx,ty,tz] -> [B,N,3]"""
    B, N, _ = lidar_xyz.shape
    homo = torch.cat([ops.zero_scalar(lidar_xyz.device, lidar_xyz.dtype).expand(B, N, 1), lidar_xyz], -1)
    homo = mul_q(mul_q(Hi_quat, homo), inv_q(Hi_quat)) + H_trans.reshape(B, 1, 4)
    return homo[:, :, 1:4]


class _WarpSplit(torch.autograd.Function):
    """warp_quat_xyz(p, q, t) * valid, then (uv, z, xyz) = (p'/(z+1e-10), p'_z, uv*z) in one launch each way
    (csrc/projection_ops.hip i2p_warp_split_fwd/bwd); gradients reach q and t (the cloud and the mask are data)."""

    @staticmethod
    def forward(ctx, p, q, t, valid):
        be = ops.get_backend()
        B, N, _ = p.shape
        p, q, t = p.detach().contiguous(), q.detach().contiguous(), t.detach().contiguous()
        v = valid.detach().reshape(B, N).contiguous() if valid is not None else None
        uv = torch.empty(B, N, 3, dtype=torch.float32, device=p.device); z = torch.empty(B, N, 1, dtype=torch.float32, device=p.device)
        xyz = torch.empty(B, N, 3, dtype=torch.float32, device=p.device)
        P = lambda x: be._p(x, torch.float32, "warp_split") if x is not None else None
        be._call("i2p_warp_split_fwd", int(B), int(N), P(p), P(q), P(t), P(v), P(uv), P(z), P(xyz), stream=be._stream())
        ctx.save_for_backward(p, q, t, v if v is not None else p.new_empty(0))
        ctx.has_valid = v is not None
        return uv, z, xyz

    @staticmethod
    def backward(ctx, g_uv, g_z, g_xyz):
        p, q, t, v = ctx.saved_tensors
        be = ops.get_backend()
        B, N, _ = p.shape
        dq = torch.empty(B, 4, dtype=torch.float32, device=p.device); dt = torch.empty(B, 4, dtype=torch.float32, device=p.device)
        P = lambda x: be._p(x.contiguous(), torch.float32, "warp_split") if x is not None else None
        be._call("i2p_warp_split_bwd", int(B), int(N), P(p), P(q), P(t), P(v) if ctx.has_valid else None, P(g_uv), P(g_z), P(g_xyz), P(dq), P(dt),
                 stream=be._stream())
        return None, dq, dt, None


def warp_split(p, q, t_quat, valid=None):
    """p [B,N,3] (no gradient), q [B,4], t_quat [B,4] = [0,t], valid [B,N,1] 0/1 or None ->
    (uv [B,N,3], z [B,N,1], xyz [B,N,3]) of the warped, masked cloud; None if the fused op does not apply"""
    be = ops.get_backend()
    if (be.name != "hip" or not p.is_cuda or p.dtype != torch.float32 or p.requires_grad or q.dtype != torch.float32
            or (valid is not None and valid.requires_grad)):
        return None
    return _WarpSplit.apply(p, q.reshape(-1, 4), t_quat.reshape(-1, 4), valid)


class _PoseCompose(torch.autograd.Function):
    """[q3 (x) q_prev, (q3 (x) [0, t_prev] (x) q3^-1)[1:4] + t3] -> [B,7] in one launch each way (csrc/projection_ops.hip
    i2p_pose_compose_fwd/bwd; modellearn_proj_center.py:388-404)."""

    @staticmethod
    def forward(ctx, q3, t3, q_prev, t_prev):
        q3, t3, q_prev, t_prev = [t.detach().contiguous() for t in (q3, t3, q_prev, t_prev)]
        be = ops.get_backend()
        B = q3.shape[0]
        out = torch.empty(B, 7, dtype=torch.float32, device=q3.device)
        P = be._p
        be._call("i2p_pose_compose_fwd", int(B), P(q3, torch.float32, "q3"), P(t3, torch.float32, "t3"), P(q_prev, torch.float32, "q_prev"),
                 P(t_prev, torch.float32, "t_prev"), P(out, torch.float32, "out"), stream=be._stream())
        ctx.save_for_backward(q3, q_prev, t_prev)
        return out

    @staticmethod
    def backward(ctx, g):
        q3, q_prev, t_prev = ctx.saved_tensors
        be = ops.get_backend()
        B = q3.shape[0]
        g = g.contiguous()
        dq3, dqp = torch.empty_like(q3), torch.empty_like(q_prev)
        dt3, dtp = torch.empty(B, 3, dtype=torch.float32, device=g.device), torch.empty(B, 4, dtype=torch.float32, device=g.device)
        P = be._p
        be._call("i2p_pose_compose_bwd", int(B), P(q3, torch.float32, "q3"), P(q_prev, torch.float32, "q_prev"), P(t_prev, torch.float32, "t_prev"),
                 P(g, torch.float32, "g"), P(dq3, torch.float32, "dq3"), P(dt3, torch.float32, "dt3"), P(dqp, torch.float32, "dqp"),
                 P(dtp, torch.float32, "dtp"), stream=be._stream())
        return dq3, dt3, dqp, dtp


def compose_pose(q3, t3, q_prev, t_prev_quat):
    """q3 [B,4], t3 [B,3], q_prev [B,4], t_prev_quat [B,4] = [0, t_prev] -> composed pose [B,7] (q = q3 * q_prev, t = R3 t_prev + t3)"""
    B = q3.shape[0]
    be = ops.get_backend()
    if (be.name == "hip" and q3.is_cuda and all(t.dtype == torch.float32 for t in (q3, t3, q_prev, t_prev_quat))
            and os.environ.get("I2P_NO_POSE_COMPOSE") != "1"):
        return _PoseCompose.apply(q3.reshape(B, 4), t3.reshape(B, 3), q_prev.reshape(B, 4), t_prev_quat.reshape(B, 4))
    out_q = mul_q(q3.view(B, 1, 4), q_prev.view(B, 1, 4)).squeeze(1)
    t3_quat = torch.cat([ops.zero_scalar(q3.device, t3.dtype).expand(B, 1), t3], 1).view(B, 1, 4)
    out_t = mul_q(mul_q(q3, t_prev_quat.view(B, 1, 4)), inv_q(q3)) + t3_quat
    return torch.cat([out_q, out_t.squeeze(1)[:, 1:]], 1)
