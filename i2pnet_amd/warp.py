"""Quaternion algebra used by the coarse-to-fine registration (reference:
src/modules/warp_utils.py:10-55, :78-94).  Device-agnostic (the reference hard-codes .cuda())."""
import torch


def inv_q(q):
    """q [B,4] or [B,1,4] (w,x,y,z) -> conj(q) / (|q|^2 + 1e-10), [B,4]"""
    B = q.shape[0]
    q = q.reshape(B, 4)
    n2 = torch.sum(q * q, dim=-1, keepdim=True) + 1e-10
    conj = torch.cat([q[:, :1], -q[:, 1:]], dim=-1)     # no host constant: hipGraph-capturable
    return conj / n2


def mul_q(a, b):
    """Hamilton product, a/b [B,N|1,4] or [B,4] -> [B,N,4]"""
    if a.ndim == 2:
        a = a.unsqueeze(1)
    if b.ndim == 2:
        b = b.unsqueeze(1)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return torch.stack([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def warp_quat_xyz(lidar_xyz, Hi_quat, H_trans):
    """p' = q [0,p] q^-1 + t for p [B,N,3], q [B,4], t [B,4] = [0,tx,ty,tz] -> [B,N,3]"""
    B, N, _ = lidar_xyz.shape
    homo = torch.cat([lidar_xyz.new_zeros(B, N, 1), lidar_xyz], -1)
    homo = mul_q(mul_q(Hi_quat, homo), inv_q(Hi_quat)) + H_trans.reshape(B, 1, 4)
    return homo[:, :, 1:4]
