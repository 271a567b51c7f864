"""Learned-uncertainty pose loss (reference: compute_loss.py:102-133, `Get_loss`)."""
import torch

from . import ops

USE_FUSED_LOSS = True       # one HIP launch for the loss and its gradient (False: the torch formulation below)


class _PoseLoss(torch.autograd.Function):
    """loss and dloss/d(out3, out4, w_x, w_q) from one launch of `pose_loss_kernel` (csrc/projection_ops.hip); the
    rotation / translation parts are returned for logging only (not differentiable)."""

    @staticmethod
    def forward(ctx, out3, out4, q_gt, t_gt, w_x, w_q, l1):
        f = lambda t: t.detach().float().contiguous()
        be = ops.get_backend()
        loss3, d3, d4, d_w = be.pose_loss(f(out3), f(out4), f(q_gt), f(t_gt), f(w_x), f(w_q), l1)
        flat = getattr(be, "last_pose_loss_flat", None)
        ctx.n7 = None
        if flat is not None and flat[0].data_ptr() == d3.data_ptr():        # device library: d3 | d4 | d_w are slices of one buffer
            ctx.n7, ctx.B = flat[1], out3.shape[0]
            ctx.save_for_backward(flat[0])
        else:
            ctx.save_for_backward(d3, d4, d_w)
        real, dual = loss3[1:2], loss3[2:3]
        ctx.mark_non_differentiable(real, dual)
        return loss3[0:1], real, dual

    @staticmethod
    def backward(ctx, g, _g_real, _g_dual):
        if ctx.n7 is not None:          # one multiply for all four gradients
            (flat,) = ctx.saved_tensors
            s, n7, B = flat * g, ctx.n7, ctx.B
            return (s[:B * 7].view(B, 7), s[n7:n7 + B * 7].view(B, 7), None, None, s[2 * n7:2 * n7 + 1], s[2 * n7 + 1:2 * n7 + 2], None)
        d3, d4, d_w = ctx.saved_tensors
        return d3 * g, d4 * g, None, None, d_w[0:1] * g, d_w[1:2] * g, None


def Get_loss(out3, out4, qq_gt, t_gt, w_x, w_q, cfg):
    """-> (loss [1], rotation part, translation part); weights 0.8 on the fine pose `out3`, 1.6 on the coarse pose
    `out4` (compute_loss.py:127-130)."""
    B = out3.shape[0]
    if USE_FUSED_LOSS and B <= 1024 and out3.dtype == torch.float32:
        return _PoseLoss.apply(out3, out4, qq_gt, t_gt, w_x, w_q, bool(cfg.l1_trans_loss))
    out = torch.stack([out4, out3])                                    # [2,B,7]: coarse, fine
    dq = qq_gt.unsqueeze(0) - out[:, :, :4]
    loss_q = torch.sqrt(torch.sum(dq * dq, dim=-1) + 1e-10).mean(dim=1)             # [2]  compute_loss.py:112
    dt = out[:, :, 4:] - t_gt.unsqueeze(0)
    if cfg.l1_trans_loss:
        loss_x = dt.abs().mean(dim=(1, 2))                                          # F.l1_loss, :115
    else:
        loss_x = torch.sqrt(torch.sum(dt * dt, dim=-1) + 1e-10).mean(dim=1)
    per_level = loss_x * torch.exp(-w_x) + w_x + loss_q * torch.exp(-w_q) + w_q     # [2]  :121
    loss = 1.6 * per_level[0:1] + 0.8 * per_level[1:2]                               # shape [1] like the reference
    return loss, 1.6 * loss_q[0] + 0.8 * loss_q[1], 1.6 * loss_x[0] + 0.8 * loss_x[1]
