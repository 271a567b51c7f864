"""Learned-uncertainty pose loss (reference: compute_loss.py:102-133, `Get_loss`)."""
import torch


def Get_loss(out3, out4, qq_gt, t_gt, w_x, w_q, cfg):
    """-> (loss, rotation part, translation part); weights 0.8 on the fine pose `out3`,
    1.6 on the coarse pose `out4` (compute_loss.py:127-130).  Both poses are evaluated in one stacked
    pass ([2,B,7]) — the same arithmetic per element as the reference's two calls, half the launches."""
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
