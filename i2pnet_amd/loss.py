"""Learned-uncertainty pose loss (reference: compute_loss.py:102-133, `Get_loss`)."""
import torch
import torch.nn.functional as F


def _pose_terms(out, q_gt, t_gt, l1_trans):
    q, t = out[:, :4], out[:, 4:]
    loss_q = torch.mean(torch.sqrt(torch.sum((q_gt - q) * (q_gt - q), dim=-1, keepdim=True) + 1e-10))
    if l1_trans:
        loss_x = F.l1_loss(t, t_gt)
    else:
        loss_x = torch.mean(torch.sqrt(torch.sum((t - t_gt) * (t - t_gt), dim=-1, keepdim=True) + 1e-10))
    return loss_q, loss_x


def Get_loss(out3, out4, qq_gt, t_gt, w_x, w_q, cfg):
    """-> (loss, rotation part, translation part); weights 0.8 on the fine pose `out3`,
    1.6 on the coarse pose `out4` (compute_loss.py:127-130)."""
    fq, fx = _pose_terms(out3, qq_gt, t_gt, cfg.l1_trans_loss)
    cq, cx = _pose_terms(out4, qq_gt, t_gt, cfg.l1_trans_loss)
    fine = fx * torch.exp(-w_x) + w_x + fq * torch.exp(-w_q) + w_q
    coarse = cx * torch.exp(-w_x) + w_x + cq * torch.exp(-w_q) + w_q
    return 1.6 * coarse + 0.8 * fine, 1.6 * cq + 0.8 * fq, 1.6 * cx + 0.8 * fx
