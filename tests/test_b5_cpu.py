"""B5 compositions (QueryAndGroup / GroupAll / KNN / gather + interpolate backward) on the CPU oracle backend:
host logic against plain-torch formulations of the same definitions (pointnet2_utils.py:262-321, 70-184)."""
import torch

from helpers import cloud


def test_b5_compositions_on_oracle(oracle_backend):
    from i2pnet_amd import ops, pointnet2_utils as pu
    prev = ops.set_backend(oracle_backend)
    try:
        B, N, M, C = 2, 256, 32, 4
        xyz = cloud(B, N, seed=4)
        feats = torch.randn(B, C, N, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
        idx = pu.furthest_point_sample(xyz, M)
        new_xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        qg = pu.QueryAndGroup(8.0, 8, use_xyz=True)(xyz, new_xyz, feats)
        bq = pu.ball_query(8.0, 8, xyz, new_xyz).long()                               # [B,M,8]
        want_xyz = torch.gather(xyz.unsqueeze(1).expand(-1, M, -1, -1), 2, bq.unsqueeze(-1).expand(-1, -1, -1, 3))
        want_xyz = (want_xyz - new_xyz.unsqueeze(2)).permute(0, 3, 1, 2)
        want_f = torch.gather(feats.unsqueeze(2).expand(-1, -1, M, -1), 3, bq.unsqueeze(1).expand(-1, C, -1, -1))
        assert torch.equal(qg[:, :3], want_xyz) and torch.equal(qg[:, 3:], want_f)
        ga = pu.GroupAll()(xyz, None, feats)
        assert torch.equal(ga, torch.cat([xyz.transpose(1, 2).unsqueeze(2), feats.unsqueeze(2)], 1))
        gat = pu.gather_operation(feats, idx)
        dist, i3 = pu.three_nn(xyz, new_xyz)
        w = 1.0 / (dist + 1e-8); w = (w / w.sum(2, keepdim=True)).contiguous()
        interp = pu.three_interpolate(gat, i3, w)
        want_i = (torch.gather(gat.unsqueeze(2).expand(-1, -1, N, -1), 3, i3.long().unsqueeze(1).expand(-1, C, -1, -1)) * w.unsqueeze(1)).sum(-1)
        assert torch.allclose(interp, want_i, rtol=1e-5, atol=1e-6)
        (interp.sum() + (gat ** 2).sum() + (qg ** 2).sum()).backward()
        g_mine = feats.grad.clone(); feats.grad = None
        gat_t = torch.gather(feats, 2, idx.long().unsqueeze(1).expand(-1, C, -1))
        want_i2 = (torch.gather(gat_t.unsqueeze(2).expand(-1, -1, N, -1), 3, i3.long().unsqueeze(1).expand(-1, C, -1, -1)) * w.unsqueeze(1)).sum(-1)
        qf = torch.gather(feats.unsqueeze(2).expand(-1, -1, M, -1), 3, bq.unsqueeze(1).expand(-1, C, -1, -1))
        (want_i2.sum() + (gat_t ** 2).sum() + (qf ** 2).sum()).backward()
        assert torch.allclose(g_mine, feats.grad, rtol=1e-4, atol=1e-5)
        kd, ki = pu.knn(4, new_xyz, xyz)
        full = ((new_xyz.unsqueeze(2) - xyz.unsqueeze(1)) ** 2).sum(-1).sqrt()
        assert torch.allclose(kd.sort(dim=2)[0], torch.topk(full, 4, dim=2, largest=False)[0].sort(dim=2)[0], rtol=1e-5, atol=1e-5)
    finally:
        ops.set_backend(prev)
