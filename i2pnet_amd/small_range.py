"""Small-range registration network on the HIP operator layer.

Counterpart of the reference's `src/modellearn.py::RegNet_v2` (:24-392) and its building blocks
`src/modules/MainModules.py::CostVolume` (:51-243), `src/modules/pointnet2_module.py::SetUpconvModule`
(:7-81), `src/modules/point_utils.py::grouping` (:68-112), `src/modules/warp_utils.py::warp_quat` /
`projection_initial` (:56-76, :146-155): a point-based pyramid (furthest-point sampling + kNN grouping instead
of range images), the same 2D-3D cost volumes, up-convolutions, mask predictors and pose heads.  Same
constructor arguments, sub-module and parameter names (=> the reference's `state_dict` loads) and outputs.

What runs underneath: FPS, kNN (no `[B,S,N]` distance matrix), channel-last row gathers with scatter-add
backward and the quaternion kernels from libi2p_ops.so; the image encoder blocks as in the projection model.
The point-branch convolutions of THIS model carry BatchNorm2d with running statistics (the projection model
uses batch statistics always), so they run as a GEMM + BatchNorm (train: batch statistics, running buffers
updated; eval: running statistics) — the fused batch-statistics layer kernels do not apply to them.
"""
import torch
import torch.nn as nn

from . import projectpn as P
from . import warp as warp_utils
from .model import scaled_intrinsic_inverse, set_id_grid
from .modules import (Conv2d, CostVolume, FlowPredictor, PoseHead, _MaxResponse, _unit_variance, createCNNs,
                      run_stack)
from .pointnet_util import PointNetSetAbstraction, index_points, knn_point


class SmallRangeConfig:
    """values of the reference's `src/config_lidarcenter.py:8-105` that the model reads"""
    rgb_encoder_channels = [(3, [16, 16, 16, 16, 32], [2, 1, 1, 1, 2]), (32, [32, 32, 32, 32, 64], [2, 1, 1, 1, 2]),
                            (64, [64, 64, 64, 64, 128], [1, 1, 1, 1, 2])]
    lidar_downsample_rate = [4, 2, 4, 4]
    lidar_in_points = 8192
    lidar_feature_size = 7
    featmode = "dim10feat"
    raw_feat_point = True
    lidar_group_samples = [32, 16, 16, 16, 16]
    lidar_encoder_mlps = [[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128, 256], [128, 64, 64]]
    cost_volume_mlps = [[128, 64, 64], [128, 64]]
    cost_volume_nsamples = [4, [-1, 32]]
    backward_validation = [True, False]
    setupconv_mlps = [[[128, 64], [64]], [[128, 64], [64]]]
    setupconv_nsamples = [8, 8]
    flow_predictor_mlps = [[128, 64], [128, 64], [128, 64]]
    head_hidden_dim = 256
    rotation_quat_head_dim = 4
    transition_vec_head_dim = 3
    head_dropout_rate = 0.5
    head_pos_embedding = False
    split_dp = False
    max_head = False
    mask_sigmoid = False
    sq_init = -2.5
    sx_init = 0.0
    l1_trans_loss = True


def grouping(feature, K, src_xyz, q_xyz, use_xyz=False, raw_feat_point=False, raw_xyz1=None, raw_xyz2=None):
    """kNN grouping (src/modules/point_utils.py:68-112): -> grouped_xyz [B,S,K,3], xyz_diff, new_points [B,S,K,c(+3)],
    indices [B,S,K], grouped_raw_xyz or None.  Neighbours ordered by (distance, index)."""
    idx = knn_point(K, src_xyz.contiguous(), q_xyz.contiguous())
    grouped_xyz = index_points(src_xyz, idx)
    grouped_raw = index_points(raw_xyz1, idx) if raw_feat_point else None
    xyz_diff = (grouped_raw - raw_xyz2.unsqueeze(2)) if raw_feat_point else (grouped_xyz - q_xyz.unsqueeze(2))
    grouped_feature = index_points(feature, idx)
    new_points = torch.cat([xyz_diff, grouped_feature], dim=-1) if use_xyz else grouped_feature
    return grouped_xyz, xyz_diff, new_points, idx, grouped_raw


class SetUpconvModule(nn.Module):
    """kNN up-convolution (src/modules/pointnet2_module.py:7-81): coarse features to the fine points."""

    def __init__(self, nsample, in_channels, mlp, mlp2, is_training=None, bn_decay=None, bn=True, pooling="max",
                 radius=None, knn=True):
        super().__init__()
        self.nsample, self.pooling = nsample, pooling
        self.mlp_conv = nn.ModuleList()
        self.mlp2_conv = nn.ModuleList()
        last = in_channels[-1] + 3
        for c in (mlp or []):
            self.mlp_conv.append(Conv2d(last, c, [1, 1], stride=[1, 1], bn=True, use_bn_input=False))
            last = c
        last = (mlp[-1] if mlp else last) + in_channels[0]
        for c in (mlp2 or []):
            self.mlp2_conv.append(Conv2d(last, c, [1, 1], stride=[1, 1], bn=True, use_bn_input=False))
            last = c

    def forward(self, xyz1, xyz2, feat1, feat2, raw_feat_point=False, raw_xyz1=None, raw_xyz2=None):
        """xyz1 [B,n1,3] fine, xyz2 [B,n2,3] coarse, feat1 [B,n1,c1], feat2 [B,n2,c2] -> [B,n1,mlp2[-1]]"""
        xyz2_grouped, _, feat2_grouped, _, raw_grouped = grouping(feat2, self.nsample, xyz2, xyz1, raw_feat_point=raw_feat_point,
                                                                  raw_xyz1=raw_xyz2, raw_xyz2=raw_xyz1)
        xyz_diff = (raw_grouped - raw_xyz1.unsqueeze(2)) if raw_feat_point else (xyz2_grouped - xyz1.unsqueeze(2))
        net = run_stack(torch.cat([feat2_grouped, xyz_diff], dim=3), self.mlp_conv)
        feat1_new = torch.max(net, dim=2)[0] if self.pooling == "max" else torch.mean(net, dim=2)
        if feat1 is not None:
            feat1_new = torch.cat([feat1_new, feat1], dim=2)
        return run_stack(feat1_new.unsqueeze(2), self.mlp2_conv).squeeze(2)


class CostVolumeKnn(CostVolume):
    """Point-based 2D-3D cost volume (src/modules/MainModules.py:51-243): the projection model's cost volume with
    (i) BatchNorm2d on running statistics, (ii) no validity masks, (iii) the pc-stage neighbourhood from a kNN search
    over the warped points.  Same sub-module names as the reference."""

    def __init__(self, radius, nsample, nsample_q, rgb_in_channels, lidar_in_channels, mlp1, mlp2, is_training=None,
                 bn_decay=None, bn=True, pooling="max", knn=True, corr_func=None, backward_validation=False,
                 max_cost=False, backward_fc=False):
        assert not max_cost and not backward_fc, "configuration not used by the reference's small-range model"
        super().__init__(H=1, W=1, kernel_size=None, distance=None, nsample=nsample, nsample_q=nsample_q,
                         rgb_in_channels=rgb_in_channels, lidar_in_channels=lidar_in_channels, mlp1=mlp1, mlp2=mlp2,
                         backward_validation=backward_validation, use_bn_input=False)
        self.mask_invalid = False

    def forward(self, warped_xyz, warped_points, f2_xyz, f2_points, lidar_z):
        """warped_xyz [B,N,3] (u,v,1); warped_points [B,N,C]; f2_xyz [B,M,3]; f2_points [B,M,C]; lidar_z [B,N,1]
        -> [B,N,mlp2[-1]]"""
        B, N, _ = warped_xyz.shape
        uv = warped_xyz
        xyz = warped_xyz.mul(lidar_z)                                           # restore depth, :139
        pts_n = _unit_variance(warped_points)                                   # :149-153
        pix_n = _unit_variance(f2_points)
        # pi-stage: the projection model's implementation (factored first layer / fused tail when the layer kernels
        # apply, i.e. in training mode; plain layers otherwise) with the point mask switched off
        if self.nsample_q > 0:
            h3, enc, pi_feat = self._pi_knn(uv, xyz, pts_n, f2_xyz, pix_n)
        else:
            h3, enc, pi_feat = self._pi_all_pixels(xyz, pts_n, f2_xyz, pix_n)
        if pi_feat is None:
            logits = run_stack(torch.cat([enc, h3], dim=3), self.mlp2_convs)    # :190-196
            pi_feat = torch.sum(torch.softmax(logits, dim=2) * h3, dim=2)       # [B,N,c]

        # pc-stage: kNN over the warped points (:204-240)
        K = self.nsample
        idx = knn_point(K, xyz.contiguous(), xyz.contiguous())
        nb_xyz, nb_feat = index_points(xyz, idx), index_points(pi_feat, idx)
        own_xyz = xyz.unsqueeze(2).expand(-1, -1, K, -1)
        diff = nb_xyz - own_xyz
        euc = torch.sqrt(torch.sum(diff * diff, dim=3, keepdim=True) + 1e-20)
        enc_pc = run_stack(torch.cat([own_xyz, nb_xyz, diff, euc], dim=3), [self.pc_encoding])
        w = run_stack(torch.cat([enc_pc, warped_points.unsqueeze(2).expand(-1, -1, K, -1), nb_feat], dim=-1), self.mlp2_convs_2)
        return torch.sum(torch.softmax(w, dim=2) * nb_feat, dim=2)


def _fp(in_channels, mlp):
    fp = FlowPredictor(in_channels=in_channels, mlp=mlp, is_training=None, bn_decay=None, bn=True, use_bn_input=False)
    return fp


class RegNet_v2(nn.Module):
    """Small-range image-to-point-cloud registration network (src/modellearn.py:24-392).
    forward(rgb_img [B,3,h,w], lidar_img [B,N,3], H_initial, intrinsic [B,3,3], resize_img, ..., lidar_feature,
    cfg, lidar_img_raw [B,N,3]) -> (out3 [B,7], out4 [B,7], None, None, sx, sq)"""

    def __init__(self, bn_decay=None, eval_info=False, cfg=SmallRangeConfig):
        super().__init__()
        self.eval_info, self.cfg = eval_info, cfg
        npts = [cfg.lidar_in_points // s for s in __import__("numpy").cumprod(cfg.lidar_downsample_rate)]
        enc = cfg.lidar_encoder_mlps
        gs = cfg.lidar_group_samples
        sa = lambda n, r, k, cin, mlp: PointNetSetAbstraction(npoint=n, radius=r, nsample=k, in_channel=cin, mlp=mlp, group_all=False)
        self.LiDAR_lv1 = sa(npts[0], 0.5, gs[0], cfg.lidar_feature_size + 3, enc[0])
        self.LiDAR_lv2 = sa(npts[1], 0.5, gs[1], enc[0][-1] + 3, enc[1])
        self.LiDAR_lv3 = sa(npts[2], 1.0, gs[2], enc[1][-1] + 3, enc[2])
        self.LiDAR_lv4 = sa(npts[3], 2.0, gs[3], enc[2][-1] + 3, enc[3])
        self.layer_idx = sa(npts[3], 2.0, gs[4], cfg.cost_volume_mlps[-1][-1] + 3, enc[4])
        self.RGB_net1 = createCNNs(*cfg.rgb_encoder_channels[0])
        self.RGB_net2 = createCNNs(*cfg.rgb_encoder_channels[1])
        self.RGB_net3 = createCNNs(*cfg.rgb_encoder_channels[2])
        for net in (self.RGB_net1, self.RGB_net2, self.RGB_net3):
            net.to(memory_format=torch.channels_last)
        cv = lambda i: CostVolumeKnn(radius=10.0, nsample=cfg.cost_volume_nsamples[0], nsample_q=cfg.cost_volume_nsamples[1][i],
                                     rgb_in_channels=cfg.rgb_encoder_channels[-1][1][-1], lidar_in_channels=enc[-3][-1],
                                     mlp1=cfg.cost_volume_mlps[0], mlp2=cfg.cost_volume_mlps[1],
                                     backward_validation=cfg.backward_validation[i])
        self.cost_volume1, self.cost_volume2 = cv(0), cv(1)
        self.flow_predictor0 = _fp(enc[-2][-1] + enc[-1][-1], cfg.flow_predictor_mlps[0])
        up = lambda i, c2: SetUpconvModule(nsample=cfg.setupconv_nsamples[i], radius=2.4, in_channels=[enc[-3][-1], c2],
                                           mlp=cfg.setupconv_mlps[i][0], mlp2=cfg.setupconv_mlps[i][1])
        self.set_upconv0_w_upsample = up(0, cfg.flow_predictor_mlps[0][-1])
        self.set_upconv0_upsample = up(1, enc[-1][-1])
        self.flow_predictor0_predict = _fp(enc[-3][-1] + cfg.setupconv_mlps[1][1][-1] + cfg.cost_volume_mlps[-1][-1],
                                           cfg.flow_predictor_mlps[1])
        self.flow_predictor0_w = _fp(enc[-3][-1] + cfg.setupconv_mlps[0][-1][-1] + cfg.flow_predictor_mlps[1][-1],
                                     cfg.flow_predictor_mlps[2])
        head = lambda c_pred, c_feat: PoseHead(in_channels=[c_pred, c_feat], mlp1=[], mlp2=[], hidden=cfg.head_hidden_dim,
                                               q_dim=cfg.rotation_quat_head_dim, t_dim=cfg.transition_vec_head_dim,
                                               dropout_rate=cfg.head_dropout_rate, split_dp=cfg.split_dp,
                                               pos_embed=cfg.head_pos_embedding, sigmoid=cfg.mask_sigmoid, maxhead=cfg.max_head)
        self.l4_head = head(enc[-1][-1], enc[-2][-1])
        self.l3_head = head(cfg.flow_predictor_mlps[1][-1], enc[-3][-1])
        self.l3_head.training_needs_weights = eval_info
        self.sq = nn.Parameter(torch.tensor([cfg.sq_init]), requires_grad=True)
        self.sx = nn.Parameter(torch.tensor([cfg.sx_init]), requires_grad=True)

    def forward(self, rgb_img, lidar_img, H_initial, intrinsic, resize_img, gt_project=None, calib=None, lidar_feature=None,
                cfg=None, lidar_img_raw=None):
        cfg = cfg or self.cfg
        dev = rgb_img.device
        intrinsic = intrinsic.float()
        B, N = lidar_img.shape[0], lidar_img.shape[1]
        rfp = cfg.raw_feat_point
        rgb_img = rgb_img.contiguous(memory_format=torch.channels_last)
        RF3 = self.RGB_net3(self.RGB_net2(self.RGB_net1(rgb_img)))
        pix_index = set_id_grid(RF3.permute(0, 2, 3, 1))                        # [B,M,3]
        xyz0 = lidar_img.permute(0, 2, 1).float()                               # [B,3,N]
        feat0 = (torch.zeros(B, N, 3, device=dev) if lidar_feature is None else lidar_feature).permute(0, 2, 1).float()
        P1, LF1, _, _, P1_raw = self.LiDAR_lv1(xyz0, feat0, feat_mode=cfg.featmode, raw_feat_point=rfp, raw_xyz=lidar_img_raw)
        P2, LF2, _, _, P2_raw = self.LiDAR_lv2(P1, LF1, raw_feat_point=rfp, raw_xyz=P1_raw)
        P3, LF3, _, _, P3_raw = self.LiDAR_lv3(P2, LF2, raw_feat_point=rfp, raw_xyz=P2_raw)
        P4, LF4, _, fps_idx_4, P4_raw = self.LiDAR_lv4(P3, LF3, raw_feat_point=rfp, raw_xyz=P3_raw)

        K3_inv = scaled_intrinsic_inverse(intrinsic, RF3, rgb_img)
        pix_rays = torch.bmm(K3_inv, pix_index.permute(0, 2, 1)).permute(0, 2, 1)
        P3_pts = P3.permute(0, 2, 1)                                            # [B,n3,3]
        LF3_pts = LF3.permute(0, 2, 1)
        lidar_z = P3_pts[:, :, 2:]                                              # projection_initial, warp_utils.py:146-155
        lidar_uv = P3_pts / lidar_z
        RF3_pts = RF3.reshape(B, RF3.shape[1], -1).permute(0, 2, 1)             # [B,M,C]

        concat_4 = self.cost_volume1(lidar_uv, LF3_pts, pix_rays, RF3_pts, lidar_z)
        P4, l4_embed, _, _, _ = self.layer_idx(P3, concat_4.permute(0, 2, 1), sample_idx=fps_idx_4, raw_feat_point=rfp, raw_xyz=P3_raw)
        l4_embed = l4_embed.permute(0, 2, 1)
        LF4_pts, P4_pts = LF4.permute(0, 2, 1), P4.permute(0, 2, 1)
        l4_mask = self.flow_predictor0(LF4_pts, None, l4_embed)
        q4, t4, _ = self.l4_head(l4_embed, l4_mask, P4_pts, LF4_pts, None)
        result_4 = torch.cat([q4, t4], dim=1)

        t4_quat = torch.cat([torch.zeros((B, 1), device=dev), t4], -1)
        warped = warp_utils.warp_quat_xyz(P3_pts, q4, t4_quat)                  # warp_quat, warp_utils.py:56-76
        lidar_z = warped[:, :, 2:]
        lidar_uv = warped / (lidar_z + 1e-10)
        concat_3 = self.cost_volume2(lidar_uv, LF3_pts, pix_rays, RF3_pts, lidar_z)
        l3_mask_up = self.set_upconv0_w_upsample(P3_pts, P4_pts, LF3_pts, l4_mask, raw_feat_point=rfp, raw_xyz1=P3_raw, raw_xyz2=P4_raw)
        l3_embed_up = self.set_upconv0_upsample(P3_pts, P4_pts, LF3_pts, l4_embed, raw_feat_point=rfp, raw_xyz1=P3_raw, raw_xyz2=P4_raw)
        l3_embed = self.flow_predictor0_predict(LF3_pts, l3_embed_up, concat_3)
        l3_mask = self.flow_predictor0_w(LF3_pts, l3_mask_up, l3_embed)
        q3, t3, W_l3 = self.l3_head(l3_embed, l3_mask, P3_pts, LF3_pts, None)

        out_q = warp_utils.mul_q(q3.view(B, 1, 4), q4.view(B, 1, 4)).squeeze(1)
        t3_quat = torch.cat([torch.zeros((B, 1), device=dev), t3], 1).view(B, 1, 4)
        out_t = warp_utils.mul_q(warp_utils.mul_q(q3, t4_quat.view(B, 1, 4)), warp_utils.inv_q(q3)) + t3_quat
        out_3 = torch.cat([out_q, out_t.squeeze(1)[:, 1:]], 1)
        if self.eval_info:
            return out_3.float(), result_4.float(), self.sx, self.sq, W_l3, P3_pts, None, None, P4_pts
        return out_3.float(), result_4.float(), None, None, self.sx, self.sq
