"""Large-range image-to-point-cloud registration network on the HIP operator layer.

Counterpart of the reference's `src/modellearn_proj_center.py::RegNet_v2` (:24-424): same
constructor / forward signature, same sub-module names (=> same 299-tensor `state_dict`,
SURVEY.md §8c), same outputs `(out3 [B,7], out4 [B,7], None, None, sx, sq)`.

Stages: image CNN (PyTorch-ROCm) -> spherical projection -> four set-abstraction levels on
range images -> coarse cost volume over all image pixels -> coarse pose -> warp -> up-conv ->
fine cost volume over 32-NN pixels -> fine pose -> pose composition.

Host-side differences from the reference: no CPU round trip for the 3x3 intrinsic inverse
(`torch.inverse(intrinsic_3.cpu())`, :282, forces a sync every forward) and no `.item()`.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import projectpn as P
from . import warp as warp_utils
from .config import I2PNetConfig as cfg_default
from .modules import (CostVolume, FlowPredictor, PoseHead, ProjectPointNet, ProjSetUpconvModule, _unit_variance,
                      createCNNs, mask_fill)


_const_cache = {}


def _const(key, make):
    """small constant tensors (pixel grid, intrinsic scale) built once per shape/device, outside any
    hipGraph capture (the first, eager, step)"""
    t = _const_cache.get(key)
    if t is None:
        t = _const_cache[key] = make()
    return t


def set_id_grid(rf):
    """rf [B,h,w,*] -> pixel coordinates (u, v, 1) [B, h*w, 3] (modellearn_proj_center.py:440-454)."""
    B, h, w = rf.shape[0], rf.shape[1], rf.shape[2]

    def make():
        v, u = torch.meshgrid(torch.arange(h, device=rf.device, dtype=rf.dtype),
                              torch.arange(w, device=rf.device, dtype=rf.dtype), indexing="ij")
        return torch.stack([u, v, torch.ones_like(u)], dim=-1).reshape(1, h * w, 3)
    return _const(("grid", h, w, str(rf.device), rf.dtype), make).expand(B, -1, -1)


def change_intrinsic(intrinsic, RF, rgb_img):
    """rescale fx, cx by w'/w and fy, cy by h'/h (modellearn_proj_center.py:457-463): one multiply
    by a constant [3,3] scale (rows 0 and 1; exact — the same products as the four in-place scalings)."""
    sx = RF.shape[3] / rgb_img.shape[3]
    sy = RF.shape[2] / rgb_img.shape[2]
    scale = _const(("kscale", sx, sy, str(intrinsic.device)),
                   lambda: torch.tensor([[sx, 1.0, sx], [1.0, sy, sy], [1.0, 1.0, 1.0]], device=intrinsic.device))
    return intrinsic * scale


def scaled_intrinsic_inverse(intrinsic, RF, rgb_img):
    """inverse_3x3(change_intrinsic(...)): on the HIP backend one launch (i2p_intrinsic_inverse, the same arithmetic) instead of eight;
    the intrinsics are inputs (no gradient)"""
    from . import ops
    be = ops.get_backend()
    if (be.name == "hip" and intrinsic.is_cuda and intrinsic.dtype == torch.float32 and not intrinsic.requires_grad and intrinsic.dim() == 3):
        return be.intrinsic_inverse(intrinsic.contiguous(), RF.shape[3] / rgb_img.shape[3], RF.shape[2] / rgb_img.shape[2])
    return inverse_3x3(change_intrinsic(intrinsic, RF, rgb_img))


def inverse_3x3(m):
    """adjugate inverse of [B,3,3] on the device (replaces torch.inverse on the CPU, :282):
    rows of the adjugate-transpose are cross products of the rows of m, det = r0 . (r1 x r2)."""
    r0, r1, r2 = m[:, 0], m[:, 1], m[:, 2]
    c0 = torch.linalg.cross(r1, r2, dim=-1)
    c1 = torch.linalg.cross(r2, r0, dim=-1)
    c2 = torch.linalg.cross(r0, r1, dim=-1)
    det = (r0 * c0).sum(-1)
    return torch.stack([c0, c1, c2], dim=-1) / det.view(-1, 1, 1)


# (Two-branch schedules — image encoder || LiDAR pyramid on two HIP streams, the first image block's statistics beside the pyramid —
#  were measured in rounds 2-4 and always lost to the single stream: 485-488 vs 491-495 samples/s, 11.79 vs 11.40 ms.  The pyramid's
#  small launches do not hide under the encoder's kernels, they slow them down.  The code paths were removed in round 5.)
class _BranchJoin(torch.autograd.Function):
    """identity on the image encoder's outputs, created on the main stream right behind the join of the two encoder streams: its
    backward marks the point of the backward pass from which the two encoders' backward passes run side by side"""

    @staticmethod
    def forward(ctx, a, b):
        return a.view_as(a), b.view_as(b)

    @staticmethod
    def backward(ctx, ga, gb):
        return ga, gb


class RegNet_v2(nn.Module):
    def __init__(self, bn_decay=None, eval_info=False, cfg=cfg_default):
        super().__init__()
        self.eval_info = eval_info
        self.cfg = cfg
        self.lidar_Hs = [int(np.ceil(cfg.init_H / s)) for s in np.cumprod(cfg.stride_Hs)]
        self.lidar_Ws = [int(np.ceil(cfg.init_W / s)) for s in np.cumprod(cfg.stride_Ws)]
        Hs, Ws = [cfg.init_H] + self.lidar_Hs, [cfg.init_W] + self.lidar_Ws
        enc = cfg.lidar_encoder_mlps
        bn = dict(use_trans=cfg.use_trans, use_bn_p=cfg.use_bn_p, use_bn_input=cfg.use_bn_input)
        add_num = 4 if cfg.using_intens else 3

        def sa(level, in_channel, mlp, nsample):
            return ProjectPointNet(H=Hs[level], W=Ws[level], out_h=Hs[level + 1], out_w=Ws[level + 1],
                                   stride_H=cfg.stride_Hs[level], stride_W=cfg.stride_Ws[level],
                                   kernel_size=cfg.kernel_sizes[level], nsample=nsample,
                                   distance=cfg.down_conv_dis[level], in_channel=in_channel, mlp=mlp, **bn)

        self.LiDAR_lv1 = sa(0, cfg.lidar_feature_size + add_num, enc[0], cfg.lidar_group_samples[0])
        self.LiDAR_lv2 = sa(1, enc[0][-1] + 3, enc[1], cfg.lidar_group_samples[1])
        self.LiDAR_lv3 = sa(2, enc[1][-1] + 3, enc[2], cfg.lidar_group_samples[2])
        self.LiDAR_lv4 = sa(3, enc[2][-1] + 3, enc[3], cfg.lidar_group_samples[3])
        self.layer_idx = sa(3, cfg.cost_volume_mlps[-1][-1] + 3, enc[4], cfg.lidar_group_samples[4])

        self.RGB_net1 = createCNNs(*cfg.rgb_encoder_channels[0])
        self.RGB_net2 = createCNNs(*cfg.rgb_encoder_channels[1])
        self.RGB_net3 = createCNNs(*cfg.rgb_encoder_channels[2])
        for i, net in enumerate((self.RGB_net1, self.RGB_net2, self.RGB_net3)):
            net.encoder_index = i                              # (bf16 storage mode: which stacks keep bf16 activations)
        # NHWC image encoder: MIOpen's fp32 3x3 conv / BN / pooling kernels run ~1.85x faster in
        # channels_last on MI355X (13.5 -> 7.3 ms fwd+bwd at B=8), and RF3 comes out as [B,h,w,C]
        for net in (self.RGB_net1, self.RGB_net2, self.RGB_net3):
            net.to(memory_format=torch.channels_last)

        def cv(i):
            return CostVolume(H=self.lidar_Hs[2], W=self.lidar_Ws[2], kernel_size=cfg.cost_volume_kernel_size[i],
                              distance=cfg.cost_volume_dis[i], nsample=cfg.cost_volume_nsamples[0],
                              nsample_q=cfg.cost_volume_nsamples[1][i],
                              rgb_in_channels=cfg.rgb_encoder_channels[-1][1][-1], lidar_in_channels=enc[-3][-1],
                              mlp1=cfg.cost_volume_mlps[0], mlp2=cfg.cost_volume_mlps[1],
                              backward_validation=cfg.backward_validation[i], **bn)

        self.cost_volume1 = cv(0)
        self.cost_volume2 = cv(1)

        fp = dict(is_training=self.training, bn_decay=bn_decay, bn=cfg.use_bn_p, use_bn_input=cfg.use_bn_input)
        self.flow_predictor0 = FlowPredictor(in_channels=enc[-2][-1] + enc[-1][-1], mlp=cfg.flow_predictor_mlps[0], **fp)

        def up(i, c_coarse):
            return ProjSetUpconvModule(H=self.lidar_Hs[-1], W=self.lidar_Ws[-1], out_h=self.lidar_Hs[-2],
                                       out_w=self.lidar_Ws[-2], kernel_size=cfg.up_conv_kernel_size[i],
                                       nsample=cfg.setupconv_nsamples[i], stride_H=cfg.stride_Hs[-1],
                                       stride_W=cfg.stride_Ws[-1], distance=cfg.up_conv_dis[i],
                                       in_channels=[enc[-3][-1], c_coarse], mlp=cfg.setupconv_mlps[i][0],
                                       mlp2=cfg.setupconv_mlps[i][1], **bn)

        self.set_upconv0_w_upsample = up(0, cfg.flow_predictor_mlps[0][-1])
        self.set_upconv0_upsample = up(1, enc[-1][-1])
        self.flow_predictor0_predict = FlowPredictor(
            in_channels=enc[-3][-1] + cfg.setupconv_mlps[1][1][-1] + cfg.cost_volume_mlps[-1][-1],
            mlp=cfg.flow_predictor_mlps[1], **fp)
        self.flow_predictor0_w = FlowPredictor(
            in_channels=enc[-3][-1] + cfg.setupconv_mlps[0][-1][-1] + cfg.flow_predictor_mlps[1][-1],
            mlp=cfg.flow_predictor_mlps[2], **fp)

        def head(c_pred, c_feat):
            return PoseHead(in_channels=[c_pred, c_feat], mlp1=[], mlp2=[], hidden=cfg.head_hidden_dim,
                            q_dim=cfg.rotation_quat_head_dim, t_dim=cfg.transition_vec_head_dim,
                            dropout_rate=cfg.head_dropout_rate, split_dp=cfg.split_dp,
                            pos_embed=cfg.head_pos_embedding, sigmoid=cfg.mask_sigmoid, maxhead=cfg.max_head)

        self.l4_head = head(enc[-1][-1], enc[-2][-1])
        self.l3_head = head(cfg.flow_predictor_mlps[1][-1], enc[-3][-1])
        self.l3_head.training_needs_weights = eval_info      # W_l3 is only returned with eval_info

        self.sq = nn.Parameter(torch.tensor([cfg.sq_init]), requires_grad=True)
        self.sx = nn.Parameter(torch.tensor([cfg.sx_init]), requires_grad=True)

    def __getstate__(self):
        # (the second stream and its event are created on first use and are not part of the model: copy.deepcopy / pickling drop them)
        state = self.__dict__.copy()
        state.pop("_side_stream", None)
        state.pop("_lidar_event", None)
        return state

    def _branch_stream(self, dev):
        """second stream for the image encoder (None on the CPU oracle backend or with I2P_ONE_STREAM=1)"""
        if dev.type != "cuda" or os.environ.get("I2P_ONE_STREAM") == "1":
            return None
        s = self.__dict__.get("_side_stream")
        if s is None or s.device != dev:
            s = torch.cuda.Stream(dev)
            self.__dict__["_side_stream"] = s
        return s

    def _image_branch(self, rgb_img, intrinsic, mark=None):
        """image encoder -> (RF3 [B,128,h3,w3], pixel rays [B,M,3], RF3 as points [B,M,C], its unit-variance form); mark = (k, fn):
        fn() is called behind block k (1 .. 15) of the encoder (training path of `_ImageCNN`)"""
        x = rgb_img
        for i, net in enumerate((self.RGB_net1, self.RGB_net2, self.RGB_net3)):
            net._mark = (mark[0] - 1 - 5 * i, mark[1]) if mark is not None else None
            try:
                x = net(x)
            finally:
                net._mark = None
        RF3 = x                                                                 # [B,128,h3,w3]
        pix_index = set_id_grid(RF3.permute(0, 2, 3, 1))                        # [B,M,3]
        # pixel rays in the normalised camera plane of the level-3 feature map
        K3_inv = scaled_intrinsic_inverse(intrinsic, RF3, rgb_img)
        pix_rays = torch.bmm(K3_inv, pix_index.permute(0, 2, 1)).permute(0, 2, 1)   # [B,M,3]
        RF3_pts = RF3.reshape(RF3.shape[0], RF3.shape[1], -1).permute(0, 2, 1)  # [B,M,C]
        return RF3, pix_rays, RF3_pts, _unit_variance(RF3_pts)

    def _lidar_branch(self, lidar_img, lidar_img_raw, lidar_feature, cfg, B, N, dev):
        """spherical projection + the four set-abstraction levels -> everything the cost volumes and heads read"""
        lidar_norm = torch.zeros(B, N, 3, device=dev) if lidar_feature is None else lidar_feature
        raw_img, (feat_img, cam_img) = P.project_seq(lidar_img_raw.float(), [lidar_norm.float(), lidar_img.float()],
                                                     cfg.init_H, cfg.init_W, cfg.rank, cfg.fup, cfg.fdown)
        rfp = cfg.raw_feat_point
        P1_raw, P1, LF1, _, _ = self.LiDAR_lv1.forward_center(raw_img, cam_img, feat_img, cfg=cfg,
                                                            using_intens=cfg.using_intens, raw_feat_point=rfp)
        P2_raw, P2, LF2, _, _ = self.LiDAR_lv2(P1_raw, P1, LF1, cfg=cfg, raw_feat_point=rfp)
        P3_raw, P3, LF3, _, _ = self.LiDAR_lv3(P2_raw, P2, LF2, cfg=cfg, raw_feat_point=rfp)
        P4_raw, P4, LF4, _, sample_idx_4 = self.LiDAR_lv4(P3_raw, P3, LF3, cfg=cfg, raw_feat_point=rfp)
        H3, W3 = self.lidar_Hs[2], self.lidar_Ws[2]
        P3_pts = P3.reshape(B, H3 * W3, 3)
        LF3_pts = LF3.reshape(B, H3 * W3, -1)
        lidar_z = P3_pts[:, :, 2:]
        lidar_uv = P3_pts / (lidar_z + 1e-10)
        return (P3_raw, P3, LF3, P4_raw, P4, LF4, sample_idx_4, P3_pts, LF3_pts, lidar_z, lidar_uv,
                _unit_variance(LF3_pts))

    def forward(self, rgb_img, lidar_img, lidar_img_raw, H_initial, intrinsic, resize_img, gt_project=None,
                calib=None, lidar_feature=None, cfg=None):
        """rgb_img [B,3,h,w]; lidar_img [B,N,3] cloud in the (mis-calibrated) camera frame;
        lidar_img_raw [B,N,3] the same points in the sensor frame (defines the range image);
        intrinsic [B,3,3]; lidar_feature [B,N,D] or None."""
        cfg = cfg or self.cfg
        dev = rgb_img.device
        intrinsic = intrinsic.float()
        B = rgb_img.shape[0]
        N = lidar_img.shape[1]

        side = self._branch_stream(dev)
        if side is None:
            RF3, pix_rays, RF3_pts, RF3_unit = self._image_branch(rgb_img, intrinsic)
            lidar = self._lidar_branch(lidar_img, lidar_img_raw, lidar_feature, cfg, B, N, dev)
        else:
            # the two encoders meet at the first cost volume: the image encoder is issued on a second HIP stream, and autograd runs
            # every backward node on the stream of its forward, so the encoders' backward passes overlap as well (in a captured step
            # they are two branches of the hipGraph).  Everything either branch allocates is its own (per-call scratch, ops.zeros
            # arena slices); the outputs that cross over are handed to the consuming stream's allocator bookkeeping.
            # The grid-barrier chain kernels must have the GPU to themselves, so the point-cloud encoder — the only
            # work that runs next to the image encoder, forward and backward — takes the layer-by-layer kernels; everything behind
            # the join keeps its chains: _BranchJoin's backward is the last node of that part (lowest sequence number, and the
            # autograd engine runs ready nodes highest-first), and the event the engine records behind it on the main stream is what the
            # image encoder's backward waits for, so the second stream is idle whenever a chain kernel runs.
            # The point-cloud encoder starts when the image encoder's fourth block is through (an event of the second stream the main
            # stream waits for; I2P_LIDAR_AFTER_BLOCK = 0 .. 15): the first blocks — the 375 x 1242 and 188 x 621 stages — fill the GPU, and
            # two GPU-filling kernels side by side finish when their sum would, while blocks 6-15 are small library kernels that leave
            # most of the GPU to the other queue.  Same box, 2 x 200 steps: configs[1] 10.23 -> 10.10 ms, configs[2] 11.60 -> 11.55 ms,
            # configs[4] 7.27 -> 7.25 ms (after block 5: 10.14 / 11.63 / 7.31; after block 6: 10.21 / 11.68 / 7.36).
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            after = int(os.environ.get("I2P_LIDAR_AFTER_BLOCK", "4"))
            ev = self.__dict__.get("_lidar_event")            # one event object for the model's life (not one per step)
            if ev is None:
                ev = self.__dict__["_lidar_event"] = torch.cuda.Event()
            reached = []

            def behind_block():
                ev.record(torch.cuda.current_stream(dev))
                reached.append(True)
            with torch.cuda.stream(side):
                image = self._image_branch(rgb_img, intrinsic, (after, behind_block) if 1 <= after <= 15 else None)
            if reached:
                main.wait_event(ev)
            with ops.chains_off():
                lidar = self._lidar_branch(lidar_img, lidar_img_raw, lidar_feature, cfg, B, N, dev)
            main.wait_stream(side)
            for t in image:
                t.record_stream(main)
            RF3, pix_rays, RF3_pts, RF3_unit = image
            RF3_pts, RF3_unit = _BranchJoin.apply(RF3_pts, RF3_unit)
        (P3_raw, P3, LF3, P4_raw, P4, LF4, sample_idx_4, P3_pts, LF3_pts, lidar_z, lidar_uv, LF3_unit) = lidar
        rfp = cfg.raw_feat_point
        H3, W3 = self.lidar_Hs[2], self.lidar_Ws[2]
        H4, W4 = self.lidar_Hs[-1], self.lidar_Ws[-1]
        l3_idx_n2 = P.get_idx_cuda(B, H3, W3, dev)
        normalised = (LF3_unit, RF3_unit)                                       # shared by both cost volumes

        # ---- coarse level --------------------------------------------------------------------
        concat_4 = self.cost_volume1(P3_raw, lidar_uv, LF3_pts, l3_idx_n2, pix_rays, RF3_pts, lidar_z, cfg=cfg,
                                     normalised=normalised)
        _, _, l4_embed, _, _ = self.layer_idx(P3_raw, P3, concat_4, sample_idx=sample_idx_4, cfg=cfg,
                                              raw_feat_point=rfp)
        l4_valid = P.check_valid(P4_raw).view(B, -1, 1)
        l4_mask = self.flow_predictor0(LF4.view(B, H4 * W4, -1), None, l4_embed.view(B, H4 * W4, -1))
        l4_mask = mask_fill(l4_mask, l4_valid)                                  # modellearn_proj_center.py:318
        q4, t4, _ = self.l4_head(l4_embed.view(B, H4 * W4, -1), l4_mask, P4.view(B, H4 * W4, 3),
                                 LF4.view(B, H4 * W4, -1), None)
        result_4 = torch.cat([q4, t4], dim=1)

        # ---- fine level ----------------------------------------------------------------------
        t4_quat = torch.cat([ops.zero_scalar(dev, t4.dtype).expand(B, 1), t4], -1)
        l3_mask_up = self.set_upconv0_w_upsample(P3_raw, P4_raw, P3, P4, l3_idx_n2, LF3,
                                                 l4_mask.view(B, H4, W4, -1), cfg=cfg, raw_feat_point=rfp)
        l3_embed_up = self.set_upconv0_upsample(P3_raw, P4_raw, P3, P4, l3_idx_n2, LF3, l4_embed, cfg=cfg,
                                                raw_feat_point=rfp)
        fine = dict(P3_pts=P3_pts, P3_raw=P3_raw, LF3_pts=LF3_pts, l3_idx_n2=l3_idx_n2, pix_rays=pix_rays, RF3_pts=RF3_pts,
                    normalised=normalised, l3_embed_up=l3_embed_up.view(B, H3 * W3, -1),
                    l3_mask_up=l3_mask_up.view(B, H3 * W3, -1), l3_valid=P.check_valid(P3_raw).view(B, -1, 1),
                    p3_valid=P.check_valid(P3_pts), cfg=cfg)
        out_3, W_l3 = self._refine(fine, q4, t4_quat)

        if self.eval_info:
            return (out_3.float(), result_4.float(), self.sx, self.sq, W_l3, P3_pts, None, None,
                    P4.view(B, H4 * W4, 3))
        return out_3.float(), result_4.float(), None, None, self.sx, self.sq

    def _fine_step(self, f, q_prev, t_prev_quat):
        """one fine registration step (modellearn_proj_center.py:332-404): warp the level-3 cloud by the
        previous estimate, 32-NN cost volume against the image, refine embedding and mask, regress (q3, t3) and
        compose with the previous estimate: q = q3 * q_prev, t = R3 t_prev + t3.
        -> (composed pose [B,7], q3 [B,4], t3 [B,3], mask weights)"""
        B = q_prev.shape[0]
        dev = q_prev.device
        # warp by the previous estimate, mask the empty cells, split off the depth: one launch each way on the device library
        split = warp_utils.warp_split(f["P3_pts"], q_prev, t_prev_quat, f["p3_valid"])
        if split is not None:
            lidar_uv, lidar_z, xyz3 = split
            P3_warped = None                                                     # (the pose head ignores its xyz argument)
        else:
            P3_warped = warp_utils.warp_quat_xyz(f["P3_pts"], q_prev, t_prev_quat) * f["p3_valid"]
            lidar_z = P3_warped[:, :, 2:]
            lidar_uv = P3_warped / (lidar_z + 1e-10)
            xyz3 = None
        concat_3 = self.cost_volume2(f["P3_raw"], lidar_uv, f["LF3_pts"], f["l3_idx_n2"], f["pix_rays"], f["RF3_pts"],
                                     lidar_z, cfg=f["cfg"], normalised=f["normalised"], xyz=xyz3)
        l3_embed = self.flow_predictor0_predict(f["LF3_pts"], f["l3_embed_up"], concat_3.view(B, concat_3.shape[1] * concat_3.shape[2], -1))
        l3_mask = self.flow_predictor0_w(f["LF3_pts"], f["l3_mask_up"], l3_embed)
        l3_mask = mask_fill(l3_mask, f["l3_valid"])                             # :376
        q3, t3, W_l3 = self.l3_head(l3_embed, l3_mask, P3_warped, f["LF3_pts"], None)
        # compose: q = q3 * q_prev, t = R3 t_prev + t3 (modellearn_proj_center.py:388-404)
        return warp_utils.compose_pose(q3, t3, q_prev, t_prev_quat), q3, t3, W_l3

    def _refine(self, fine, q4, t4_quat):
        out_3, _, _, W_l3 = self._fine_step(fine, q4, t4_quat)
        return out_3, W_l3

    def set_bn(self):
        for m in [self.flow_predictor0, self.flow_predictor0_w, self.flow_predictor0_predict, self.LiDAR_lv1,
                  self.LiDAR_lv2, self.LiDAR_lv3, self.LiDAR_lv4, self.layer_idx, self.set_upconv0_upsample,
                  self.set_upconv0_w_upsample, self.cost_volume1, self.cost_volume2]:
            m.set_bn()


def get_num_parameters(model, trainable=False):
    return sum(p.numel() for p in model.parameters() if (p.requires_grad or not trainable))


class RegNet_v2_iter(RegNet_v2):
    """Iterative fine registration (reference: src/modellearn_proj_center_iter.py:346-404): the fine step is
    repeated `n_iters` times.  As in the reference, iteration i >= 1 warps the cloud by the RAW head output
    (q3, t3) of iteration i-1 (not by the composed pose), and the returned pose composes the last (q3, t3) with
    the estimate that iteration started from.  Same parameters / state_dict as RegNet_v2."""

    n_iters = 6

    def _refine(self, fine, q4, t4_quat):
        B = q4.shape[0]
        q_it, t_it = q4, t4_quat
        out_3 = W_l3 = None
        for _ in range(self.n_iters):
            out_3, q3, t3, W_l3 = self._fine_step(fine, q_it, t_it)
            q_it, t_it = q3, torch.cat([ops.zero_scalar(q3.device, t3.dtype).expand(B, 1), t3], -1)
        return out_3, W_l3
