"""Training step of the registration network: forward, learned-uncertainty pose loss, backward,
global-norm clip, Adam — the settings of the reference trainer
(`train20v2learn_wandb_proj.py:198-205` Adam lr 1e-3 betas (0.9,0.999) eps 1e-8 wd 1e-4,
ExponentialLR 0.99/epoch; `:457-483` step order; `--clip 10`).

Data parallel: one process per GPU, `DistributedDataParallel` over RCCL (backend "nccl" on
ROCm); the whole gradient (< 3.4 MB) is one bucket, BN statistics stay local to a rank exactly
like the reference's single-GPU batch of 8 (no SyncBN, `broadcast_buffers=False`).
The step never synchronises with the host (the reference calls `.item()` three times per step).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .config import I2PNetConfig
from .loss import Get_loss
from .model import RegNet_v2


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class Trainer:
    def __init__(self, cfg=I2PNetConfig, device="cuda", lr=1e-3, clip=10.0, world_size=1, local_rank=0,
                 seed=0):
        torch.manual_seed(seed)                 # identical initial weights on every rank
        self.cfg, self.device, self.clip = cfg, torch.device(device), clip
        self.net = RegNet_v2(cfg=cfg).to(self.device)
        self.model = self.net
        if world_size > 1:
            kw = dict(device_ids=[local_rank]) if self.device.type == "cuda" else {}
            self.model = nn.parallel.DistributedDataParallel(
                self.net, bucket_cap_mb=64, broadcast_buffers=False, gradient_as_bucket_view=True, **kw)
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0001)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, 0.99)

    def step(self, batch):
        """one optimisation step on a sample dict (keys of the reference loader); returns the
        loss tensors without synchronising."""
        self.model.train()
        out3, out4, _, _, sx, sq = self.model(batch["rgb"], batch["lidar"], batch["raw_point_xyz"],
                                              batch.get("init_extrinsic"), batch["init_intrinsic"], None, None, None,
                                              batch["lidar_feats"], cfg=self.cfg)
        self.optimizer.zero_grad(set_to_none=True)
        loss, real_loss, dual_loss = Get_loss(out3, out4, batch["decalib_real_gt"], batch["decalib_dual_gt"], sx, sq,
                                              cfg=self.cfg)
        loss.backward()
        if self.clip > 0.0:
            nn.utils.clip_grad_norm_(self.params, self.clip)
        self.optimizer.step()
        return loss.detach(), real_loss.detach(), dual_loss.detach()
