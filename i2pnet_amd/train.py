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
                 seed=0, capturable=False):
        torch.manual_seed(seed)                 # identical initial weights on every rank
        self.cfg, self.device, self.clip = cfg, torch.device(device), clip
        self.net = RegNet_v2(cfg=cfg).to(self.device)
        self.model = self.net
        if world_size > 1:
            kw = dict(device_ids=[local_rank]) if self.device.type == "cuda" else {}
            self.model = nn.parallel.DistributedDataParallel(
                self.net, bucket_cap_mb=64, broadcast_buffers=False, gradient_as_bucket_view=True, **kw)
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0001,
                                          capturable=capturable and self.device.type == "cuda")
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, 0.99)
        self._graph = None
        self._static = None
        self._static_out = None

    # ---- whole-step hipGraph ------------------------------------------------------------------
    def capture(self, batch, warmup=3):
        """Capture forward+loss+backward+clip+Adam as ONE hipGraph (static shapes; ~2000 kernel
        launches per step otherwise keep the host thread on the critical path).  `batch` provides
        the static input buffers; later `step()` calls copy into them and replay.
        Requires `capturable=True`.  Returns True if the graph is live, False if capture failed
        (the trainer then keeps running eagerly)."""
        assert self.device.type == "cuda"
        self._static = {k: v.clone() for k, v in batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager_step(self._static)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._static_out = self._eager_step(self._static)
            self._graph = graph
            return True
        except Exception as e:                      # noqa: BLE001 — fall back to eager, loudly
            print(f"[i2pnet_amd.train] hipGraph capture failed, staying eager: {type(e).__name__}: {e}", flush=True)
            self._graph = None
            torch.cuda.synchronize()
            return False

    def step(self, batch):
        """one optimisation step on a sample dict (keys of the reference loader); returns the
        loss tensors without synchronising."""
        if self._graph is not None:
            for k, v in batch.items():
                if self._static[k] is not v:
                    self._static[k].copy_(v, non_blocking=True)
            self._graph.replay()
            return self._static_out
        return self._eager_step(batch)

    def _eager_step(self, batch):
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
