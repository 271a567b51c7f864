"""Training step of the registration network: forward, learned-uncertainty pose loss, backward,
global-norm clip, Adam — the settings of the reference trainer
(`train20v2learn_wandb_proj.py:198-205` Adam lr 1e-3 betas (0.9,0.999) eps 1e-8 wd 1e-4,
ExponentialLR 0.99/epoch; `:457-483` step order; `--clip 10`).

Data parallel (the reference is single-GPU): one process per GPU.  Parameters, Adam
moments and the gathered gradient each live in ONE flat fp32 buffer (3.4 MB; the module parameters are
views into it), so a step is

    [hipGraph A: forward, loss, backward, pack gradients into the flat buffer]
    ->  one RCCL all-reduce of the flat buffer
    ->  [hipGraph B: average, clip global norm, Adam on the flat buffers (a dozen elementwise kernels)]

i.e. two graph replays and one collective per step on the host thread (≈ 2000 kernel launches
otherwise), no collective inside a captured graph, no bucketing machinery.  BN statistics stay local
to a rank exactly like the reference's single-GPU batch of 8 (no SyncBN).  With one GPU the two
graphs are captured as one.  The step never synchronises with the host (the reference calls
`.item()` three times per step).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .config import I2PNetConfig
from .loss import Get_loss
from .model import RegNet_v2


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


# hipGraph capture mode.  With a process group alive, the RCCL watchdog thread polls its work events
# (hipEventQuery) at any time; under the default "global" mode such a call from ANOTHER thread during a capture is an
# error that invalidates the capture and kills the watchdog (SIGABRT, seen in ~1 of 4 launches).  "thread_local"
# restricts the check to the capturing thread.
_CAPTURE_MODE = "thread_local"


def init_distributed(backend):
    rank, local_rank, world = dist_env()
    if (world > 1 or os.environ.get("I2P_FORCE_DP")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class FlatAdam:
    """Adam with L2 weight decay (the arithmetic of `torch.optim.Adam`, no amsgrad) on flat buffers;
    step count and learning rate are device scalars so the update can sit in a hipGraph and the
    ExponentialLR decay is one multiply."""

    def __init__(self, param, grad, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.param, self.grad = param, grad
        self.beta1, self.beta2, self.eps, self.weight_decay = betas[0], betas[1], eps, weight_decay
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.step_t = torch.zeros((), dtype=torch.float32, device=param.device)
        self.lr_t = torch.full((), lr, dtype=torch.float32, device=param.device)

    @torch.no_grad()
    def step(self):
        g = self.grad
        self.step_t += 1.0
        if self.weight_decay != 0.0:
            g = g.add(self.param, alpha=self.weight_decay)
        self.exp_avg.lerp_(g, 1.0 - self.beta1)
        self.exp_avg_sq.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
        bc1 = 1.0 - torch.pow(self.beta1, self.step_t)
        bc2_sqrt = (1.0 - torch.pow(self.beta2, self.step_t)).sqrt()
        denom = (self.exp_avg_sq.sqrt() / bc2_sqrt).add_(self.eps)
        self.param.sub_(self.exp_avg / denom * (self.lr_t / bc1))

    @torch.no_grad()
    def decay_lr(self, gamma):
        self.lr_t.mul_(gamma)


class Trainer:
    def __init__(self, cfg=I2PNetConfig, device="cuda", lr=1e-3, clip=10.0, world_size=1, local_rank=0,
                 seed=0, capturable=False, net_cls=RegNet_v2, call=None):
        """`net_cls` / `call`: another registration network with the same outputs (e.g. the small-range model,
        i2pnet_amd.small_range.RegNet_v2) and how to call it: call(net, batch, cfg) -> its output tuple."""
        torch.manual_seed(seed)                 # identical initial weights on every rank
        self.cfg, self.device, self.clip = cfg, torch.device(device), clip
        self.world_size = world_size
        self._call = call
        self.net = net_cls(cfg=cfg).to(self.device)
        self.model = self.net
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        if world_size > 1:                      # belt and braces: same seed already gives identical replicas
            for p in self.net.parameters():
                dist.broadcast(p.data, src=0)
            for b in self.net.buffers():
                dist.broadcast(b.data, src=0)
        # trainable parameters become views into one flat buffer; gradients are packed into another:
        # one all-reduce, one norm, one Adam
        # every parameter starts on a 16-byte boundary of the flat buffers (the layer kernels stage weights with
        # 16-byte loads when they can); the padding elements stay zero in all four buffers
        self._offsets, n = [], 0
        for p in self.params:
            self._offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._zero = torch.zeros((), dtype=torch.float32, device=self.device)
        self._nhwc = []                         # 4-D parameters kept in channels_last storage (image-encoder conv weights)
        with torch.no_grad():
            for p, off in zip(self.params, self._offsets):
                seg = self.flat_param[off:off + p.numel()]
                nhwc = p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last)
                self._nhwc.append(nhwc)
                # the view keeps the parameter's memory format (MIOpen's NHWC kernels would otherwise re-layout the
                # weights on every call); flat order = storage order, so gradients are packed in the same order
                view = seg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if nhwc else seg.view_as(p)
                view.copy_(p.data)
                p.data = view
        self.optimizer = FlatAdam(self.flat_param, self.flat_grad, lr, betas=(0.9, 0.999), eps=1e-08,
                                  weight_decay=0.0001)
        self.lr_gamma = 0.99                    # ExponentialLR(0.99) per epoch: call `epoch_end()`
        self._graph_a = self._graph_b = None
        self._static = None
        self._static_out = None

    # ---- the three pieces of a step -----------------------------------------------------------------
    def _forward_backward(self, batch):
        self.model.train()
        ops.begin_step(self.device)             # one memset for every small accumulator of this step
        for p in self.params:                   # autograd then hands its buffers over instead of accumulating
            p.grad = None
        if self._call is not None:
            out3, out4, _, _, sx, sq = self._call(self.model, batch, self.cfg)
        else:
            out3, out4, _, _, sx, sq = self.model(batch["rgb"], batch["lidar"], batch["raw_point_xyz"],
                                                  batch.get("init_extrinsic"), batch["init_intrinsic"], None, None, None,
                                                  batch["lidar_feats"], cfg=self.cfg)
        loss, real_loss, dual_loss = Get_loss(out3, out4, batch["decalib_real_gt"], batch["decalib_dual_gt"], sx, sq,
                                              cfg=self.cfg)
        loss.backward()
        zero = self._zero
        grads = []
        for p, nhwc in zip(self.params, self._nhwc):
            if p.grad is None:                  # parameter not reached by this loss
                grads.append(zero.expand(p.numel()))
            else:
                grads.append((p.grad.permute(0, 2, 3, 1) if nhwc else p.grad).reshape(-1))
            if p.numel() % 4:                   # alignment padding of the flat layout
                grads.append(zero.expand(4 - p.numel() % 4))
        torch.cat(grads, out=self.flat_grad)
        return loss.detach(), real_loss.detach(), dual_loss.detach()

    def named_grads(self):
        """the gradients the optimiser consumed last step (all-reduced, averaged, clipped), by parameter name"""
        names = [k for k, p in self.net.named_parameters() if p.requires_grad]
        out = {}
        for k, p, nhwc, off in zip(names, self.params, self._nhwc, self._offsets):
            seg = self.flat_grad[off:off + p.numel()]
            out[k] = (seg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if nhwc else seg.view_as(p)).clone()
        return out

    def epoch_end(self):
        self.optimizer.decay_lr(self.lr_gamma)

    def _all_reduce(self):
        if self.world_size > 1 or (os.environ.get("I2P_FORCE_DP") and dist.is_initialized()):
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)

    def _update(self):
        if self.world_size > 1:
            self.flat_grad.mul_(1.0 / self.world_size)
        if self.clip > 0.0:                     # clip_grad_norm_ on the flat buffer (same total norm)
            total = torch.linalg.vector_norm(self.flat_grad)
            self.flat_grad.mul_(torch.clamp(self.clip / (total + 1e-6), max=1.0))
        self.optimizer.step()

    def _eager_step(self, batch):
        out = self._forward_backward(batch)
        self._all_reduce()
        self._update()
        return out

    # ---- hipGraph capture ---------------------------------------------------------------------------------
    def capture(self, batch, warmup=3):
        """Capture the step as hipGraphs (static shapes).  `batch` provides the static input buffers; later
        `step()` calls copy into them and replay.  Requires `capturable=True`.  Returns True if the graphs
        are live, False if capture failed (the trainer then keeps running eagerly)."""
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
            if self.world_size == 1 and not os.environ.get("I2P_FORCE_DP"):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode=_CAPTURE_MODE):
                    self._static_out = self._forward_backward(self._static)
                    self._update()
                self._graph_a, self._graph_b = graph, None
            else:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, capture_error_mode=_CAPTURE_MODE):
                    self._static_out = self._forward_backward(self._static)
                with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode=_CAPTURE_MODE):
                    self._update()
                self._graph_a, self._graph_b = ga, gb
            return True
        except Exception as e:                      # noqa: BLE001 — fall back to eager, loudly
            print(f"[i2pnet_amd.train] hipGraph capture failed, staying eager: {type(e).__name__}: {e}", flush=True)
            self._graph_a = self._graph_b = None
            torch.cuda.synchronize()
            return False

    def step(self, batch):
        """one optimisation step on a sample dict (keys of the reference loader); returns the
        loss tensors without synchronising."""
        if self._graph_a is not None:
            for k, v in batch.items():
                if self._static[k] is not v:
                    self._static[k].copy_(v, non_blocking=True)
            self._graph_a.replay()
            if self._graph_b is not None:
                self._all_reduce()
                self._graph_b.replay()
            return self._static_out
        return self._eager_step(batch)
