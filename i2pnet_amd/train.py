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
import time

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .config import I2PNetConfig
from .loss import Get_loss
from .model import RegNet_v2


# global-norm clip + Adam as two HIP launches (csrc/optim.hip); I2P_NO_FUSED_ADAM=1: the elementwise torch formulation
USE_FUSED_ADAM = os.environ.get("I2P_NO_FUSED_ADAM", "0") != "1"


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


# hipGraph capture mode.  With a process group alive, the RCCL watchdog thread polls its work events
# (hipEventQuery) at any time; under the default "global" mode such a call from ANOTHER thread during a capture is an
# error that invalidates the capture and kills the watchdog (SIGABRT, seen in ~1 of 4 launches).  "thread_local"
# restricts the check to the capturing thread.  That covers the stream the capture is begun on; a stream that JOINS it later (the model's
# second encoder stream) is not exempt on ROCm 7.0: a watchdog query while such a capture runs invalidates it ("capture failed, staying
# eager"), leaves the joined stream capturing, the next all-reduce's event is then "recorded in a capturing stream" and the watchdog
# dies on it (SIGABRT at the next destroy_process_group(); 2 of 3 runs of the GPU suite).  The watchdog only queries the events of work
# it has not retired yet, so `capture()` lets it retire the warm-up steps' all-reduces first (_quiesce_watchdog) — graph A contains no
# collective, and graph B is captured on one stream.
_CAPTURE_MODE = "thread_local"


def _quiesce_watchdog(device):
    """nothing outstanding for the process group's watchdog thread to poll: device idle, then a few of its 100 ms cycles"""
    torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        time.sleep(0.5)


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
        # 1 where the segment belongs to a parameter that receives a gradient, 0 where autograd leaves `.grad` None:
        # torch.optim.Adam skips such parameters entirely (no decay, no moments); with g*mask == 0 both moments stay
        # zero and the update is 0 / (0 + eps) = 0
        self.mask = None
        self._partials = self._total = None

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_t, "lr": self.lr_t}

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.step_t.copy_(sd["step"]); self.lr_t.copy_(sd["lr"])

    @torch.no_grad()
    def step(self, gate=None):
        """`gate`: 0-d bool tensor or None; False leaves parameters, both moments and the step count untouched (a poisoned step,
        see Trainer._update) without a host synchronisation.  The gate multiplies the INPUTS of the update (step increment, lerp
        weight, decay of the second moment, learning rate): with gate = 0 every line below is an exact identity, no model-sized
        copies are made"""
        g = self.grad
        on = 1.0 if gate is None else gate.to(torch.float32)
        if gate is not None:                  # a poisoned gradient may hold NaN / inf: 0 * NaN would leak through the zero weights
            g = torch.where(gate, g, torch.zeros((), dtype=g.dtype, device=g.device))
        self.step_t += on
        if self.weight_decay != 0.0:
            g = g.add(self.param, alpha=self.weight_decay)
        if self.mask is not None:
            g = g * self.mask
        if gate is None:
            self.exp_avg.lerp_(g, 1.0 - self.beta1)
            self.exp_avg_sq.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
        else:
            self.exp_avg.lerp_(g, on * (1.0 - self.beta1))                        # weight 0: exp_avg + 0 * (g - exp_avg)
            self.exp_avg_sq.mul_(1.0 - on * (1.0 - self.beta2)).add_(g * g * (on * (1.0 - self.beta2)))
        # (a gated first step has step_t = 0: the bias corrections are 0; clamp so that lr_t * 0 / tiny stays 0, not NaN)
        bc1 = (1.0 - torch.pow(self.beta1, self.step_t)).clamp_min(1e-30)
        bc2_sqrt = (1.0 - torch.pow(self.beta2, self.step_t)).clamp_min(1e-30).sqrt()
        denom = (self.exp_avg_sq.sqrt() / bc2_sqrt).add_(self.eps)
        self.param.sub_(self.exp_avg / denom * (self.lr_t * on / bc1))

    @torch.no_grad()
    def fused_clip_step(self, clip, gscale=1.0, poison=None):
        """average (gscale = 1/world), clip to global norm `clip` and take the Adam step in two launches of
        i2p_clip_adam (csrc/optim.hip) instead of ~25 elementwise launches; same arithmetic as `_update()`'s torch
        formulation (tests/test_train_gpu.py).  HIP backend / device tensors only.  `poison`: device fp32 [>= 1] or None — a
        non-zero value makes the call a no-op (Trainer: the chain kernels' error counter behind the all-reduced gradient)."""
        be = ops.get_backend()
        if self._partials is None:
            self._partials = torch.zeros(256, dtype=torch.float64, device=self.param.device)
            self._total = torch.zeros(1, dtype=torch.float32, device=self.param.device)
        P = lambda t, dt=torch.float32: be._p(t, dt, "adam")
        be._call("i2p_clip_adam", int(self.param.numel()), P(self.param), P(self.grad), P(self.exp_avg), P(self.exp_avg_sq),
                 P(self.mask) if self.mask is not None else None, P(self._partials, torch.float64), P(self.step_t.view(1)),
                 P(self.lr_t.view(1)), float(self.beta1), float(self.beta2), float(self.eps), float(self.weight_decay), float(clip),
                 float(gscale), P(self._total), P(poison) if poison is not None else None, stream=be._stream())

    @torch.no_grad()
    def decay_lr(self, gamma):
        self.lr_t.mul_(gamma)


# what a step reads of a loader sample dict (kitti_odometry_corr_lidarnone_proj.py:757-789 also carries strings
# such as `path_info` and unused tensors such as `resize_img`: they never reach the device)
STEP_KEYS = ("rgb", "lidar", "raw_point_xyz", "lidar_feats", "init_intrinsic", "init_extrinsic", "decalib_real_gt",
             "decalib_dual_gt")


class Trainer:
    def __init__(self, cfg=I2PNetConfig, device="cuda", lr=1e-3, clip=10.0, world_size=1, local_rank=0,
                 seed=0, capturable=None, net_cls=RegNet_v2, call=None, on_chain_error="raise"):
        """`capturable` is accepted for compatibility and ignored (FlatAdam is always graph-safe).
        `net_cls` / `call`: another registration network with the same outputs (e.g. the small-range model,
        i2pnet_amd.small_range.RegNet_v2) and how to call it: call(net, batch, cfg) -> its output tuple.
        `on_chain_error`: what `step()` does when a one-launch MLP chain reported an abandoned grid barrier (its grid was not
        co-resident — CU mask, another process on the GPU; csrc/mlp_chain.hip): "raise" ops.ChainBarrierTimeout (default), or
        "fallback": log, switch the chains off (I2P_NO_CHAIN=1: the layer-by-layer kernels), re-capture and go on (single-process
        training only: with several ranks a rank-local re-capture would desynchronise the collectives, so it always raises).
        Either way the poisoned step is never applied: the error counter rides at the end of the flat gradient, through the
        all-reduce, and a non-zero value turns clip + Adam into a no-op on every rank (i2p_clip_adam `poison`; the torch fallback of
        `_update` gates its update the same way).  The word is the counter's increase over THIS step, so a transient time-out skips
        one step.  The faulty rank raises at its next `step()` (host-mapped flag); with world_size > 1 every rank reads the all-reduced
        word of step k at the top of step k + 2 and raises there (`_check_reduced_poison`), and check_chain_errors(sync=True)
        (epoch_end, save_checkpoint) reads it as well."""
        torch.manual_seed(seed)                 # identical initial weights on every rank
        self.cfg, self.device, self.clip = cfg, torch.device(device), clip
        self.world_size = world_size
        self._call = call
        self.net = net_cls(cfg=cfg).to(self.device)
        self.model = self.net
        # every parameter the reference's torch.optim.Adam steps — i.e. all of them (train20v2learn_wandb_proj.py:198): conv biases
        # in front of a batch-statistics BN (`_i2p_cancelled`, modules.Conv2d / createCNNs) have an exactly-zero gradient and never
        # enter autograd here, but the reference still moves them by weight decay (g = 0 + 1e-4 p, i.e. ~lr per step through Adam's
        # normalisation): they sit in the flat buffers with a zero gradient so that parameters evolve as in the reference
        self.params = [p for p in self.net.parameters() if p.requires_grad or getattr(p, "_i2p_cancelled", False)]
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
        # the gradient buffer carries four more floats: [n] = the chain kernels' error counter of this step (0 on a healthy GPU),
        # summed over the ranks by the same all-reduce; `flat_grad` is the view the optimiser and named_grads() see
        self._grad_buf = torch.zeros(n + 4, dtype=torch.float32, device=self.device)
        self.flat_grad = self._grad_buf[:n]
        self._poison = self._grad_buf[n:]
        self.on_chain_error = on_chain_error
        self._chain_words = None                # (device counter fp32 [4], pinned host flag) of this device, HIP backend only
        if self.device.type == "cuda" and ops.get_backend().name == "hip":
            with torch.cuda.device(self.device):
                self._chain_words = ops.chain_error_words(self.device)
        # the poison word of a step is the counter's INCREASE over that step (snapshot at the step's start): one transient
        # time-out poisons one step, not every later one
        self._poison_base = torch.zeros(4, dtype=torch.float32, device=self.device)
        # world_size > 1: the all-reduced poison word of step k is copied to pinned memory after the update and read at the top of
        # step k + 2 (its copy has long finished: no stall) — EVERY rank then raises at the same step, instead of the healthy ranks
        # hanging in an all-reduce the faulty rank (which sees its host flag one step earlier) never joins
        self._poison_ring = None
        if self._chain_words is not None and (world_size > 1 or os.environ.get("I2P_FORCE_DP")):
            self._poison_ring = ([torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(2)], [None, None])
        self._steps_issued = 0
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
        self._lr0 = lr
        self._graph_a = self._graph_b = None
        self._static = None
        self._static_out = None
        self._mask_known = False

    # ---- the three pieces of a step -----------------------------------------------------------------
    def _forward_backward(self, batch):
        self.model.train()
        ops.begin_step(self.device)             # one memset for every small accumulator of this step
        # the weight gradients' slab sums are recorded during backward and formed in ONE launch behind it (csrc/deferred.hip): nothing
        # reads a weight gradient before the packing below
        self._deferring = self.device.type == "cuda" and ops.defer_begin()
        try:
            return self._forward_backward_in_step(batch)
        finally:
            # the arena belongs to THIS step: outside, ops.zeros() is torch.zeros again (a later graph capture — the
            # evaluator's, bench_infer's — would otherwise bake arena slices in as "zero" accumulators without a memset)
            ops.end_step(self.device)
            if self._deferring:
                self._deferring = False
                ops.defer_end()

    def _forward_backward_in_step(self, batch):
        for p in self.params:                   # autograd then hands its buffers over instead of accumulating
            p.grad = None
        if self._chain_words is not None:
            self._poison_base.copy_(self._chain_words[0])
        if self._call is not None:
            out3, out4, _, _, sx, sq = self._call(self.model, batch, self.cfg)
        else:
            out3, out4, _, _, sx, sq = self.model(batch["rgb"], batch["lidar"], batch["raw_point_xyz"],
                                                  batch.get("init_extrinsic"), batch["init_intrinsic"], None, None, None,
                                                  batch["lidar_feats"], cfg=self.cfg)
        loss, real_loss, dual_loss = Get_loss(out3, out4, batch["decalib_real_gt"], batch["decalib_dual_gt"], sx, sq,
                                              cfg=self.cfg)
        loss.backward()
        if getattr(self, "_deferring", False):
            # every weight gradient whose sum is still pending must have become a parameter's `.grad` UNTOUCHED (adopted, not cloned,
            # not concatenated or added by autograd): anything else would have read it before it exists
            have = {p.grad.data_ptr() for p in self.params if p.grad is not None}
            early = [x for x in ops.defer_noted() if x not in have]
            if early:
                raise RuntimeError(f"{len(early)} weight gradients with a deferred reduction were consumed before the flush "
                                   "(autograd post-processed them): wrap their producer in ops.defer_paused() or set I2P_NO_DEFER=1")
            ops.defer_flush()
        zero = self._zero
        grads = []
        if not self._mask_known:                # first step: which parameters does this loss reach at all? (host-side
            self._mask_known = True             # knowledge: no synchronisation)
            unreached = lambda p: p.grad is None and not getattr(p, "_i2p_cancelled", False)
            if any(unreached(p) for p in self.params):
                mask = torch.ones_like(self.flat_param)
                for p, off in zip(self.params, self._offsets):
                    if unreached(p):
                        mask[off:off + (p.numel() + 3) // 4 * 4] = 0.0
                self.optimizer.mask = mask
        for p, nhwc in zip(self.params, self._nhwc):
            if p.grad is None:                  # parameter not reached by this loss
                grads.append(zero.expand(p.numel()))
            else:
                grads.append((p.grad.permute(0, 2, 3, 1) if nhwc else p.grad).reshape(-1))
            if p.numel() % 4:                   # alignment padding of the flat layout
                grads.append(zero.expand(4 - p.numel() % 4))
        grads.append(self._chain_words[0] - self._poison_base if self._chain_words is not None else zero.expand(4))
        torch.cat(grads, out=self._grad_buf)
        return loss.detach(), real_loss.detach(), dual_loss.detach()

    def named_grads(self):
        """the gradients the optimiser consumed last step (all-reduced, averaged, clipped), by parameter name"""
        names = [k for k, p in self.net.named_parameters() if p.requires_grad or getattr(p, "_i2p_cancelled", False)]
        out = {}
        for k, p, nhwc, off in zip(names, self.params, self._nhwc, self._offsets):
            seg = self.flat_grad[off:off + p.numel()]
            out[k] = (seg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if nhwc else seg.view_as(p)).clone()
        return out

    def epoch_end(self):
        self.check_chain_errors(sync=True)
        self.optimizer.decay_lr(self.lr_gamma)

    # ---- abandoned grid barriers of the chain kernels reach the caller (VERDICT r3 #1c / ADVICE r3) ----------------------------------
    def check_chain_errors(self, sync=False):
        """Raise (or fall back, see `on_chain_error`) if a chain launch of this device reported a timed-out grid barrier.
        sync=False reads the host-mapped flag only — free, called at the top of every `step()`, sees every launch that has run
        so far; sync=True also synchronises and reads the device counter (epoch_end, save_checkpoint)."""
        if self._chain_words is None:
            return
        # sync=True also reads the ALL-REDUCED poison word of the last step: with world_size > 1 only the faulty rank's host flag is
        # set, the healthy ranks see the fault through the reduced gradient buffer and raise with it (instead of skipping steps
        # silently and then hanging in the next all-reduce).  The counter is cumulative per device until chain_errors_reset().
        bad = ops.chain_error_flag(self.device) or (sync and (ops.chain_errors(self.device) != 0 or float(self._poison[0]) != 0.0))
        if not bad:
            return
        torch.cuda.synchronize(self.device)
        count = ops.chain_errors(self.device)
        msg = (f"{count} chain launch(es) on {self.device} abandoned a grid barrier: the grid was not co-resident (CU mask, another "
               "process on the GPU, overlapping chain launches).  The affected optimisation steps were NOT applied (clip + Adam skip "
               "a poisoned gradient).  I2P_NO_CHAIN=1 runs the layer-by-layer kernels instead.")
        if self.on_chain_error == "fallback" and self.world_size == 1 and not dist.is_initialized():
            print("[i2pnet_amd.train] " + msg + "  Falling back to I2P_NO_CHAIN=1 and re-capturing.", flush=True)
            os.environ["I2P_NO_CHAIN"] = "1"
            ops.chain_errors_reset(self.device)
            if self._graph_a is not None:
                self._graph_a = self._graph_b = None
                self.capture(self._static)
            return
        raise ops.ChainBarrierTimeout(msg)

    def _post_poison(self):
        """after the update of step k: the all-reduced poison word to pinned memory, asynchronously"""
        if self._poison_ring is None:
            return
        host, events = self._poison_ring
        slot = self._steps_issued % 2
        host[slot].copy_(self._poison[:1], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.device))
        events[slot] = ev
        self._steps_issued += 1

    def _check_reduced_poison(self):
        """top of step k + 2: did any rank's chain kernels abandon a barrier in step k?  Same answer on every rank."""
        if self._poison_ring is None or self._steps_issued < 2:
            return
        host, events = self._poison_ring
        slot = self._steps_issued % 2              # the slot written two steps ago (about to be re-used)
        events[slot].synchronize()
        if float(host[slot][0]) != 0.0:
            raise ops.ChainBarrierTimeout(
                f"step {self._steps_issued - 2}: {int(host[slot][0])} chain launch(es) across the {self.world_size} ranks abandoned a grid "
                "barrier; that step was not applied on any rank.  Raised on every rank at the same step.")

    def _all_reduce(self):
        if self.world_size > 1 or (os.environ.get("I2P_FORCE_DP") and dist.is_initialized()):
            dist.all_reduce(self._grad_buf, op=dist.ReduceOp.SUM)

    def _update(self):
        be = ops.get_backend()
        if (USE_FUSED_ADAM and self.device.type == "cuda" and be.name == "hip" and self.flat_param.numel() % 4 == 0
                and self.flat_param.data_ptr() % 16 == 0 and self.flat_grad.data_ptr() % 16 == 0):
            self.optimizer.fused_clip_step(self.clip, 1.0 / self.world_size, poison=self._poison if self._chain_words is not None else None)
            return
        if self.world_size > 1:
            self.flat_grad.mul_(1.0 / self.world_size)
        if self.clip > 0.0:                     # clip_grad_norm_ on the flat buffer (same total norm)
            total = torch.linalg.vector_norm(self.flat_grad)
            self.flat_grad.mul_(torch.clamp(self.clip / (total + 1e-6), max=1.0))
        # (the un-fused path honours the poison word too: a step whose chain kernels abandoned a barrier is never applied)
        self.optimizer.step(gate=(self._poison[0] == 0) if self._chain_words is not None else None)

    def _to_device(self, batch):
        """the tensors a step reads, on the trainer's device as fp32 (the reference moves each entry explicitly,
        train20v2learn_wandb_proj.py:440-450); everything else in the sample dict is ignored"""
        out = {}
        for k in STEP_KEYS:
            v = batch.get(k)
            if isinstance(v, torch.Tensor):
                if v.device != self.device or v.dtype != torch.float32:
                    v = v.to(self.device, dtype=torch.float32, non_blocking=True)
                out[k] = v
        return out

    def _eager_step(self, batch):
        batch = self._to_device(batch)
        out = self._forward_backward(batch)
        self._all_reduce()
        self._update()
        return out

    # ---- hipGraph capture ---------------------------------------------------------------------------------
    def capture(self, batch, warmup=3):
        """Capture the step as hipGraphs (static shapes).  `batch` provides the static input buffers; later
        `step()` calls copy into them and replay.  Returns True if the graphs are live, False if capture failed
        (the trainer then keeps running eagerly).  The `warmup` eager steps (allocator, MIOpen solver search) and the
        capture pass itself run on `batch`, but parameters, Adam state, BN buffers and the RNG state are restored
        afterwards: the first `step()` after `capture()` is optimisation step 1, as in eager training."""
        assert self.device.type == "cuda"
        self._static = {k: v.clone() for k, v in self._to_device(batch).items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snap = self._snapshot()
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager_step(self._static)
            torch.cuda.current_stream().wait_stream(side)
            _quiesce_watchdog(self.device)
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
            torch.cuda.synchronize()
            self._restore(snap)
            return True
        except Exception as e:                      # noqa: BLE001 — fall back to eager, loudly
            print(f"[i2pnet_amd.train] hipGraph capture failed, staying eager: {type(e).__name__}: {e}", flush=True)
            self._graph_a = self._graph_b = None
            # A stream that had joined the failed capture (the model's second encoder stream) may be left in capture mode: eager work issued
            # on it afterwards would be recorded instead of run, and pull every stream that waits for it into the dead capture.  The eager
            # steps that follow get a fresh stream (and the event that goes with it).
            for m in self.net.modules():
                m.__dict__.pop("_side_stream", None); m.__dict__.pop("_lidar_event", None)
            torch.cuda.synchronize()
            self._restore(snap)
            return False

    def _snapshot(self):
        return {"param": self.flat_param.clone(), "opt": {k: v.clone() for k, v in self.optimizer.state_dict().items()},
                "buffers": [b.clone() for b in self.net.buffers()], "frozen": [p.detach().clone() for p in self.net.parameters() if not p.requires_grad],
                "rng": torch.cuda.get_rng_state(self.device), "rng_cpu": torch.get_rng_state()}

    @torch.no_grad()
    def _restore(self, snap):
        self.flat_param.copy_(snap["param"])
        self.optimizer.load_state_dict(snap["opt"])
        for b, v in zip(self.net.buffers(), snap["buffers"]):
            b.copy_(v)
        for p, v in zip([p for p in self.net.parameters() if not p.requires_grad], snap["frozen"]):
            p.copy_(v)
        torch.cuda.set_rng_state(snap["rng"], self.device); torch.set_rng_state(snap["rng_cpu"])

    # ---- checkpoints: the reference's dict layout (train20v2learn_wandb_proj.py:254-268) ------------------------------------
    # `optimizer_state_dict` / `scheduler_state_dict` are written in torch.optim.Adam's / ExponentialLR's own state_dict layout
    # over ALL parameters of the network in `parameters()` order — what the reference trainer's optimizer.load_state_dict /
    # scheduler.load_state_dict (:218-221) accept — sliced from the flat moment buffers; parameters that take no part in
    # training here (conv biases in front of batch-statistics BNs) carry zero moments.
    def _adam_state_dict(self, epoch):
        opt = self.optimizer
        index = {id(p): i for i, p in enumerate(self.params)}
        state, ids = {}, []
        for j, p in enumerate(self.net.parameters()):
            ids.append(j)
            i = index.get(id(p))
            if i is None:
                m = v = torch.zeros_like(p, device="cpu")
            else:
                off, nhwc = self._offsets[i], self._nhwc[i]
                def view(buf):
                    seg = buf[off:off + p.numel()]
                    t = seg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if nhwc else seg.view_as(p)
                    return t.detach().cpu().contiguous()
                m, v = view(opt.exp_avg), view(opt.exp_avg_sq)
            state[j] = {"step": opt.step_t.detach().cpu().clone(), "exp_avg": m, "exp_avg_sq": v}
        lr = float(opt.lr_t)
        group = {"lr": lr, "betas": (opt.beta1, opt.beta2), "eps": opt.eps, "weight_decay": opt.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "initial_lr": self._lr0, "params": ids}
        sched = {"gamma": self.lr_gamma, "base_lrs": [self._lr0], "last_epoch": int(epoch), "verbose": False, "_step_count": int(epoch) + 1,
                 "_get_lr_called_within_step": False, "_last_lr": [lr]}
        return {"state": state, "param_groups": [group]}, sched

    def save_checkpoint(self, path, epoch=0):
        self.check_chain_errors(sync=True)       # never write weights of a run whose chains computed garbage without saying so
        opt_sd, sched_sd = self._adam_state_dict(epoch)
        torch.save({"epoch": epoch, "model_state_dict": self.net.state_dict(), "optimizer_state_dict": opt_sd,
                    "scheduler_state_dict": sched_sd}, path)

    def load_checkpoint(self, path, trust_pickle=False):
        """a checkpoint of this trainer OR of the reference trainer (torch.optim.Adam + ExponentialLR state dicts): weights,
        Adam moments / step count and the decayed learning rate are all restored.  The layout holds tensors and plain
        containers only, so the file is read with `weights_only=True`; `trust_pickle=True` opts into full unpickling for
        third-party files that carry other objects (arbitrary code execution: only for files you trust)."""
        if trust_pickle:
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
        else:
            # the reference trainer's ckpt.pt (train20v2learn_wandb_proj.py:255-268) also stores `training_params` (plain containers)
            # and best_rotation_error / best_transition_error / best_acc, which are numpy float64 scalars: allow exactly those
            # numpy reconstructors next to tensors and containers, nothing else
            import numpy as np
            allowed = [np.dtype, np.float64, np.float32, np.int64, np.ndarray]
            for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
                try:
                    m = __import__(mod, fromlist=["scalar"])
                    allowed += [m.scalar, m._reconstruct]
                except (ImportError, AttributeError):
                    pass
            allowed += [type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64))]
            try:
                with torch.serialization.safe_globals(allowed):
                    ckpt = torch.load(path, map_location="cpu", weights_only=True)
            except Exception as e:     # (pickle.UnpicklingError and friends)
                raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}).  If the file is trusted and "
                                   "carries other pickled objects, call load_checkpoint(path, trust_pickle=True).") from e
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in ckpt["model_state_dict"].items()}
        with torch.no_grad():                       # parameters are views into the flat buffer: copy, never rebind
            self.net.load_state_dict(sd, strict=True)
        opt = ckpt.get("optimizer_state_dict")
        if opt is not None and "exp_avg" in opt:    # (round-1/2 checkpoints of this trainer: the flat buffers themselves)
            self.optimizer.load_state_dict({k: v.to(self.device) for k, v in opt.items()})
        elif opt is not None and "state" in opt:
            index = {id(p): i for i, p in enumerate(self.params)}
            ids = opt["param_groups"][0]["params"]
            with torch.no_grad():
                step = None
                for j, p in zip(ids, self.net.parameters()):
                    st = opt["state"].get(j)
                    i = index.get(id(p))
                    if st is None or i is None:
                        continue
                    off, nhwc = self._offsets[i], self._nhwc[i]
                    for key, buf in (("exp_avg", self.optimizer.exp_avg), ("exp_avg_sq", self.optimizer.exp_avg_sq)):
                        seg = buf[off:off + p.numel()]
                        dst = seg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if nhwc else seg.view_as(p)
                        dst.copy_(st[key].to(self.device).view_as(p))
                    step = st["step"] if step is None else step
                if step is not None:
                    self.optimizer.step_t.fill_(float(step))
                self.optimizer.lr_t.fill_(float(opt["param_groups"][0]["lr"]))
        sched = ckpt.get("scheduler_state_dict")
        if sched is not None and "_last_lr" in sched and not (opt is not None and "state" in opt):
            self.optimizer.lr_t.fill_(float(sched["_last_lr"][0]))
        return ckpt.get("epoch", 0)

    def step(self, batch):
        """one optimisation step on a sample dict (keys of the reference loader); returns the
        loss tensors without synchronising."""
        if self._poison_ring is not None:         # world_size > 1: every rank raises together, two steps after the fault — the faulty
            self._check_reduced_poison()          # rank must not leave earlier on its own host flag (the others would hang in the all-reduce)
        else:
            self.check_chain_errors()             # host-mapped flag of the chain kernels: no synchronisation
        if self._graph_a is not None:
            dsts, srcs = [], []
            for k, dst in self._static.items():
                v = batch[k]
                if dst is v:
                    continue
                if v.device == dst.device and v.dtype == dst.dtype and v.shape == dst.shape:
                    dsts.append(dst); srcs.append(v)
                else:
                    dst.copy_(v, non_blocking=True)
            if dsts:            # the step's eight input tensors into the graph's static buffers: ONE launch, not eight memcpy nodes (~5 us each)
                torch._foreach_copy_(dsts, srcs)
            self._graph_a.replay()
            if self._graph_b is not None:
                self._all_reduce()
                self._graph_b.replay()
            self._post_poison()
            return self._static_out
        out = self._eager_step(batch)
        self._post_poison()
        return out
