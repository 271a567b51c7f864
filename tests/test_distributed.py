"""Data-parallel step on CPU: 2 processes, gloo, oracle operators.  Checks what the RCCL path
relies on: identical parameters on every rank after a step, and gradients equal to the mean of
the per-rank gradients (flat-buffer all-reduce), with BN statistics local to a rank."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from i2pnet_amd import ops, synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.train import Trainer, init_distributed
    from oracle import oracle
    ops.set_backend(oracle.backend())
    init_distributed("gloo")
    tr = Trainer(cfg=cfg, device="cpu", world_size=world, local_rank=rank, seed=0)
    tr.net.l3_head.DP1.p = 0.0; tr.net.l4_head.DP1.p = 0.0          # dropout off: ranks must be comparable
    batch = synth.make_batch(1, 4096, 160, 512, seed=10 + rank)
    tr.step(batch)
    grads = tr.named_grads()
    params = {k: p.detach().clone() for k, p in tr.net.named_parameters()}
    torch.save({"grads": grads, "params": params}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _single(seed_rank, out):
    from i2pnet_amd import ops, synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.loss import Get_loss
    from i2pnet_amd.train import Trainer
    tr = Trainer(cfg=cfg, device="cpu", seed=0)
    tr.net.l3_head.DP1.p = 0.0; tr.net.l4_head.DP1.p = 0.0
    tr.model.train()
    b = synth.make_batch(1, 4096, 160, 512, seed=10 + seed_rank)
    o3, o4, _, _, sx, sq = tr.model(b["rgb"], b["lidar"], b["raw_point_xyz"], None, b["init_intrinsic"], None, None, None,
                                    b["lidar_feats"], cfg=cfg)
    loss, _, _ = Get_loss(o3, o4, b["decalib_real_gt"], b["decalib_dual_gt"], sx, sq, cfg=cfg)
    loss.backward()
    out[seed_rank] = {k: p.grad.clone() for k, p in tr.net.named_parameters() if p.grad is not None}


def test_ddp_gloo_two_ranks(tmp_path, oracle_backend):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    for k in r0["params"]:                                    # replicas stay in lock-step
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    for k in r0["grads"]:                                     # both ranks hold the all-reduced gradient
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k
    # ... which is the mean of the two single-process gradients (pre-clip: clip acts after the all-reduce)
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        torch.set_num_threads(4)
        single = {}
        _single(0, single); _single(1, single)
    finally:
        ops.set_backend(prev)
    tot = sum(float((0.5 * (single[0][k] + single[1][k])).double().norm() ** 2) for k in single[0]) ** 0.5
    scale = min(1.0, 10.0 / (tot + 1e-6))                     # Trainer clips the global norm at 10
    # per-parameter check, except the mask up-conv whose fp32 gradient is rounding noise in the
    # reference as well (it consumes -1e10 mask values) — that one only enters the global norm
    num = den = 0.0
    errs = {}
    for k in single[0]:
        want = 0.5 * (single[0][k] + single[1][k]) * scale
        diff = (r0["grads"][k] - want).double()
        num += float(diff.norm() ** 2); den += float(want.double().norm() ** 2)
        errs[k] = float(diff.norm()) / (float(want.double().norm()) + 1e-9)
    # (conv biases in front of a train-mode BN likewise carry pure rounding noise: only parameters
    # holding at least 1e-3 of the global gradient norm are checked individually)
    wn = {k: float((0.5 * (single[0][k] + single[1][k]) * scale).double().norm()) for k in single[0]}
    bad = {k: v for k, v in errs.items() if v > 5e-2 and not k.startswith("set_upconv0_w_upsample")
           and wn[k] > 1e-3 * den ** 0.5}
    assert not bad, bad
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_flat_adam_matches_torch_adam():
    """FlatAdam (device-scalar step/lr, flat buffers) reproduces torch.optim.Adam with L2 weight decay."""
    from i2pnet_amd.train import FlatAdam
    torch.manual_seed(0)
    w = torch.randn(1000)
    ref = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.99)
    flat, g = w.clone(), torch.zeros(1000)
    mine = FlatAdam(flat, g, 1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    for it in range(20):
        grad = torch.randn(1000) * (10.0 ** (it % 3 - 1))
        ref.grad = grad.clone(); opt.step()
        g.copy_(grad); mine.step()
        if it % 5 == 4:
            sched.step(); mine.decay_lr(0.99)
        assert torch.allclose(flat, ref.data, rtol=1e-5, atol=1e-7), it


def test_checkpoint_is_torch_adam_compatible(tmp_path, oracle_backend):
    """Trainer.save_checkpoint writes what the reference trainer loads (train20v2learn_wandb_proj.py:218-221): a
    torch.optim.Adam / ExponentialLR state dict over all parameters of the network; loading it into torch's own optimizer and
    scheduler works, the moments are the flat buffers' slices, and Trainer.load_checkpoint restores weights, moments, step count
    and the decayed learning rate from it."""
    import torch
    from i2pnet_amd import ops, synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.train import Trainer
    prev = ops.set_backend(oracle_backend)
    try:
        tr = Trainer(cfg=cfg, device="cpu", seed=0)
        batch = synth.make_batch(1, 2048, 160, 512, seed=3)
        for _ in range(2):
            tr.step(batch)
        tr.epoch_end()
        tr.save_checkpoint(tmp_path / "ck.pth", epoch=1)
        ck = torch.load(tmp_path / "ck.pth", weights_only=False)
        # the reference's side: torch.optim.Adam over model.parameters() + ExponentialLR
        ref_params = [torch.nn.Parameter(p.detach().clone()) for p in tr.net.parameters()]
        opt = torch.optim.Adam(ref_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0001)
        sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.99)
        opt.load_state_dict(ck["optimizer_state_dict"]); sch.load_state_dict(ck["scheduler_state_dict"])
        assert abs(opt.param_groups[0]["lr"] - 1e-3 * 0.99) < 1e-9 and sch.last_epoch == 1       # (the rate lives in an fp32 device scalar)
        names = [k for k, _ in tr.net.named_parameters()]
        k = names.index("cost_volume1.mlp1_convs.1.conv.weight")
        st = opt.state[ref_params[k]]
        assert float(st["step"]) == 2.0 and float(st["exp_avg"].abs().max()) > 0
        tr2 = Trainer(cfg=cfg, device="cpu", seed=1)
        assert tr2.load_checkpoint(tmp_path / "ck.pth") == 1
        assert torch.equal(tr2.flat_param, tr.flat_param)
        assert torch.equal(tr2.optimizer.exp_avg, tr.optimizer.exp_avg) and torch.equal(tr2.optimizer.exp_avg_sq, tr.optimizer.exp_avg_sq)
        assert float(tr2.optimizer.step_t) == 2.0 and abs(float(tr2.optimizer.lr_t) - 1e-3 * 0.99) < 1e-9
        # a file shaped like the REFERENCE trainer's ckpt.pt (train20v2learn_wandb_proj.py:255-268): `training_params` and the
        # best_* meters, which are numpy float64 scalars — loads under the default weights_only=True
        import numpy as np
        ck["training_params"] = {"dataset": "kitti", "max_t": 10.0, "val_sequence": [0]}
        ck["best_rotation_error"] = np.float64(1.25); ck["best_transition_error"] = np.float64(0.5); ck["best_acc"] = np.float64(0.0)
        torch.save(ck, tmp_path / "ref_like.pth")
        tr3 = Trainer(cfg=cfg, device="cpu", seed=2)
        assert tr3.load_checkpoint(tmp_path / "ref_like.pth") == 1 and torch.equal(tr3.flat_param, tr.flat_param)

        class Evil:
            def __reduce__(self):
                return (print, ("pickle payload ran",))
        ck["extra"] = Evil()
        torch.save(ck, tmp_path / "evil.pth")
        with pytest.raises(RuntimeError, match="trust_pickle"):
            tr3.load_checkpoint(tmp_path / "evil.pth")
    finally:
        ops.set_backend(prev)


def test_flat_adam_gate_skips_a_step():
    """the torch fallback of Trainer._update gates its update by the poison word: gate False leaves everything untouched"""
    import torch
    from i2pnet_amd.train import FlatAdam
    w, g = torch.randn(64), torch.randn(64)
    opt = FlatAdam(w, g, 1e-2)
    opt.step(gate=torch.tensor(True))
    w1, m1, v1, t1 = w.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), float(opt.step_t)
    g_good = g.clone()
    g.fill_(float("nan"))                 # what an abandoned grid barrier may leave behind
    opt.step(gate=torch.tensor(False))
    assert torch.equal(w, w1) and torch.equal(opt.exp_avg, m1) and torch.equal(opt.exp_avg_sq, v1) and float(opt.step_t) == t1 == 1.0
    g.copy_(g_good)
    opt.step(gate=torch.tensor(True))
    assert not torch.equal(w, w1) and float(opt.step_t) == 2.0
    # a gated-on step is the ungated step
    w2, g2 = w1.clone(), g_good.clone()
    ref = FlatAdam(w2, g2, 1e-2)
    ref.exp_avg.copy_(m1); ref.exp_avg_sq.copy_(v1); ref.step_t.fill_(1.0)
    ref.step()
    assert torch.allclose(w, w2, rtol=0, atol=1e-7) and torch.allclose(opt.exp_avg_sq, ref.exp_avg_sq, rtol=1e-6, atol=0)
    # a gated-off FIRST step (step count 0: zero bias corrections) stays finite and is the identity
    w3 = torch.randn(16); w3_0 = w3.clone()
    o3 = FlatAdam(w3, torch.randn(16), 1e-2)
    o3.step(gate=torch.tensor(False))
    assert torch.equal(w3, w3_0) and float(o3.step_t) == 0.0
