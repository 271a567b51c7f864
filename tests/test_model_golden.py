"""The network (i2pnet_amd.model.RegNet_v2) against golden vectors produced by the REFERENCE
Python model (tools/gen_golden.py; reference imported in the build container with the CPU
oracle as its native extension).  CPU variant: our host logic on the oracle backend;
GPU variant: the product path on libi2p_ops.so.  Tolerance: 1e-4 relative (BASELINE.json)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import synthetic_state

GOLD = Path(__file__).resolve().parent / "golden"
GPU_GRAD_TOL = 1e-3          # gradient norms (well-conditioned parameters) on the GPU
GPU_GRAD_TENSOR_TOL = 2e-3   # recorded gradient tensors on the GPU
CASES = ["kitti", "nus"]


def _rel(a, b):
    a = torch.as_tensor(a, dtype=torch.float64); b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _run(tag, device):
    from i2pnet_amd import synth
    from i2pnet_amd.config import CONFIGS
    from i2pnet_amd.loss import Get_loss
    from i2pnet_amd.model import RegNet_v2

    gold = np.load(GOLD / f"model_{tag}.npz")
    cfg_name, B, N, img_h, img_w, seed, beams = gold["meta"].tolist()
    B, N, img_h, img_w, seed, beams = int(B), int(N), int(img_h), int(img_w), int(seed), int(beams)
    cfg = CONFIGS[cfg_name]
    model = RegNet_v2(cfg=cfg)
    # state_dict layout == the reference's (names, order-insensitive, shapes)
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    theirs = {k: tuple(int(x) for x in s.split(",") if x) for k, s in zip(gold["state_keys"].tolist(), gold["state_shapes"].tolist())}
    assert ours == theirs
    model.load_state_dict(synthetic_state(list(theirs.items()), seed=seed))
    model.eval().to(device)
    batch = {k: v.to(device) for k, v in synth.make_batch(B, N, img_h, img_w, seed=seed, beams=beams, fup=cfg.fup,
                                                           fdown=cfg.fdown, unique_cells=(cfg.init_H, cfg.init_W)).items()}
    acts = {}

    def keep(name):
        def f(mod, inp, out):
            o = out[2] if isinstance(out, tuple) else out
            if o.requires_grad:
                o.retain_grad()
            acts[name] = o
        return f

    for name in [k[4:] for k in gold.files if k.startswith("act.")]:
        if name == "LiDAR_lv1":
            orig = model.LiDAR_lv1.forward_center

            def fc(*a, _orig=orig, **k):
                out = _orig(*a, **k)
                out[2].retain_grad(); acts["LiDAR_lv1"] = out[2]
                return out
            model.LiDAR_lv1.forward_center = fc
        else:
            getattr(model, name).register_forward_hook(keep(name))

    out3, out4, _, _, sx, sq = model(batch["rgb"], batch["lidar"], batch["raw_point_xyz"], batch["init_extrinsic"],
                                     batch["init_intrinsic"], None, None, None, batch["lidar_feats"], cfg=cfg)
    loss, lq, lx = Get_loss(out3, out4, batch["decalib_real_gt"], batch["decalib_dual_gt"], sx, sq, cfg=cfg)
    loss.backward()
    return gold, model, acts, out3, out4, loss


def _check(gold, model, acts, out3, out4, loss, tol, grad_tol=None, grad_tensor_tol=1e-3, rgb_tol=None):
    grad_tol = grad_tol or tol
    report = {}
    for name, t in acts.items():
        report["act." + name] = _rel(t.detach().cpu().reshape(-1, t.shape[-1]), gold["act." + name])
    report["out3"] = _rel(out3.detach().cpu(), gold["out3"])
    report["out4"] = _rel(out4.detach().cpu(), gold["out4"])
    report["loss"] = abs(loss.item() - gold["loss"][0]) / abs(gold["loss"][0])
    for name, t in acts.items():
        if "actgrad." + name in gold.files and t.grad is not None:
            report["actgrad." + name] = _rel(t.grad.cpu().reshape(-1, t.shape[-1]), gold["actgrad." + name])
    params = dict(model.named_parameters())
    for k in [f[6:] for f in gold.files if f.startswith("pgrad.")]:
        report["pgrad." + k] = _rel(params[k].grad.cpu(), gold["pgrad." + k])
    # Gradient norms of all parameters.  Several of the reference's gradients are ill-conditioned
    # in fp32 (set_upconv0_w_upsample consumes the -1e10 mask values as features; level 1 consumes
    # absolute coordinates): the fixture carries an fp64 evaluation, and |ref32 - fp64| is the
    # noise floor no fp32 implementation can beat.  We must be as close to fp64 as the reference is
    # (x4 margin), and within 1e-4 wherever the reference itself is.
    gn = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm"].tolist()))
    g64 = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm64"].tolist()))
    worst, worst_key = 0.0, None
    mod_floor = {}                     # fp32 noise floor of the reference per top-level module
    for k in params:
        if gn[k] > 1e-4 and g64[k] > 0.0:
            m = k.split(".")[0]
            mod_floor[m] = max(mod_floor.get(m, 0.0), abs(gn[k] - g64[k]) / g64[k])
    for k, p in params.items():
        g = 0.0 if p.grad is None else float(p.grad.double().norm())
        if gn[k] <= 1e-4 or g64[k] == 0.0:
            # conv biases in front of a batch-stat BN: exactly 0 here and in fp64, rounding noise there
            assert g <= max(1e-3, 4 * gn[k]), (k, g, gn[k])
            continue
        floor = mod_floor[k.split(".")[0]]
        if floor > 0.05:
            continue     # the reference's own fp32 gradient is >5% off its fp64 value here: not a reproducible quantity
        err = abs(g - g64[k]) / g64[k]
        score = err / max(4 * floor, rgb_tol if (rgb_tol and k.startswith("RGB_net")) else grad_tol)
        if score > worst:
            worst, worst_key = score, (k, err, floor)
    report["grad_norm_worst"] = worst
    report["grad_norm_worst_key"] = 0.0 if worst <= 1.0 else str(worst_key)
    limits = {k: tol for k in report}
    limits["grad_norm_worst"] = 1.0                       # already normalised by its own limit
    for k in report:                                       # measured fp32 noise of the reference itself
        if k.startswith("actgrad.") or k.startswith("pgrad."):
            limits[k] = grad_tensor_tol
    limits["pgrad.LiDAR_lv1.mlp_convs.0.conv.weight"] = max(5e-3, grad_tensor_tol)   # |ref32 - fp64| = 1.7e-3
    if rgb_tol:                                                # image-encoder gradients: MIOpen's kernels
        for k in report:
            if k.startswith("pgrad.RGB_net"):
                limits[k] = max(limits[k], rgb_tol)
    bad = {k: v for k, v in report.items() if isinstance(v, str) or not v <= limits[k]}
    assert not bad, f"beyond {tol}: {bad}\nall: {report}"
    return report


@pytest.mark.parametrize("tag", CASES)
def test_model_matches_reference_on_cpu_oracle(tag, oracle_backend):
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        torch.manual_seed(0)
        res = _run(tag, "cpu")
        _check(*res, tol=1e-4)
    finally:
        ops.set_backend(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_model_matches_reference_on_gpu(tag, hip_backend):
    torch.manual_seed(0)
    res = _run(tag, "cuda")
    # Forward tensors: 1e-4.  Gradients: the hand-written backward kernels accumulate in a fixed order (no
    # floating-point atomics: csrc/scatter_det.hip, slab reductions in the pair kernels), so the limits are the CPU
    # variant's (1e-4 on gradient norms where the reference itself is well-conditioned, 1e-3 per recorded tensor);
    # round 1 needed 5e-2 / 2e-2 here because the atomics' rounding noise was amplified by the ill-conditioned parts.
    # The image encoder's convolutions are MIOpen's (north star: it stays on PyTorch-ROCm); its split-K weight-gradient
    # kernels accumulate with atomics, so RGB_net gradients keep a statistical limit (1.5e-2).
    _check(*res, tol=1e-4, grad_tol=GPU_GRAD_TOL, grad_tensor_tol=GPU_GRAD_TENSOR_TOL, rgb_tol=1.5e-2)


@pytest.mark.gpu
def test_gradients_are_bitwise_reproducible(hip_backend):
    """two identical runs give identical gradients (MIOpen's deterministic algorithms selected for the image encoder)"""
    prev = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        def grads():
            torch.manual_seed(0)
            gold, model, acts, out3, out4, loss = _run("kitti", "cuda")
            return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        a, b = grads(), grads()
    finally:
        torch.backends.cudnn.deterministic = prev
    ours = [k for k in a if not k.startswith("RGB_net")]
    diff = [k for k in ours if not torch.equal(a[k], b[k])]
    assert not diff, diff[:8]
    # (image-encoder weights: MIOpen's kernels; reported, not asserted)
    rgb = [k for k in a if k.startswith("RGB_net") and not torch.equal(a[k], b[k])]
    if rgb:
        print("MIOpen gradients differ run to run:", rgb[:4])


def _run_iter(device):
    """iterative fine registration against the reference's modellearn_proj_center_iter outputs"""
    from i2pnet_amd import synth
    from i2pnet_amd.config import CONFIGS
    from i2pnet_amd.model import RegNet_v2_iter

    gold = np.load(GOLD / "model_kitti_iter.npz")
    cfg_name, B, N, img_h, img_w, seed, beams = gold["meta"].tolist()
    B, N, img_h, img_w, seed, beams = int(B), int(N), int(img_h), int(img_w), int(seed), int(beams)
    cfg = CONFIGS[cfg_name]
    model = RegNet_v2_iter(cfg=cfg)
    assert sorted(model.state_dict().keys()) == sorted(gold["state_keys"].tolist())
    model.load_state_dict(synthetic_state([(k, tuple(v.shape)) for k, v in model.state_dict().items()], seed=seed))
    model.eval().to(device)
    batch = {k: v.to(device) for k, v in synth.make_batch(B, N, img_h, img_w, seed=seed, beams=beams, fup=cfg.fup,
                                                           fdown=cfg.fdown, unique_cells=(cfg.init_H, cfg.init_W)).items()}
    with torch.no_grad():
        out3, out4 = model(batch["rgb"], batch["lidar"], batch["raw_point_xyz"], batch["init_extrinsic"],
                           batch["init_intrinsic"], None, None, None, batch["lidar_feats"], cfg=cfg)[:2]
    return gold, out3.cpu(), out4.cpu()


def test_iterative_model_vs_reference_cpu_oracle(oracle_backend):
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        gold, out3, out4 = _run_iter("cpu")
    finally:
        ops.set_backend(prev)
    # six chained fine steps: each step's 1e-5-level differences feed the next warp
    assert _rel(out4, gold["out4"]) < 1e-4
    assert _rel(out3, gold["out3"]) < 5e-4


@pytest.mark.gpu
def test_iterative_model_vs_reference_gpu():
    gold, out3, out4 = _run_iter("cuda")
    assert _rel(out4, gold["out4"]) < 1e-4
    assert _rel(out3, gold["out3"]) < 5e-4


def _run_train_mode(device):
    """one TRAIN-mode step of the main model against the reference (fixture: tools/gen_golden.py train — the reference
    module in .train(), dropout probability 0): image-encoder BatchNorm2d layers on batch statistics + running-buffer
    update through the fused image-block kernels, end to end"""
    from i2pnet_amd import synth
    from i2pnet_amd.config import CONFIGS
    from i2pnet_amd.loss import Get_loss
    from i2pnet_amd.model import RegNet_v2

    gold = np.load(GOLD / "model_kitti_train.npz")
    cfg_name, B, N, img_h, img_w, seed, beams = gold["meta"].tolist()
    B, N, img_h, img_w, seed, beams = int(B), int(N), int(img_h), int(img_w), int(seed), int(beams)
    cfg = CONFIGS[cfg_name]
    model = RegNet_v2(cfg=cfg)
    theirs = {k: tuple(int(x) for x in s.split(",") if x) for k, s in zip(gold["state_keys"].tolist(), gold["state_shapes"].tolist())}
    model.load_state_dict(synthetic_state(list(theirs.items()), seed=seed))
    model.train().to(device)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    batch = {k: v.to(device) for k, v in synth.make_batch(B, N, img_h, img_w, seed=seed, beams=beams, fup=cfg.fup,
                                                           fdown=cfg.fdown, unique_cells=(cfg.init_H, cfg.init_W)).items()}
    out3, out4, _, _, sx, sq = model(batch["rgb"], batch["lidar"], batch["raw_point_xyz"], batch["init_extrinsic"],
                                     batch["init_intrinsic"], None, None, None, batch["lidar_feats"], cfg=cfg)
    loss, _, _ = Get_loss(out3, out4, batch["decalib_real_gt"], batch["decalib_dual_gt"], sx, sq, cfg=cfg)
    loss.backward()
    return gold, model, out3, out4, loss


def _check_train_mode(gold, model, out3, out4, loss, tol, grad_tol):
    assert _rel(out3.detach().cpu(), gold["out3"]) <= tol and _rel(out4.detach().cpu(), gold["out4"]) <= tol
    assert abs(loss.item() - gold["loss"][0]) <= tol * abs(gold["loss"][0])
    # running buffers after the step (sum and absolute sum of every buffer; num_batches_tracked exactly)
    state = model.state_dict()
    for k, s, a in zip(gold["buf_keys"].tolist(), gold["buf_sum"].tolist(), gold["buf_abs_sum"].tolist()):
        v = state[k].double()
        assert abs(float(v.sum()) - s) <= 1e-4 * max(a, 1e-6) and abs(float(v.abs().sum()) - a) <= 1e-4 * max(a, 1e-6), k
    # gradients: global norm, and the norm per top-level module where the reference's own value is not at rounding level
    gn = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm"].tolist()))
    mine, ref = {}, {}
    for k, p in model.named_parameters():
        m = k.split(".")[0]
        g = 0.0 if p.grad is None else float(p.grad.double().norm())
        mine[m] = mine.get(m, 0.0) + g * g; ref[m] = ref.get(m, 0.0) + gn[k] ** 2
    tot_m, tot_r = sum(mine.values()) ** 0.5, sum(ref.values()) ** 0.5
    assert abs(tot_m - tot_r) <= grad_tol * tot_r, (tot_m, tot_r)
    # (the mask up-convolution's gradient is 35 % away from fp64 in the reference itself, DESIGN.md §2: not compared)
    bad = {m: (mine[m] ** 0.5, ref[m] ** 0.5) for m in ref
           if m != "set_upconv0_w_upsample" and ref[m] ** 0.5 > 1e-3 * tot_r and abs(mine[m] ** 0.5 - ref[m] ** 0.5) > 20 * grad_tol * ref[m] ** 0.5}
    assert not bad, bad


def test_train_mode_step_matches_reference_on_cpu_oracle(oracle_backend):
    from i2pnet_amd import ops
    prev = ops.set_backend(oracle_backend)
    try:
        torch.manual_seed(0)
        _check_train_mode(*_run_train_mode("cpu"), tol=1e-4, grad_tol=1e-3)
    finally:
        ops.set_backend(prev)


@pytest.mark.gpu
def test_train_mode_step_matches_reference_on_gpu(hip_backend):
    torch.manual_seed(0)
    _check_train_mode(*_run_train_mode("cuda"), tol=2e-4, grad_tol=2e-3)
