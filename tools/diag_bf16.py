"""bf16 storage mode vs fp32 on the golden-vector model: per-activation and per-parameter-gradient deviations.
    python tools/diag_bf16.py [kitti|nus] [min_rows ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import test_model_golden as G
from i2pnet_amd import ops

tag = sys.argv[1] if len(sys.argv) > 1 else "kitti"
thresholds = [int(a) for a in sys.argv[2:]] or [0]


def run(prec, min_rows):
    prev_p, prev_r = ops.set_precision(prec), ops.BF16_MIN_ROWS
    ops.BF16_MIN_ROWS = min_rows
    try:
        torch.manual_seed(0)
        gold, model, acts, out3, out4, loss = G._run(tag, "cuda")
    finally:
        ops.set_precision(prev_p); ops.BF16_MIN_ROWS = prev_r
    grads = {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None}
    a = {k: v.detach().double().clone() for k, v in acts.items()}
    ag = {k: v.grad.detach().double().clone() for k, v in acts.items() if v.grad is not None}
    return gold, a, ag, grads, out3.detach().double(), out4.detach().double(), float(loss)


rel2 = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
relm = lambda x, y: float((x - y).abs().max() / (y.abs().max() + 1e-30))
gold, a0, ag0, g0, o3, o4, l0 = run("fp32", 0)
_, a0b, ag0b, g0b, _, _, _ = run("fp32", 0)
print("fp32 run-to-run: worst pgrad L2", max(rel2(g0b[k], g0[k]) for k in g0))
for thr in thresholds:
    _, a1, ag1, g1, p3, p4, l1 = run("bf16", thr)
    print(f"== min_rows {thr}: loss {l0:.5f} -> {l1:.5f}; out3 max-rel {relm(p3, o3):.4f} out4 {relm(p4, o4):.4f}")
    for k in a0:
        print(f"   act {k:28s} L2 {rel2(a1[k], a0[k]):.4f} max {relm(a1[k], a0[k]):.4f}" + (f"   grad L2 {rel2(ag1[k], ag0[k]):.4f}" if k in ag1 else ""))
    rows = sorted(((rel2(g1[k], g0[k]), relm(g1[k], g0[k]), k) for k in g0), reverse=True)
    tot = (sum(float((g1[k] - g0[k]).norm() ** 2) for k in g0) / sum(float(g0[k].norm() ** 2) for k in g0)) ** 0.5
    print(f"   whole-gradient L2 {tot:.4f}; worst parameter tensors:")
    for r2, rm, k in rows[:12]:
        print(f"     {k:55s} L2 {r2:.4f} max {rm:.4f} |g| {float(g0[k].norm()):.3e}")
