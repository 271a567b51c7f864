"""Run-to-run reproducibility of the GPU gradients (diagnostic)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
import test_model_golden as T
from i2pnet_amd import modules

def grads(tag, fused):
    modules.USE_FUSED_MLP = fused
    torch.manual_seed(0)
    gold, model, acts, out3, out4, loss = T._run(tag, "cuda")
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}, {k: v.grad.detach().clone() for k, v in acts.items() if v.grad is not None}

for tag in ["kitti", "nus"]:
    for fused in [True, False]:
        runs = [grads(tag, fused) for _ in range(4)]
        worst = {}
        for r in runs[1:]:
            for k in runs[0][0]:
                d = float((r[0][k] - runs[0][0][k]).abs().max() / (runs[0][0][k].abs().max() + 1e-12))
                worst[k] = max(worst.get(k, 0), d)
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        aw = {}
        for r in runs[1:]:
            for k in runs[0][1]:
                d = float((r[1][k] - runs[0][1][k]).abs().max() / (runs[0][1][k].abs().max() + 1e-12))
                aw[k] = max(aw.get(k, 0), d)
        print(tag, "fused" if fused else "unfused", "param-grad run-to-run:", [(k, "%.1e" % v) for k, v in top], "actgrad:", sorted([(k, "%.1e" % v) for k, v in aw.items()], key=lambda kv: -float(kv[1]))[:3])
