"""Run-to-run reproducibility of the GPU gradients: the hand-written backward kernels accumulate in a fixed order
(csrc/scatter_det.hip, the slab reductions of the pair kernels), so with MIOpen's deterministic convolution
algorithms selected two identical runs must give bit-identical gradients."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
import test_model_golden as T

torch.backends.cudnn.deterministic = True


def grads(tag):
    torch.manual_seed(0)
    gold, model, acts, out3, out4, loss = T._run(tag, "cuda")
    return ({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
            {k: v.grad.detach().clone() for k, v in acts.items() if v.grad is not None})


for tag in ["kitti", "nus"]:
    runs = [grads(tag) for _ in range(4)]
    bad = []
    for r in runs[1:]:
        for which in (0, 1):
            for k in runs[0][which]:
                if not torch.equal(r[which][k], runs[0][which][k]):
                    d = float((r[which][k] - runs[0][which][k]).abs().max() / (runs[0][which][k].abs().max() + 1e-12))
                    bad.append((k, "%.1e" % d))
    print(tag, "bit-identical over 4 runs" if not bad else f"NOT reproducible: {sorted(set(bad))[:12]}")
