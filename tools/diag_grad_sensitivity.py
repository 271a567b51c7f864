"""How far the sized fixture's gradient norms move under a last-bit perturbation of the INPUT IMAGE (x * (1 + eps * n), n ~ N(0,1),
eps = 6e-8 = half an fp32 ulp), the encoder's first block on MIOpen's convolution (I2P_NO_IMG_FIRST=1) in every pass: the fp32
conditioning of each checked tensor, against which a 1e-3 gradient-norm tolerance has to be read.  Prints, per parameter tensor, the
relative norm error against the fixture's fp64 value for the unperturbed pass and the spread over the perturbed passes.
usage: diag_grad_sensitivity.py [kitti_b16] [passes]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tools"):
    sys.path.insert(0, str(p))
import torch  # noqa: E402

import test_model_sized as T  # noqa: E402
from i2pnet_amd.model import RegNet_v2  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "kitti_b16"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
first = os.environ.get("I2P_NO_IMG_FIRST", "1")
os.environ["I2P_NO_IMG_FIRST"] = first
torch.backends.cudnn.benchmark = False
orig_init = RegNet_v2.__init__
norms, outs = [], []
for k in range(passes + 1):
    def init(self, *a, _k=k, **kw):
        orig_init(self, *a, **kw)
        if _k:
            g = torch.Generator(device="cuda").manual_seed(100 + _k)
            self.RGB_net1.register_forward_pre_hook(
                lambda mod, inp: (inp[0] * (1.0 + 6e-8 * torch.randn(inp[0].shape, device=inp[0].device, generator=g)),))
    RegNet_v2.__init__ = init
    try:
        torch.manual_seed(0)
        gold, model, acts, out3, out4, loss = T._run_sized(tag, "cuda")
    finally:
        RegNet_v2.__init__ = orig_init
    norms.append({n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None})
    outs.append((T._rel(out3.detach().cpu(), gold["out3"]), T._rel(out4.detach().cpu(), gold["out4"])))
    del model
    torch.cuda.empty_cache()
g64 = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm64"].tolist()))
print(f"# {tag}, first block on {'MIOpen' if first == '1' else 'image_first.hip'}; out3/out4 error vs fixture per pass: "
      + ", ".join(f"{a:.1e}/{b:.1e}" for a, b in outs))
rows = []
for n, v in norms[0].items():
    if g64.get(n, 0.0) <= 0.0 or n.startswith("RGB_net"):
        continue
    errs = [(d[n] - g64[n]) / g64[n] for d in norms]
    rows.append((max(errs) - min(errs), errs, n))
rows.sort(reverse=True)
print("spread of the relative norm error over the passes | unperturbed | perturbed passes | tensor")
for sp, errs, n in rows[:16]:
    print(f"{sp:.2e} | {errs[0]:+.2e} | " + " ".join(f"{e:+.2e}" for e in errs[1:]) + f" | {n}")
