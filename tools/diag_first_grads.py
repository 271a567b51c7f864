"""per-parameter gradient difference between the fused first block and MIOpen's convolution there (same process, sized fixture)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tools"):
    sys.path.insert(0, str(p))
import torch
import bench_mode_parity as BMP

tag = sys.argv[1] if len(sys.argv) > 1 else "kitti_b16"
torch.backends.cudnn.benchmark = False
grads = {}
for env in ("1", "0", "0"):
    os.environ["I2P_NO_IMG_FIRST"] = env
    gold, model, rep, knn = BMP.run_pass(tag)
    grads.setdefault(env, []).append({k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None})
    del model
a, b, c = grads["1"][0], grads["0"][0], grads["0"][1]
rows = []
for k in a:
    n = a[k].norm().item()
    rows.append(((a[k] - b[k]).norm().item() / max(n, 1e-30), (b[k] - c[k]).norm().item() / max(n, 1e-30), abs(a[k].norm().item() - b[k].norm().item()) / max(n, 1e-30), n, k))
rows.sort(reverse=True)
print("rel |miopen - fused|   rel |fused - fused(rerun)|   rel norm diff   norm   name")
for r in rows[:25]:
    print(f"{r[0]:.3e}  {r[1]:.3e}  {r[2]:.3e}  {r[3]:.3e}  {r[4]}")
import numpy as np
gn = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm"].tolist()))
g64 = dict(zip(gold["grad_keys"].tolist(), gold["grad_norm64"].tolist()))
print("\nkey: gold fp32 norm, gold fp64 norm, miopen-first, fused, fused rerun (relative to fp64)")
for k in a:
    if k.startswith("l4_head") or k.startswith("l3_head"):
        print(f"{k:55s} {gn[k]:.6e} {g64[k]:.6e}  {(a[k].norm().item()-g64[k])/g64[k]:+.3e} {(b[k].norm().item()-g64[k])/g64[k]:+.3e} {(c[k].norm().item()-g64[k])/g64[k]:+.3e}  |a-b|/|a| {(a[k]-b[k]).norm().item()/a[k].norm().item():.2e}")
