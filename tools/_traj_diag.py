"""diagnostic: per-parameter gradient norms of trajectory step 1 on this device vs a saved CPU (oracle backend) evaluation"""
import json, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from helpers import synthetic_state
from i2pnet_amd import ops, synth
from i2pnet_amd.config import CONFIGS
from i2pnet_amd.train import Trainer
gold = np.load(ROOT / "tests/golden/model_kitti_traj.npz")
cfg_name, B, N, img_h, img_w, seed, beams, steps = gold["meta"].tolist()
B, N, img_h, img_w, seed, beams = map(int, (B, N, img_h, img_w, seed, beams))
cfg = CONFIGS[cfg_name]
dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
if dev.type == "cpu":
    from oracle import oracle
    ops.set_backend(oracle.backend())
tr = Trainer(cfg=cfg, device=dev, seed=0)
theirs = {k: tuple(int(x) for x in s.split(",") if x) for k, s in zip(gold["state_keys"].tolist(), gold["state_shapes"].tolist())}
with torch.no_grad():
    tr.net.load_state_dict(synthetic_state(list(theirs.items()), seed=seed))
tr.net.l3_head.DP1.p = 0.0; tr.net.l4_head.DP1.p = 0.0
b = synth.make_batch(B, N, img_h, img_w, seed=seed, beams=beams, fup=cfg.fup, fdown=cfg.fdown, unique_cells=(cfg.init_H, cfg.init_W))
b = tr._to_device({k: v.to(dev) for k, v in b.items()})
out = tr._forward_backward(b)
norms = {k: float(v.double().norm()) for k, v in tr.named_grads().items()}
f = ROOT / "tools/_tmp/traj_cpu_norms.json"
if dev.type == "cpu":
    f.write_text(json.dumps(norms)); print("saved", len(norms), "total", sum(v * v for v in norms.values()) ** 0.5)
else:
    ref = json.loads(f.read_text())
    rows = sorted(((abs(norms[k] ** 2 - ref[k] ** 2), k, norms[k], ref[k]) for k in ref), reverse=True)[:12]
    print("loss", [float(x) for x in out], "total", sum(v * v for v in norms.values()) ** 0.5, "cpu", sum(v * v for v in ref.values()) ** 0.5)
    for d, k, a, r in rows:
        print(f"{k:60s} gpu {a:12.4f} cpu {r:12.4f}")
