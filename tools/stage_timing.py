"""Forward/backward time per top-level stage of the network on the GPU (torch events around
module calls; backward via full-backward hooks).  Diagnostic only.

    python tools/stage_timing.py [--batch 8] [--iters 5]
"""
import argparse
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from i2pnet_amd import synth  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.train import Trainer  # noqa: E402

STAGES = ["RGB_net1", "RGB_net2", "RGB_net3", "LiDAR_lv1", "LiDAR_lv2", "LiDAR_lv3", "LiDAR_lv4", "cost_volume1",
          "layer_idx", "flow_predictor0", "l4_head", "set_upconv0_w_upsample", "set_upconv0_upsample", "cost_volume2",
          "flow_predictor0_predict", "flow_predictor0_w", "l3_head"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tr = Trainer(cfg=cfg, device=dev)
    batch = synth.make_batch(a.batch, 8192, 375, 1242, seed=1, device=dev)
    ev = defaultdict(list)

    def wrap(name, mod):
        for fn_name in (["forward_center"] if name == "LiDAR_lv1" else ["forward"]):
            orig = getattr(mod, fn_name)

            def timed(*args, _orig=orig, _name=name, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); out = _orig(*args, **kw); e.record()
                ev["fwd." + _name].append((s, e))
                return out
            setattr(mod, fn_name, timed)

    for n in STAGES:
        wrap(n, getattr(tr.net, n))
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize()
    ev.clear()
    tot = []
    for _ in range(a.iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); tr.step(batch); e.record(); tot.append((s, e))
    torch.cuda.synchronize()
    step_ms = sum(s.elapsed_time(e) for s, e in tot) / len(tot)
    fwd_total = 0.0
    print(f"step {step_ms:.2f} ms (batch {a.batch})")
    for n in STAGES:
        ms = sum(s.elapsed_time(e) for s, e in ev["fwd." + n]) / a.iters
        fwd_total += ms
        print(f"  fwd {n:28s} {ms:8.3f} ms")
    print(f"  fwd sum of stages            {fwd_total:8.3f} ms ; rest (projection, glue, loss, backward, optimiser) {step_ms - fwd_total:8.3f} ms")


if __name__ == "__main__":
    main()
