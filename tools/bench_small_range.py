"""Training-step throughput of the small-range model on synthetic KITTI-shaped batches
(375x1242 RGB + 8192 points, fp32, forward + loss + backward + clip + Adam, step captured in one hipGraph).
    python tools/bench_small_range.py [--batch 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import synth  # noqa: E402
from i2pnet_amd.small_range import RegNet_v2, SmallRangeConfig  # noqa: E402
from i2pnet_amd.train import Trainer  # noqa: E402


def call(net, b, cfg):
    return net(b["rgb"], b["lidar"], b.get("init_extrinsic"), b["init_intrinsic"], None, None, None, b["lidar_feats"], cfg=cfg,
               lidar_img_raw=b["raw_point_xyz"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--graph", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tr = Trainer(cfg=SmallRangeConfig, device=dev, net_cls=RegNet_v2, call=call, capturable=True)
    batch = synth.make_batch(a.batch, 8192, 375, 1242, seed=1, device=dev)
    live = tr.capture(batch) if a.graph else False
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, _, _ = tr.step(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    if not torch.isfinite(loss).all():
        print("warning: non-finite loss on the synthetic batch (projection_initial divides by z without epsilon, "
              "src/modules/warp_utils.py:153)", flush=True)
    print(f"small-range model, batch {a.batch}, hipgraph {live}: {dt * 1e3:.2f} ms/step  {a.batch / dt:.1f} samples/s", flush=True)


if __name__ == "__main__":
    main()
