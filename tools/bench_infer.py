"""Inference throughput (eval forward captured in one hipGraph) of the single-pass and the iterative
(6 fine steps) registration network on the BASELINE configs[1] shapes.
    python tools/bench_infer.py [--batch 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import synth  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.model import RegNet_v2, RegNet_v2_iter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    batch = synth.make_batch(a.batch, 8192, 375, 1242, seed=1, device=dev)
    for name, cls in (("single pass", RegNet_v2), ("iterative x6", RegNet_v2_iter)):
        torch.manual_seed(0)
        net = cls(cfg=cfg).to(dev).eval()
        fwd = lambda: net(batch["rgb"], batch["lidar"], batch["raw_point_xyz"], None, batch["init_intrinsic"], None, None, None,
                          batch["lidar_feats"], cfg=cfg)[:2]
        with torch.no_grad():
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd()
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fwd()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.iters):
                g.replay()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
        assert torch.isfinite(out[0]).all()
        print(f"{name:14s} batch {a.batch}: {dt * 1e3:7.3f} ms / forward  {a.batch / dt:8.1f} samples/s", flush=True)


if __name__ == "__main__":
    main()
