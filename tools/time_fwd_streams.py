"""Captured forward pass (train mode, no backward) at configs[1]'s shapes, one stream vs two: does the point-cloud encoder run under
the image encoder when no profiler is attached?   python tools/time_fwd_streams.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops, synth                      # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg      # noqa: E402
from i2pnet_amd.model import RegNet_v2                 # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 8))
batch = synth.make_batch(B, 8192, 160, 512, seed=3, device=dev)
torch.manual_seed(0)
net = RegNet_v2(cfg=cfg).to(dev).train()


def fwd():
    with torch.no_grad():
        return net(batch["rgb"], batch["lidar"], batch["raw_point_xyz"], None, batch["init_intrinsic"], None, None, None,
                   batch["lidar_feats"], cfg=cfg)[0]


def only(which):
    """one encoder alone, on the current stream"""
    with torch.no_grad():
        if which == "image":
            return net._image_branch(batch["rgb"], batch["init_intrinsic"].float())[2]
        with ops.chains_off():
            return net._lidar_branch(batch["lidar"], batch["raw_point_xyz"], batch["lidar_feats"], cfg, B, 8192, dev)[2]


def timed(fn, tag):
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"{tag:42s} {a.elapsed_time(b) / 50 * 1e3:8.1f} us per replay", flush=True)


os.environ["I2P_ONE_STREAM"] = "1"
timed(fwd, "forward, one stream")
timed(lambda: only("image"), "image encoder alone")
timed(lambda: only("lidar"), "point-cloud encoder alone (layer kernels)")
del os.environ["I2P_ONE_STREAM"]
timed(fwd, "forward, two streams")
