"""Which work on a second stream disturbs the resident-grid chain kernels of the captured training step?

The chain kernels (mlp_chain.hip) synchronise their blocks with a grid barrier in global memory, so every block of a launch has
to be resident at once.  This tool replays the captured step while a second stream runs one kind of work at a time — the pieces
of the loader's device build — and reports the step time and the number of chain launches that abandoned a barrier.

    python tools/chain_cotenancy.py [steps]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from i2pnet_amd import ops, synth, data as D
from i2pnet_amd.config import I2PNetConfig as cfg
from i2pnet_amd.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 8
dev = torch.device("cuda", 0)
tr = Trainer(cfg=cfg, device=dev, capturable=True)
batch = synth.make_batch(B, 150000, 160, 512, seed=1000, device=dev, fup=cfg.fup, fdown=cfg.fdown)     # the loader's shapes
tr.capture(batch)
side = torch.cuda.Stream(dev)
g = torch.Generator().manual_seed(0)
scan_h = (torch.randn(120000, 4, generator=g) * 20).pin_memory()
img_h = torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8).pin_memory()
scan_d, img_d = scan_h.to(dev), img_h.to(dev)
perm_d = torch.randperm(120000, device=dev)
E = np.random.RandomState(0).randn(3, 4)
Ed = torch.as_tensor(E, dtype=torch.float64, device=dev)
host = {"scan": scan_h, "image": img_h, "Tr": np.hstack([np.eye(3), np.zeros((3, 1))]), "K": np.array([[700.0, 0, 600], [0, 700.0, 180], [0, 0, 1]]),
        "P2": np.eye(4), "index": 0, "path_info": "0"}
builder = D.DeviceSampleBuilder(dev, mode="train")

WORK = {
    "nothing": lambda: None,
    "h2d copies (pinned, 3.4 MB x 8)": lambda: [(scan_h.to(dev, non_blocking=True), img_h.to(dev, non_blocking=True)) for _ in range(8)],
    "zero fills (3 x 14 MB)": lambda: [torch.zeros(B, 150000, 3, device=dev) for _ in range(3)],
    "randperm(120000) x 8": lambda: [torch.randperm(120000, device=dev) for _ in range(8)],
    "row gather scan[perm] x 8": lambda: [scan_d[perm_d][:100000] for _ in range(8)],
    "rocBLAS f64 product [N,3]x[3,3] x 8": lambda: [(scan_d[:, :3].double() @ Ed[:, :3].t() + Ed[:, 3]).float() for _ in range(8)],
    "f64 transform, elementwise x 8": lambda: [D.affine_f64(scan_d[:, :3], E) for _ in range(8)],
    "point jitter x 8": lambda: [scan_d[:, :3] + torch.clamp(0.01 * torch.randn_like(scan_d[:, :3]), -0.05, 0.05) for _ in range(8)],
    "image halve + crop x 8": lambda: [D.resize_linear_u8(img_d[50:], 162, 621)[:160, :512].permute(2, 0, 1).float() for _ in range(8)],
    "small pageable copies x 24 (host waits for the side stream)": lambda: [torch.as_tensor(E, dtype=torch.float64, device=dev) for _ in range(24)],
    "whole build of 8 samples": lambda: builder([dict(host) for _ in range(8)]),
}

for name, work in WORK.items():
    ops.chain_errors_reset(dev)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    worst = 0.0
    s.record()
    t_prev = time.perf_counter()
    err = None
    for i in range(steps):
        with torch.cuda.stream(side):
            work()
        try:
            tr.step(batch)
        except ops.ChainBarrierTimeout:
            err = i
            ops.chain_errors_reset(dev)
        now = time.perf_counter(); worst = max(worst, now - t_prev); t_prev = now
    e.record(); torch.cuda.synchronize()
    n = ops.chain_errors(dev)
    print(f"{name:62s} {s.elapsed_time(e) / steps:8.2f} ms/step  worst host gap {1e3 * worst:7.1f} ms  abandoned barriers: {n}"
          + (f" (raised at step {err})" if err is not None else ""), flush=True)
ops.chain_errors_reset(dev)
