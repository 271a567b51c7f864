"""3x3 convolutions of the image encoder (15 layers, channels_last) on MIOpen: fp32 against bf16 storage, forward and
backward (data + weights), per layer and in total.   python tools/bench_img_conv.py [--batch 16]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402


def layers(h=375, w=1242):
    out = []
    for cin, chans, strides in cfg.rgb_encoder_channels:
        for c, s in zip(chans, strides):
            out.append((cin, c, h, w))
            cin = c
            if s == 2:
                h, w = (h + 1) // 2, (w + 1) // 2
    return out


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    tot = {}
    for cin, cout, h, w in layers():
        row = []
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(a.batch, cin, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
            wt = torch.randn(cout, cin, 3, 3, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
            x.requires_grad_(cin != 3)
            wt.requires_grad_(True)
            y = F.conv2d(x, wt, None, 1, 1)
            g = torch.randn_like(y)
            tf = timeit(lambda: F.conv2d(x, wt, None, 1, 1))
            ins = [t for t in (x, wt) if t.requires_grad]
            tb = timeit(lambda: torch.autograd.grad(y, ins, g, retain_graph=True))
            row += [tf, tb]
            tot[dt] = tot.get(dt, 0.0) + tf + tb
        print(f"{cin:4d}->{cout:4d} {h:4d}x{w:4d}  fp32 fwd {row[0]:8.1f} bwd {row[1]:8.1f} | bf16 fwd {row[2]:8.1f} bwd {row[3]:8.1f} us",
              flush=True)
    print({str(k): round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
