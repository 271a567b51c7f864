"""The fp32 and bf16 loss curves of tests/test_bf16_gpu.py::test_bf16_training_tracks_fp32, printed (every 5th step and the
tail means), with switches to take kernel families out:  I2P_NO_WREG=1 (in-register layer kernels off),
--no-gemm-tn (plain linears' wgrad back on rocBLAS), --seed / --steps to see the run-to-run spread of the statement."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import fused, ops, synth  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.train import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--no-gemm-tn", action="store_true")
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--data-seed", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
if a.no_gemm_tn:
    fused.LINEAR_TN_MIN_ROWS = 1 << 60
dev = torch.device("cuda", 0)
batch = synth.make_batch(4, 8192, 160, 512, seed=a.data_seed, device=dev)


def curve(prec):
    prev = ops.set_precision(prec); prev_r = ops.BF16_MIN_ROWS
    ops.BF16_MIN_ROWS = 4096
    try:
        tr = Trainer(cfg=cfg, device=dev, seed=a.seed)
        tr.net.l3_head.DP1.p = 0.0; tr.net.l4_head.DP1.p = 0.0
        return [float(tr.step(batch)[0]) for _ in range(a.steps)]
    finally:
        ops.set_precision(prev); ops.BF16_MIN_ROWS = prev_r


c32, c32b, c16 = curve("fp32"), curve("fp32"), curve("bf16")
f = lambda c: " ".join(f"{v:7.3f}" for v in c[::5] + c[-1:])
print("fp32 :", f(c32)); print("fp32':", f(c32b)); print("bf16 :", f(c16))
t = lambda c: sum(c[-5:]) / 5
print(f"tails fp32 {t(c32):.4f} fp32(second run) {t(c32b):.4f} bf16 {t(c16):.4f}  rel diff {abs(t(c16) - t(c32)) / abs(t(c32)):.4f}")
