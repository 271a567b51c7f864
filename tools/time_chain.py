"""Per-launch time of the one-launch MLP chains (csrc/mlp_chain.hip) on the step's shapes, next to the layer-by-layer kernels,
with the kernel's diagnostic ablation bits (I2P_CHAIN_ABL: 1 no MFMA loop, 2 constant weights (no global loads in the loop),
4 no statistics atomics, 8 no grid barrier, 16 no replica-sum loads, 32 no y store; in the backward: 1 no wgrad, 2 no dgrad,
4 no atomics, 8 no barrier).  20 launches per hipGraph replay.

    python tools/time_chain.py [abl ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from i2pnet_amd import fused, ops  # noqa: E402

DEV = "cuda"
CASES = [(29184, 128, 67, (64, 64, 128), 16), (14848, 132, 131, (128, 128, 256), 16), (14848, 132, 131, (128, 128), 0), (14848, 128, 67, (128, 64, 64), 16),
         (14592, 128, 67, (128, 64), 8), (7296, 12, 10, (64,), 0), (7296, 128, 128, (64,), 0), (1824, 128, 128, (64,), 0),
         (928, 128, 128, (64,), 0)]
N = 20


def make(rows, c0, cin, widths):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, c0, generator=g)).to(DEV)
    params, cp = [], cin
    for c in widths:
        params += [(torch.randn(c, cp, generator=g) / cp ** 0.5).to(DEV), torch.ones(c, device=DEV), torch.zeros(c, device=DEV)]
        cp = c
    return x, params


def timed(fn):
    ops.begin_step(torch.device(DEV, 0))
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(N):
                fn()
    finally:
        ops.end_step(torch.device(DEV, 0))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / (10 * N)


def main():
    abls = [int(a) for a in sys.argv[1:]] or [0]
    be = ops.hip_backend()
    pick = os.environ.get("TIME_CHAIN_CASES")          # comma-separated case indices (PMC passes: tools/pmc_chain.sh)
    cases = [CASES[int(i)] for i in pick.split(",")] if pick else CASES
    for rows, c0, cin, widths, pool_k in cases:
        x, params = make(rows, c0, cin, widths)
        slopes = (1.0,) + (0.1,) * len(widths)
        line = f"{rows:6d} x {c0:3d} -> {'-'.join(map(str, widths)):>10s} k{pool_k:<2d}"
        os.environ["I2P_NO_CHAIN"] = "1"
        with torch.no_grad():
            t = timed(lambda: fused._MlpChain.apply(x, False, slopes, pool_k, None, *params))
        line += f"  layers {t:6.1f} us |"
        os.environ["I2P_NO_CHAIN"] = "0"
        for a in abls:
            os.environ["I2P_CHAIN_ABL"] = str(a)
            with torch.no_grad():
                t = timed(lambda: fused._MlpChain.apply(x, False, slopes, pool_k, None, *params))
            line += f"  abl{a}: {t:6.1f}"
        os.environ.pop("I2P_CHAIN_ABL", None)
        print(line, flush=True)
        # forward + backward of the autograd node: layer by layer / one-launch forward only / one-launch both ways
        xs = x.clone().requires_grad_(True)
        ps = [q.clone().requires_grad_(True) for q in params]
        go = torch.randn(rows // pool_k if pool_k else rows, widths[-1], device=DEV)

        def fb():
            xs.grad = None
            for q in ps:
                q.grad = None
            fused._MlpChain.apply(xs, False, slopes, pool_k, None, *ps).backward(go)
        line = " " * 42 + "fwd+bwd:"
        for tag, nc, cb in (("layers", "1", "0"), ("chain fwd", "0", "0"), ("chain both (where taken)", "0", "1")):
            os.environ["I2P_NO_CHAIN"], os.environ["I2P_CHAIN_BWD"] = nc, cb
            if tag.startswith("chain both"):
                for a in abls:
                    os.environ["I2P_CHAIN_ABL"] = str(a)
                    line += f"  {tag} abl{a} {timed(fb):6.1f}"
                os.environ.pop("I2P_CHAIN_ABL", None)
            else:
                line += f"  {tag} {timed(fb):6.1f} us |"
        os.environ["I2P_NO_CHAIN"] = "0"
        os.environ.pop("I2P_CHAIN_BWD", None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
