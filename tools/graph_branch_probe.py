"""How does a replayed two-branch hipGraph overlap its branches?  Branch A (side stream) and branch B (capturing stream) are chains of
one-block spin kernels (torch.cuda._sleep); replay time against the longer chain alone."""
import sys
import torch
dev = torch.device("cuda", 0)
CYC = int(20e-6 * 2.4e9 / 1.0)      # ~20 us at 2.4 GHz (calibrated below)


def chain(n):
    for _ in range(n):
        torch.cuda._sleep(CYC)


def build(na, nb, pre=3, post=3, order="AB"):
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        chain(pre)
        side.wait_stream(main)
        for c in order:
            if c == "A":
                with torch.cuda.stream(side):
                    chain(na)
            else:
                chain(nb)
        main.wait_stream(side)
        chain(post)
    return g


def t(g, n=20):
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


one = t(build(0, 100, 0, 0, "B")) / 100
print(f"one spin kernel in a chain: {one:.1f} us")
for na, nb in [(10, 10), (50, 50), (100, 100), (200, 200), (100, 10), (10, 100), (60, 20), (20, 60)]:
    for order in ("AB", "BA"):
        x = t(build(na, nb, order=order))
        print(f"A={na:4d} B={nb:4d} captured {order}: {x:8.1f} us   serial {(na + nb + 6) * one:8.1f}   ideal {(max(na, nb) + 6) * one:8.1f}", flush=True)
