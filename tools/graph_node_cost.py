"""Cost of one small kernel node inside a replayed hipGraph: N dependent launches of a one-element add, replayed 20 times.
Tells how much of the step is per-launch overhead (launches per step x this figure)."""
import time
import torch

dev = torch.device("cuda", 0)
for n_el in (1, 4096, 1 << 20):
    x = torch.zeros(n_el, device=dev)
    for N in (200, 1000):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                x.add_(1.0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(N):
                    x.add_(1.0)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"elements {n_el:8d}  nodes {N:5d}: {dt * 1e6 / N:6.2f} us per node")
