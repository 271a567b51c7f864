"""FPS at the small-range model's sizes: time per launch and per dependent iteration.
    python tools/bench_fps.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd.pointnet2_utils import furthest_point_sample
g = torch.Generator().manual_seed(0)
for B, N, M in [(8, 8192, 2048), (8, 2048, 1024), (8, 1024, 256), (8, 256, 64), (1, 8192, 2048)]:
    xyz = ((torch.rand(B, N, 3, generator=g) - 0.5) * 60).cuda()
    for _ in range(2):
        furthest_point_sample(xyz, M)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        furthest_point_sample(xyz, M)
    e.record(); e.synchronize()
    t = s.elapsed_time(e) / 5 * 1e3
    print(f"B={B} N={N} -> {M}: {t:8.1f} us  {t / (M - 1):6.3f} us/iteration", flush=True)
