"""kNN at the model's sizes: time per launch ."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
hip = ops.hip_backend()
g = torch.Generator().manual_seed(0)
for B, N, S, k in [(8, 8192, 2048, 32), (8, 2048, 1024, 16), (8, 468, 228, 32), (8, 256, 256, 4)]:
    xyz = ((torch.rand(B, N, 3, generator=g) - 0.5) * 60).cuda(); q = xyz[:, :S].contiguous() if S <= N else None
    idx = torch.empty(B, S, k, dtype=torch.int32, device="cuda")
    for _ in range(2):
        hip.knn(xyz, q, k, idx)
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(5):
        hip.knn(xyz, q, k, idx)
    e_.record(); e_.synchronize()
    print(f"B={B} N={N} S={S} k={k}: {s_.elapsed_time(e_) / 5 * 1e3:9.1f} us", flush=True)
