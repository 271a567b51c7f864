"""gather_rows_grad at the shapes of one training step: deterministic owner-scan vs the atomic kernel (I2P_ATOMIC_SCATTER=1)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from i2pnet_amd import ops
hip = ops.hip_backend()
dev = "cuda:0"
B = 8
for name, Q, HW, C in [("L2", 904 * 16, 3600, 32), ("L3", 228 * 16, 904, 64), ("L4", 116 * 16, 228, 128), ("cv2 knn", 228 * 32, 468, 128),
                       ("pc xyz", 228 * 4, 228, 3), ("upconv", 228 * 8, 116, 64), ("small-range", 2048 * 32, 8192, 64)]:
    g = torch.Generator(device=dev).manual_seed(0)
    go = torch.randn(B, Q, C, generator=g, device=dev)
    w = torch.randint(0, HW, (B, Q), generator=g, device=dev)
    if len(sys.argv) > 1 and sys.argv[1] == "hot":          # sparse scan: 93 % of the rows repeat cell 0 in long runs
        hot = (torch.arange(Q, device=dev) // 64) % 14 != 0
        w = torch.where(hot.unsqueeze(0), torch.zeros_like(w), w)
    h = torch.zeros_like(w)
    out = torch.zeros(B, HW, C, device=dev)
    for _ in range(3):
        hip.gather_rows_grad(go, h, w, HW, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.gather_rows_grad(go, h, w, HW, out)
    e1.record(); e1.synchronize()
    print(f"{name:12s} Q={Q:6d} HW={HW:5d} C={C:3d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
