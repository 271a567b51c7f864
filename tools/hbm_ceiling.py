"""What a plain streaming kernel reaches on this GPU: device-to-device copy (read + write) and a read-only reduction of tensors far
larger than the 256 MB MALL — the practical ceiling the HBM-bound layer kernels are compared with next to the 8 TB/s nominal peak."""
import torch

dev = torch.device("cuda", 0)
for mb in (512, 2048):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for _ in range(20):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print(f"copy   {mb:5d} MB: {2 * n * 4 / t / 1e12:.2f} TB/s (read + write)")
    for _ in range(10):
        x.sum()
    e0.record()
    for _ in range(50):
        x.sum()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print(f"sum    {mb:5d} MB: {n * 4 / t / 1e12:.2f} TB/s (read only)")
    for _ in range(10):
        y.fill_(1.0)
    e0.record()
    for _ in range(50):
        y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print(f"fill   {mb:5d} MB: {n * 4 / t / 1e12:.2f} TB/s (write only)")
