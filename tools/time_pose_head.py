"""time the one-block pose-head kernels at the network's shape (B = 8, C = 64, H = 256)"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from i2pnet_amd import modules
dev = torch.device("cuda", 0)
B, C, H = 8, 64, 256
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=dev)
pooled, w1, b1, wq, bq, wt, bt = r(B, C), r(H, C), r(H), r(4, H), r(4), r(3, H), r(3)
mask = (torch.rand(B, H, generator=g, device=dev) > 0.5).float() * 2
ins = [t.requires_grad_(True) for t in (pooled, w1, b1, wq, bq, wt, bt)]
def fwd():
    return modules._PoseHeadMlp.apply(*ins, mask)
t_f = bench._event_time_us(lambda: fwd(), 200)
q, t = fwd()
gq, gt = r(B, 4), r(B, 3)
def both():
    q, t = fwd()
    torch.autograd.grad([q, t], ins, [gq, gt])
t_b = bench._event_time_us(both, 200)
print("pose head forward", round(t_f, 1), "us; forward + backward", round(t_b, 1), "us (eager, host-bound below ~20 us)")
# ---- backward-validation feature (i2p_max_response_fwd/bwd) at the network's shape ------------------------------------------
B, N, M, C = 8, 228, 468, 128
pts, pix = r(B, N, C).requires_grad_(True), r(B, M, C).requires_grad_(True)
valid = (torch.rand(B, N, 1, generator=g, device=dev) > 0.1).float()
go = r(B, M, C)
def mr():
    out = modules.max_response(pts, pix, valid)
    torch.autograd.grad(out, (pts, pix), go)
print("max_response forward + backward", round(bench._event_time_us(mr, 200), 1), "us (eager)")
