import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torch.profiler import profile, ProfilerActivity
from i2pnet_amd.modules import createCNNs
import torch.nn.functional as F
net = createCNNs(3, [16, 16, 16, 16, 32], [2, 1, 1, 1, 2]).cuda().to(memory_format=torch.channels_last)
x = torch.randn(8, 3, 375, 1242, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
for _ in range(3):
    net(x).sum().backward()
conv = net[0]
y = F.conv2d(x, conv.weight, None, 1, 1)
print("y strides", y.shape, y.stride(), y.is_contiguous(memory_format=torch.channels_last), conv.weight.stride())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    out = net(x); out.sum().backward(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
