"""Image encoder (3 x createCNNs, channels_last, fused BN/act/pool tail) forward+backward time with MIOpen's
default find mode vs torch.backends.cudnn.benchmark = True (exhaustive find)."""
import sys, time, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from i2pnet_amd.config import I2PNetConfig as cfg
from i2pnet_amd.modules import createCNNs
import torch.nn as nn
def build():
    return nn.Sequential(createCNNs(*cfg.rgb_encoder_channels[0]), createCNNs(*cfg.rgb_encoder_channels[1]),
                         createCNNs(*cfg.rgb_encoder_channels[2])).cuda().to(memory_format=torch.channels_last)
x = (torch.rand(8, 3, 375, 1242, device='cuda') * 255).contiguous(memory_format=torch.channels_last)
def bench(net, inp, iters=10):
    for _ in range(4):
        out = net(inp); out.sum().backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters):
        out = net(inp); out.sum().backward()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / iters * 1e3
torch.manual_seed(0)
print("default            fwd+bwd ms", round(bench(build(), x), 3), flush=True)
torch.backends.cudnn.benchmark = True
print("cudnn.benchmark    fwd+bwd ms", round(bench(build(), x), 3), flush=True)
