import sys, time, torch
sys.path.insert(0, '/root/repo')
from i2pnet_amd.config import I2PNetConfig as cfg
from i2pnet_amd.modules import createCNNs
import torch.nn as nn
def build():
    return nn.Sequential(createCNNs(*cfg.rgb_encoder_channels[0]), createCNNs(*cfg.rgb_encoder_channels[1]), createCNNs(*cfg.rgb_encoder_channels[2])).cuda()
x = torch.rand(8,3,375,1242,device='cuda')*255
def bench(net, inp, iters=10):
    for _ in range(3):
        out = net(inp); out.sum().backward()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(iters):
        out = net(inp); out.sum().backward()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/iters*1e3
net = build()
print("default fwd+bwd ms", bench(net, x))
torch.backends.cudnn.benchmark = True
net2 = build(); print("benchmark=True", bench(net2, x))
net3 = build().to(memory_format=torch.channels_last); xc = x.contiguous(memory_format=torch.channels_last)
print("channels_last", bench(net3, xc))
torch.backends.cudnn.benchmark = False
