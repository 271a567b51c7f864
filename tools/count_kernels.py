"""Kernel launches of one eager training step by kernel name (torch.profiler): `python tools/count_kernels.py`"""
import sys, collections, torch
sys.path.insert(0, "/root/repo")
from torch.profiler import ProfilerActivity, profile
from i2pnet_amd import synth
from i2pnet_amd.config import I2PNetConfig as cfg
from i2pnet_amd.train import Trainer
dev = torch.device("cuda", 0)
import bench
bench.process_setup()                     # the bench's MIOpen mode
tr = Trainer(cfg=cfg, device=dev)
batch = synth.make_batch(8, 8192, 375, 1242, seed=1, device=dev)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.step(batch); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        n = e.name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
        cnt[n.split("(")[0].split("<")[0][:50]] += 1
print("total", sum(cnt.values()))
for k, v in cnt.most_common(45):
    print(v, k)
