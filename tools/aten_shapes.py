"""Device time of the ATen ops (glue) of one eager training step, grouped by op and input shapes — the large elementwise
adds / copies / reductions that are worth fusing show up at the top."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import synth  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
tr = Trainer(cfg=cfg, device=dev)
batch = synth.make_batch(8, 8192, 375, 1242, seed=1, device=dev)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.self_device_time_total > 0 and "convolution" not in e.key:
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total aten self device time (us):", tot)
for t, n, k, s in rows[:45]:
    print(f"{t:9.1f} us  x{n:<3d} {k:22s} {s}")
