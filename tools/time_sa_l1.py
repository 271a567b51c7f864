import torch, sys
sys.path.insert(0, ".")
import bench
from i2pnet_amd import ops, synth
hip = ops.hip_backend(); dev = torch.device("cuda", 0); B = 8
for name, npts, layout, zr in (("scan", 8192, "scan", 0), ("centre", 8192, "centre", 0), ("150k", 150000, "scan", 30000)):
    cloud = synth.lidar_scan(B, npts, torch.Generator(device=dev).manual_seed(1), dev, layout=layout, zero_rows=zr, beams=64)
    im, _, _ = hip.project_seq(cloud, [], 64, 1800, 2.0, -24.8)
    bench._event_time_us(lambda: hip.sa_l1_group(im, im, 16, 225, 4, 8, 9, 15, 32, 0.75), 200)
    t = bench._event_time_us(lambda: hip.sa_l1_group(im, im, 16, 225, 4, 8, 9, 15, 32, 0.75), 100)
    by = B * (2 * 64 * 1800 * 12 + 3600 * 32 * 48)
    print(name, round(t, 1), "us", round(by / t / 1e3 / 8000, 3))
