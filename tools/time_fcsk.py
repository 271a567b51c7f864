"""Time of i2p_fused_conv_select_k (csrc/fused_conv_select_k.hip) at the level-1 shape of the network (64x1800 image, 3600 queries,
9x15 window, K = 32, batch 8) on the three input densities of bench.py's grouping entry; algorithmic bytes = SURVEY 8(d): the image +
idx_n2 read once, K x 28 B written per query."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from i2pnet_amd import ops, projectpn as P, synth
hip = ops.hip_backend(); dev = torch.device("cuda", 0); B = 8
for name, npts, layout, zr in (("scan", 8192, "scan", 0), ("centre", 8192, "centre", 0), ("150k", 150000, "scan", 30000)):
    cloud = synth.lidar_scan(B, npts, torch.Generator(device=dev).manual_seed(1), dev, layout=layout, zero_rows=zr, beams=64)
    img, _, _ = hip.project_seq(cloud, [], 64, 1800, 2.0, -24.8)
    idx = P.get_stride_idx_cuda(B, 16, 225, 4, 8, dev)
    rhw = torch.arange(135, dtype=torch.int32, device=dev)
    sel = torch.zeros(3, B, 3600, 32, 1, dtype=torch.long, device=dev)
    mask = torch.zeros(B, 3600, 32, 1, device=dev); un = torch.zeros(1, device=dev)
    run = lambda: hip.fused_conv_select_k(img, img, idx, rhw, 64, 1800, 3600, 9, 15, 32, 3, 0.75, 1, 1, sel[0], sel[1], sel[2], un, un, mask, 64, 1800)
    bench._event_time_us(run, 200)
    t = bench._event_time_us(run, 100)
    by = B * (64 * 1800 * 12 + 3600 * 8 + 3600 * 32 * 28)
    print(name, round(t, 1), "us", round(by / t / 1e3 / 8000, 3))
