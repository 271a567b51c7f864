"""Workload of the round-3 PMC passes (tools/pmc_step.sh): the kernels of the fp32 step on their real shapes —

* the cost_volume1 pi-stage node forward + backward at batch 8 (bench.Cv1Chain = fused._CvPiTail: every wreg_* instantiation the
  step runs, sm_fwd / sm_bwd, outer_sum, pair_sum);
* the level-1 grouping front end: i2p_sa_l1_group (sa_l1_kernel) and i2p_fused_conv_select_k (fcsk_kernel) on the 8192-point scan
  (7 % occupancy), the centre-aligned cloud (all 3600 queries live) and the 150 000-point scan;
* two calibration kernels with exactly known traffic on a [853632, 128] fp32 tensor: bn_stats_v4 (reads rows*128*4 B, writes
  nothing) and bn_act_fwd_v4 (reads and writes rows*128*4 B).
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from i2pnet_amd import ops, projectpn as P, synth  # noqa: E402

B = 8
dev = torch.device("cuda", 0)
hip = ops.hip_backend()
rows, C = B * 228 * 468, 128
x = torch.randn(rows, C, device=dev)
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
chain = bench.Cv1Chain(B, dev)
clouds = []
for npts, layout, zero_rows in ((8192, "scan", 0), (8192, "centre", 0), (150000, "scan", 30000)):
    cloud = synth.lidar_scan(B, npts, torch.Generator(device=dev).manual_seed(1), dev, layout=layout, zero_rows=zero_rows, beams=64)
    clouds.append(hip.project_seq(cloud, [], 64, 1800, 2.0, -24.8)[0])
idx = P.get_stride_idx_cuda(B, 16, 225, 4, 8, dev)
rhw = torch.arange(135, dtype=torch.int32, device=dev)
sel = torch.zeros(3, B, 3600, 32, 1, dtype=torch.long, device=dev)
mask = torch.zeros(B, 3600, 32, 1, device=dev)
unused = torch.zeros(1, device=dev)
for it in range(4):
    hip.bn_stats(x)                                        # calibration: pure 16 B/lane streaming read
    hip.bn_act_forward(x, gam, bet, 1e-5, 0.1)             # calibration: read + write (plus a second bn_stats)
    ops.begin_step(dev)
    out = chain.forward()
    out.backward(chain.g_out)
    ops.end_step(dev)
    for t in chain.inputs + list(chain.cv.parameters()):
        t.grad = None
    # one density per iteration 1..3 (iteration 0 = warm-up on the scan): the per-kernel averages below are per density
    im = clouds[max(it - 1, 0)]
    hip.sa_l1_group(im, im, 16, 225, 4, 8, 9, 15, 32, 0.75)
    hip.fused_conv_select_k(im, im, idx, rhw, 64, 1800, 3600, 9, 15, 32, 3, 0.75, 1, 1, sel[0], sel[1], sel[2], unused, unused, mask, 64, 1800)
torch.cuda.synchronize()
print("tensor bytes", rows * C * 4)
