"""Workload for the HBM-traffic PMC passes of the bf16 roofline kernel (tools/pmc_traffic.sh bf16): the fused 128->128
cost-volume layer forward on bf16 storage (rg_fwd_kernel<4,true,false>) next to two calibration kernels with exactly
known traffic on the same bf16 tensor: `bwd_stats_bf16_kernel` (reads 2 x rows*128*2 B, writes nothing) and
`bn_act_fwd_bf16_kernel` (reads rows*128*2 B, writes rows*128*4 B), plus the layer's dgrad / wgrad and the pair kernels."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from i2pnet_amd import ops  # noqa: E402

B, N, M, C = 8, 228, 468, 128
rows = B * N * M
dev = "cuda"
bf = torch.bfloat16
hip = ops.hip_backend()
x0 = torch.randn(rows, C, device=dev).to(bf); w = torch.randn(C, C, device=dev) / C ** 0.5
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
for _ in range(5):
    x, s0 = hip.lin_forward(x0, None, 1.0, w, out_dtype=bf)
    coef, mi = hip.bn_finalize(rows, s0, gam, bet, 1e-5)
    y, sy = hip.lin_forward(x, coef, 0.1, w, out_dtype=bf)             # the roofline kernel (BN + activation on load)
    oc, om = hip.bn_finalize(rows, sy, gam, bet, 1e-5)
    out = hip.bn_act_apply_bf16(y, oc, 0.1)                            # calibration: read bf16, write fp32
    gz = (torch.randn(rows, C, device=dev) * 0.1).to(bf)
    ods = hip.bn_act_backward_stats_bf16(gz, y, oc, om, 1.0)           # calibration: two bf16 streaming reads
    hip.lin_backward(gz, y, oc, om, ods, x, coef, mi, 0.1, w)
    f = torch.randn(B, N, C, device=dev); g = torch.randn(B, M, C, device=dev)
    bn = torch.randn(B, N, C, device=dev); bk = torch.randn(B, M, C, device=dev)
    yp, sp = hip.pair_lin_forward(f, g, bn, bk, w, out_dtype=bf)
    pc, pm = hip.bn_finalize(rows, sp, gam, bet, 1e-5)
    pds = hip.bn_act_backward_stats_bf16(gz, yp, pc, pm, 1.0)
    hip.pair_lin_backward(gz, f, g, w, y=yp, out_coef=pc, out_mi=pm, out_dsums=pds)
    del out, gz, yp
torch.cuda.synchronize()
print("bf16 tensor bytes", rows * C * 2)
