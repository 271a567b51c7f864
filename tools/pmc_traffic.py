"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`,
see tools/pmc_traffic.sh): the roofline kernel of bench.py (fused 128->128 layer forward on the
cost-volume shape) next to two calibration kernels with exactly known traffic on the same tensor:
`bn_stats_v4` (reads rows*128*4 B, writes nothing) and `bn_act_fwd_v4` (reads and writes rows*128*4 B)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from i2pnet_amd import ops  # noqa: E402

B, N, M, C = 8, 228, 468, 128
rows = B * N * M
dev = "cuda"
hip = ops.hip_backend()
x = torch.randn(rows, C, device=dev); w = torch.randn(C, C, device=dev) / C ** 0.5
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
for _ in range(5):
    sx = hip.bn_stats(x)                                   # calibration: pure 16 B/lane streaming read
    coef, mi = hip.bn_finalize(rows, sx, gam, bet, 1e-5)
    out, _ = hip.bn_act_forward(x, gam, bet, 1e-5, 0.1)    # calibration: read + write (plus a second bn_stats)
    y, sy = hip.lin_forward(x, coef, 0.1, w)               # the roofline kernel
    f = torch.randn(B, N, C, device=dev); g = torch.randn(B, M, C, device=dev)
    bn = torch.randn(B, N, C, device=dev); bk = torch.randn(B, M, C, device=dev)
    yp, sp = hip.pair_lin_forward(f, g, bn, bk, w)
    oc, om = hip.bn_finalize(rows, sy, gam, bet, 1e-5)
    ods = torch.zeros(ops.BN_REPLICAS * 2 * C, dtype=torch.float64, device=dev)
    gz = torch.randn(rows, C, device=dev)                  # (its own tensor: dgrad reads gz, y and x = 3 x 437 MB)
    hip.lin_backward(gz, y, oc, om, ods, x, coef, mi, 0.1, w)
    del gz
torch.cuda.synchronize()
print("tensor bytes", rows * C * 4)
