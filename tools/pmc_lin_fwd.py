"""SQ counters of the fused layer forward kernel: 3 launches full, then 3 launches with I2P_LIN_ABLATE=14 (MFMA loop only).
Run under rocprofv3 --pmc ... --kernel-trace (tools/pmc_lin_fwd.sh)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from i2pnet_amd import ops
hip = ops.hip_backend()
rows, cin, cout = 8 * 228 * 468, 128, 128
x = torch.randn(rows, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda") / 11
coef = torch.stack([torch.zeros(cin), torch.ones(cin), torch.zeros(cin)]).cuda().contiguous()
for ab in (0, 14):
    os.environ["I2P_LIN_ABLATE"] = str(ab)
    for _ in range(3):
        hip.lin_forward(x, coef, 0.1, w)
    torch.cuda.synchronize()
