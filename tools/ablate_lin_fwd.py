"""Ablation of the fused layer forward kernel (gen-2) on the cost-volume shape: I2P_LIN_ABLATE bits
1 = no MFMA loop, 2 = no stores, 4 = no statistics, 8 = no global loads / LDS staging after the first strip."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from i2pnet_amd import ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from bench_kernels import timeit
hip = ops.hip_backend()
for rows, cin, cout in [(8 * 228 * 468, 128, 128), (8 * 228 * 468, 64, 64), (8 * 3600 * 32, 16, 32)]:
    x = torch.randn(rows, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda") / 11
    coef = torch.stack([torch.zeros(cin), torch.ones(cin), torch.zeros(cin)]).cuda().contiguous()
    for ab, what in [(0, "full"), (1, "no mfma"), (2, "no stores"), (4, "no stats"), (8, "no loads/staging"), (14, "mfma only"),
                     (9, "epilogue+stores only"), (15, "skeleton"), (0, "full")]:
        os.environ["I2P_LIN_ABLATE"] = str(ab)
        t = timeit(lambda: hip.lin_forward(x, coef, 0.1, w))
        print(f"{rows}x{cin}->{cout}  ablate {ab:2d} {what:22s} {t:8.1f} us", flush=True)
os.environ["I2P_LIN_ABLATE"] = "0"
