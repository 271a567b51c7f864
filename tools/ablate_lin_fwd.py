import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from i2pnet_amd import ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from bench_kernels import timeit
hip = ops.hip_backend()
rows, cin, cout = 8*228*468, 128, 128
x = torch.randn(rows, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda")/11
coef = torch.stack([torch.zeros(cin), torch.ones(cin), torch.zeros(cin)]).cuda().contiguous()
for ab in [0, 0, 0]:
    os.environ["I2P_LIN_ABLATE"] = str(ab)
    t = timeit(lambda: hip.lin_forward(x, coef, 0.1, w))
    print("ablate", ab, "%.1f us" % t)
