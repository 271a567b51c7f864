"""PMC workload: the weights-in-registers layer forward alone (run under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from i2pnet_amd import ops
rows, C = 8 * 228 * 468, 128
hip = ops.hip_backend(); dev = "cuda"
x = torch.randn(rows, C, device=dev); w = torch.randn(C, C, device=dev) / C ** 0.5
coef = torch.stack([torch.zeros(C), torch.ones(C), torch.zeros(C)]).to(dev).contiguous()
for _ in range(5):
    hip.lin_forward(x, coef, 0.1, w)
torch.cuda.synchronize()
