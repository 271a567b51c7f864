"""i2p_gemm_tn against rocBLAS (a.t() @ b) on the weight-gradient shapes of the plain linear layers (host-timed loops: both
include the launch overhead; the in-graph figures are in profiles/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops
be=ops.hip_backend()
for rows,m,n in [(14848,256,128),(7296,128,256),(3744,3,64),(3744,128,128),(1824,64,192),(928,128,320)]:
    a=torch.randn(rows,m,device='cuda'); b=torch.randn(rows,n,device='cuda')
    for _ in range(50): be.gemm_tn(a,b)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): be.gemm_tn(a,b)
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/200*1e3
    e0.record()
    for _ in range(200): a.t()@b
    e1.record(); torch.cuda.synchronize()
    print(rows,m,n, f"gemm_tn {t:.1f} us  rocblas {e0.elapsed_time(e1)/200*1e3:.1f} us")
