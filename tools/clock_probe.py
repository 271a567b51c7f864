"""Sample the engine clock / power (rocm-smi) while one layer kernel runs in a loop (is the fp32 MFMA stream clock- or power-limited?)."""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import ops

hip = ops.hip_backend(); dev = "cuda"
rows, cin, cout = 853632, 128, 128
x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
coef = torch.stack([torch.zeros(cin), torch.ones(cin), torch.zeros(cin)]).to(dev).contiguous()
use_bn = os.environ.get("BN", "1") == "1"
stop = False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l]
        print(" | ".join(keep), flush=True)
        time.sleep(0.3)
t = threading.Thread(target=sampler); t.start()
t0 = time.time()
while time.time() - t0 < 4.0:
    for _ in range(200): hip.lin_forward(x, coef if use_bn else None, 0.1, w)
    torch.cuda.synchronize()
stop = True; t.join()
