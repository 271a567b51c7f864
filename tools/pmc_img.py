"""Workload of the image-block-tail PMC passes (tools/pmc_img.sh): the BN + LeakyReLU + MaxPool3 tail of the image
encoder's 15 blocks (csrc/image_block.hip: bn_stats_v4 / img_pool_fwd / img_bwd_stats / img_bwd_dx) on the conv-output shapes of the
fp32 step at batch 8 (375x1242 image), without the library convolutions in between, plus two calibration kernels with exactly known
traffic (bn_stats_v4: reads the tensor once; bn_act_fwd_v4: reads and writes it once).  4 iterations, the first is warm-up."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from i2pnet_amd import ops  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.modules import _BnActPool  # noqa: E402

B = 8
dev = torch.device("cuda", 0)
hip = ops.hip_backend()
shapes = []          # (H, W, C, pool stride) of every conv output
h, w = 375, 1242
for cin, chans, strides in cfg.rgb_encoder_channels:
    for c, s in zip(chans, strides):
        shapes.append((h, w, c, s))
        h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
print("shapes", shapes)
tens = []
g = torch.Generator(device=dev).manual_seed(0)
for (H, W, C, s) in shapes:
    y = torch.randn(B, C, H, W, device=dev, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    tens.append((y, torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True), torch.zeros(C, device=dev),
                 torch.zeros(C, device=dev), torch.ones(C, device=dev), s))
rows, C = B * 228 * 468, 128
x = torch.randn(rows, C, device=dev)
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
for it in range(4):
    hip.bn_stats(x)
    hip.bn_act_forward(x, gam, bet, 1e-5, 0.1)
    for (y, gamma, beta, bias, rm, rv, s) in tens:
        out = _BnActPool.apply(y, gamma, beta, bias, rm, rv, s, 0.1, 1e-5, 0.1, False)
        out.backward(torch.ones_like(out))
        y.grad = gamma.grad = beta.grad = None
torch.cuda.synchronize()
print("calibration tensor bytes", rows * C * 4)
for (H, W, C, s) in shapes:
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    print(f"shape {H}x{W}x{C} s{s}: y {B * H * W * C * 4} B, pooled {B * Ho * Wo * C * 4} B, arg {B * Ho * Wo * C} B")
