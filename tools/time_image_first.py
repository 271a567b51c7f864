"""Event timing of the image encoder's first block at BASELINE configs[1] size (8 x 3 x 375 x 1242): csrc/image_first.hip (Gram +
coefficients + conv/BN/act/pool forward; sparse pass + finalisation backward) against the path it replaces (channels_last copy +
MIOpen convolution + i2p_img_block_fwd; i2p_img_block_bwd + MIOpen weight gradient).  Prints per-launch-group microseconds."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from i2pnet_amd import ops  # noqa: E402


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 375, 1242)))
    dev = "cuda"
    torch.backends.cudnn.benchmark = True
    be = ops.hip_backend()
    torch.manual_seed(0)
    x = torch.randn(B, 3, H, W, device=dev)
    w = (torch.randn(16, 3, 3, 3, device=dev) * 0.3).contiguous(memory_format=torch.channels_last)
    gam, bet = torch.randn(16, device=dev), torch.randn(16, device=dev) * 0.2
    out, arg, mi, gram = be.img_first_forward(x, w, gam, bet, 1e-5, 0.1, 2)
    g = torch.randn_like(out)
    print(f"size {B}x3x{H}x{W}")
    print(f"first-block forward  (3 launches): {timed(lambda: be.img_first_forward(x, w, gam, bet, 1e-5, 0.1, 2)):8.1f} us")
    print(f"first-block backward (2 launches): {timed(lambda: be.img_first_backward(g, arg, x, w, gam, bet, 0.1, 2, mi, gram)):8.1f} us")

    if os.environ.get("I2P_TIME_FIRST_ONLY") == "1":        # (counter passes: only the new kernels)
        return

    def old_fwd():
        xc = x.contiguous(memory_format=torch.channels_last)
        y = F.conv2d(xc, w, None, 1, 1)
        return xc, y, be.img_block_forward(y.permute(0, 2, 3, 1), gam, bet, 1e-5, 0.1, 2)
    xc, y, (o2, a2, mi2) = old_fwd()
    print(f"replaced forward  (copy + MIOpen conv + stats + pool): {timed(old_fwd):8.1f} us")

    def old_bwd():
        dy, dg, db = be.img_block_backward(g, a2, y.permute(0, 2, 3, 1), mi2, gam, bet, 0.1, 2)
        return torch.ops.aten.convolution_backward(dy.permute(0, 3, 1, 2), xc, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                   (False, True, False))
    old_bwd()
    print(f"replaced backward (stats + dy + MIOpen weight gradient): {timed(old_bwd):8.1f} us")
    # batch statistics against the fp64 evaluation of the convolution (the statistics error is coherent over the whole tensor)
    y64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    m64, v64 = y64.mean((0, 2, 3)), y64.var((0, 2, 3), unbiased=False)
    del y64
    is64 = (v64 + 1e-5).rsqrt()
    for tag, m in (("gram", mi), ("stats of MIOpen y", mi2)):
        print(f"{tag:18s}: max |mean - mean64| / std = {((m[:16].double() - m64).abs() * is64).max().item():.2e}, "
              f"max |invstd / invstd64 - 1| = {(m[16:].double() / is64 - 1).abs().max().item():.2e}")
    d = (out - o2).abs().max().item() / o2.abs().max().item()
    dW = be.img_first_backward(g, arg, x, w, gam, bet, 0.1, 2, mi, gram)[0]
    dW2 = old_bwd()[1]
    print(f"max |out - out_miopen| / max = {d:.2e}; arg equal {(arg == a2).float().mean().item():.6f}; "
          f"max |dW - dW_miopen| / max = {(dW - dW2).abs().max().item() / dW2.abs().max().item():.2e}")


if __name__ == "__main__":
    main()
