"""Event timing of csrc/image_conv16.hip at the encoder's block 2-4 size (8 x 188 x 621 x 16, BASELINE configs[1]) against MIOpen's
convolution (cudnn.benchmark find) + the statistics pass it saves."""
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
    B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 188, 621)))
    cout = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    torch.backends.cudnn.benchmark = True
    be = ops.hip_backend()
    torch.manual_seed(0)
    x = torch.randn(B, H, W, 16, device="cuda")
    w = (torch.randn(cout, 16, 3, 3, device="cuda") * 0.2).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, cout, device="cuda")
    xc, dyc = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
    mb = x.numel() * 4 / 1e6
    print(f"size {B}x{H}x{W}x16 -> {cout} channels ({mb:.0f} MB input)")
    t = timed(lambda: be.img_conv16(x, w, with_sums=True))
    print(f"conv16 forward + BN sums : {t:7.1f} us  ({(1 + cout / 16) * mb / t:.2f} TB/s on x + y, {2 * 144 * cout * x.numel() / 16 / t / 1e6:.1f} TFLOP/s)")
    t = timed(lambda: be.img_conv16(dy, w, input_grad=True))
    print(f"conv16 input gradient    : {t:7.1f} us")
    print(f"conv16 weight gradient   : {timed(lambda: be.img_conv16_wgrad(x, dy, w)):7.1f} us  (pass + 9-block reduction)")
    if cout == 16:
        gam, bet = torch.randn(16, device="cuda"), torch.randn(16, device="cuda") * 0.2
        yk, sums = be.img_conv16(x, w, with_sums=True)
        out, arg, mi = be.img_block_forward(yk, gam, bet, 1e-5, 0.1, 1, sums=sums.clone())
        t1 = timed(lambda: be.img_block_backward(dy, arg, yk, mi, gam, bet, 0.1, 1))
        dyk = be.img_block_backward(dy, arg, yk, mi, gam, bet, 0.1, 1)[0]
        t2 = timed(lambda: be.img_conv16(dyk, w, input_grad=True))
        t3 = timed(lambda: be.img_conv16_tail_backward(dy, arg, yk, mi, gam, bet, 0.1, w))
        print(f"block backward, stride-1 pool : statistics + dy {t1:6.1f} us, then input gradient {t2:6.1f} us; statistics + ONE kernel {t3:6.1f} us")
        t4 = timed(lambda: be.img_block_forward(yk, gam, bet, 1e-5, 0.1, 1, sums=sums))
        print(f"block tail forward {t4:6.1f} us (+ the next convolution above)")
    import os
    if os.environ.get("I2P_TIME_CONV16_ONLY") == "1":
        return
    print(f"MIOpen forward           : {timed(lambda: F.conv2d(xc, w, None, 1, 1)):7.1f} us")
    bw = lambda: torch.ops.aten.convolution_backward(dyc, xc, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))
    print(f"MIOpen input gradient    : {timed(bw):7.1f} us")
    ww = lambda: torch.ops.aten.convolution_backward(dyc, xc, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))
    print(f"MIOpen weight gradient   : {timed(ww):7.1f} us")
    y = be.img_conv16(x, w)
    ref = F.conv2d(xc, w, None, 1, 1).permute(0, 2, 3, 1)
    dW, dW2 = be.img_conv16_wgrad(x, dy, w), ww()[1]
    print(f"max |dW - dW_miopen| / max = {(dW - dW2).abs().max().item() / dW2.abs().max().item():.2e}")
    print(f"max |y - y_miopen| / max = {(y - ref).abs().max().item() / ref.abs().max().item():.2e}")


if __name__ == "__main__":
    main()
