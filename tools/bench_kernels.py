"""Micro-benchmarks of the hand-written kernels at the model's shapes (events on the launch
stream; achieved GB/s on algorithmic bytes).  `python tools/bench_kernels.py [--batch 8]`"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from i2pnet_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = "cuda"
    hip = ops.hip_backend()
    B = a.batch
    print(f"{'kernel':34s} {'shape':24s} {'us':>9s} {'GB/s':>9s}")
    for name, rows, c in [("cv1 128ch", B * 228 * 468, 128), ("cv1 64ch", B * 228 * 468, 64), ("L1 16ch", B * 3600 * 32, 16),
                          ("L1 32ch", B * 3600 * 32, 32), ("L2 64ch", B * 904 * 16, 64), ("L4 256ch", B * 116 * 16, 256)]:
        y = torch.randn(rows, c, device=dev); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
        go = torch.randn(rows, c, device=dev)
        sums = torch.zeros(32 * 2 * c, dtype=torch.float64, device=dev)
        out = torch.empty_like(y); mi = torch.empty(2 * c, device=dev); dy = torch.empty_like(y)
        dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        P = lambda t: t.data_ptr()
        nbytes = rows * c * 4
        from i2pnet_amd import _lib
        t = timeit(lambda: _lib.call("i2p_bn_stats", rows, c, P(y), P(sums), stream=st))
        print(f"{'bn_stats':34s} {name:24s} {t:9.1f} {nbytes / t / 1e3:9.0f}")
        t = timeit(lambda: _lib.call("i2p_bn_act_fwd", rows, c, P(y), P(sums), P(g), P(b), 1e-5, 0.1, P(out), P(mi), stream=st))
        print(f"{'bn_act_fwd':34s} {name:24s} {t:9.1f} {2 * nbytes / t / 1e3:9.0f}")
        t = timeit(lambda: _lib.call("i2p_bn_act_bwd_stats", rows, c, P(go), P(y), P(mi), P(g), P(b), 0.1, P(sums), stream=st))
        print(f"{'bn_act_bwd_stats':34s} {name:24s} {t:9.1f} {2 * nbytes / t / 1e3:9.0f}")
        t = timeit(lambda: _lib.call("i2p_bn_act_bwd", rows, c, P(go), P(y), P(mi), P(g), P(b), 0.1, P(sums), P(dy), P(dg), P(db), stream=st))
        print(f"{'bn_act_bwd':34s} {name:24s} {t:9.1f} {3 * nbytes / t / 1e3:9.0f}")
    bench_lin(B)
    bench_pair(B)


def bench_lin(B):
    import torch.nn.functional as F
    hip = ops.hip_backend()
    dev = "cuda"
    print(f"{'layer':34s} {'shape':24s} {'us':>9s} {'TFLOP/s':>9s} {'GB/s':>9s}")
    for name, rows, cin, cout in [("cv1 128->128", B * 228 * 468, 128, 128), ("cv1 128->64", B * 228 * 468, 128, 64),
                                  ("cv1 64->64", B * 228 * 468, 64, 64), ("L1 16->32", B * 3600 * 32, 16, 32),
                                  ("L1 10->16", B * 3600 * 32, 10, 16), ("L2 35->32", B * 904 * 16, 35, 32),
                                  ("L3 64->128", B * 228 * 16, 64, 128)]:
        x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        coef = torch.stack([torch.zeros(cin), torch.ones(cin), torch.zeros(cin)]).to(dev).contiguous()
        t = timeit(lambda: hip.lin_forward(x, coef, 0.1, w))
        fl = 2.0 * rows * cin * cout; by = rows * (cin + cout) * 4
        print(f"{'lin_fwd (BN-on-load + stats)':34s} {name:24s} {t:9.1f} {fl / t / 1e6:9.1f} {by / t / 1e3:9.0f}")
        t = timeit(lambda: F.linear(x, w))
        print(f"{'torch F.linear (hipBLASLt)':34s} {name:24s} {t:9.1f} {fl / t / 1e6:9.1f} {by / t / 1e3:9.0f}")


def bench_pair(B):
    hip = ops.hip_backend(); dev = "cuda"
    N, M, C = 228, 468, 128
    f = torch.randn(B, N, C, device=dev); g = torch.randn(B, M, C, device=dev)
    bn = torch.randn(B, N, C, device=dev); bk = torch.randn(B, M, C, device=dev); w = torch.randn(C, C, device=dev) / 11
    t = timeit(lambda: hip.pair_lin_forward(f, g, bn, bk, w))
    fl = 2.0 * B * N * M * C * C
    print(f"{'pair_lin_fwd (cv1 layer 1)':34s} {'B*228*468 x 128->128':24s} {t:9.1f} {fl / t / 1e6:9.1f}")
    gy = torch.randn(B * N * M, C, device=dev)
    t = timeit(lambda: hip.pair_lin_backward(gy, f, g, w))
    print(f"{'pair_lin_bwd (cv1 layer 1)':34s} {'B*228*468 x 128->128':24s} {t:9.1f} {2 * fl / t / 1e6:9.1f}")
    import torch.nn.functional as F
    def torch_path():
        corr = f.unsqueeze(2) * g.unsqueeze(1)
        return F.linear(corr, w) + bn.unsqueeze(2) + bk.unsqueeze(1)
    t = timeit(torch_path)
    print(f"{'torch: mul + F.linear + 2 adds':34s} {'(forward only)':24s} {t:9.1f} {fl / t / 1e6:9.1f}")


if __name__ == "__main__":
    main()
