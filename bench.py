"""Benchmark of the hot path: data-parallel training step of the registration network on
synthetic KITTI-shaped batches (BASELINE.json configs[1]: 375x1242 RGB + 8192-point cloud,
batch 8 per GPU, fp32, forward + loss + backward + clip + Adam).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant
hand-written kernel, timed live with events on the launch stream) and, at N=1, `cpu_baseline`
(the same step on the host cores with the CPU oracle as operator backend, bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* (fp32 in/acc) = the fp32 vector peak


def process_setup():
    """The MIOpen configuration of the measured process, in ONE place: tools/bench_mode_parity.py (driven by
    tests/test_bench_mode_gpu.py) calls the same function, so the reference-pinned parity test runs under exactly the
    solver-selection mode the throughput line is measured in."""
    torch.backends.cudnn.benchmark = True   # MIOpen exhaustive find for the 15 image-encoder convolutions (in the warm-up steps)


def miopen_env():
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("MIOPEN_") and k != "MIOPEN_USER_DB_PATH"}


def solvers_from_find_db():
    """{problem key: fastest solver} from the user find-db text files of this process's MIOpen (what its find wrote / found)"""
    import glob
    roots = [os.environ.get("MIOPEN_USER_DB_PATH") or os.path.expanduser("~/.config/miopen")]
    out = {}
    for root in roots:
        for f in glob.glob(os.path.join(root, "**", "*.ufdb.txt"), recursive=True):
            try:
                lines = open(f, errors="replace").read().splitlines()
            except OSError:
                continue
            for line in lines:
                if "=" not in line:
                    continue
                key, val = line.strip().split("=", 1)
                best = None
                for e in val.split(";"):
                    if ":" not in e:
                        continue
                    name, rest = e.split(":", 1)
                    try:
                        t = float(rest.split(",")[0])
                    except ValueError:
                        continue
                    if best is None or t < best[1]:
                        best = (name, t)
                if best:
                    out[key] = best[0]
    return out


def _event_time_us(launch, iters):
    for _ in range(3):
        launch()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()                    # torch's current stream == the stream the kernels are launched on
    for _ in range(iters):
        launch()
    end.record(); end.synchronize()
    return start.elapsed_time(end) / iters * 1e3


def _graph_time_us(launch, per_replay=20, replays=10):
    """per-launch time of a host-bound small launch: `per_replay` launches captured in one hipGraph (zero-arena active, as in a step)"""
    from i2pnet_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.begin_step(dev)
    try:
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(per_replay):
                launch()
    finally:
        ops.end_step(dev)
    for _ in range(3):
        g.replay()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(replays):
        g.replay()
    end.record(); end.synchronize()
    return start.elapsed_time(end) * 1e3 / (replays * per_replay)


def _mlp_chain_entry(B, device):
    """level-3 set-abstraction MLP (67 -> 64 -> 64 -> 128 on B*228*16 rows, max over K = 16) as ONE launch (chain_fwd_kernel: resident
    grid, strips in LDS, a grid barrier per BN) against the layer-by-layer launches it replaces (3 x lin_fwd_fin + weight pad + BN /
    activation / max tail); in-graph launches, forward only"""
    import os
    from i2pnet_amd import fused
    rows, c0, cin, widths, K = B * 228 * 16, 128, 67, (64, 64, 128), 16
    g = torch.Generator(device=device).manual_seed(5)
    x = torch.randn(rows, c0, generator=g, device=device); x[:, cin:] = 0
    params, cp = [], cin
    for c in widths:
        params += [torch.randn(c, cp, generator=g, device=device) / cp ** 0.5, torch.ones(c, device=device), torch.zeros(c, device=device)]
        cp = c
    slopes = (1.0, 0.0, 0.0, 0.0)
    run = lambda: fused._MlpChain.apply(x, False, slopes, K, None, *params)
    old = os.environ.get("I2P_NO_CHAIN")
    try:
        with torch.no_grad():
            os.environ["I2P_NO_CHAIN"] = "1"; t_layers = _graph_time_us(run)
            os.environ["I2P_NO_CHAIN"] = "0"; t_chain = _graph_time_us(run)
    finally:
        if old is None:
            os.environ.pop("I2P_NO_CHAIN", None)
        else:
            os.environ["I2P_NO_CHAIN"] = old
    by = rows * 4 * (c0 + sum(widths)) + rows // K * widths[-1] * 5
    fl = 2.0 * rows * (c0 * 64 + 64 * 64 + 64 * 128)
    return {"kernel": "chain_fwd_kernel (level-3 set-abstraction MLP 67->64->64->128 + max over K=16, one launch; 12 such chains per step)",
            "bound": "latency (3 grid barriers, 456 resident blocks)", "avg_kernel_us": round(t_chain, 1), "layer_by_layer_us": round(t_layers, 1),
            "achieved": round(by / t_chain / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / t_chain / 1e3 / HBM_PEAK_GBS, 4),
            "bytes_per_launch_algorithmic": by, "mfma_TFLOPs": round(fl / t_chain / 1e6, 1), "timed_as": "20 launches per hipGraph replay"}


PMC_JSON = ROOT / "profiles" / "r06_pmc_traffic.json"            # written by tools/pmc_step.sh r06 on the round's final kernels
PMC_BF16_JSON = ROOT / "profiles" / "r05_pmc_bf16_traffic.json"   # {kernel: {"bytes_per_row": ...}}, tools/pmc_r05_bf16.py


def _pmc_traffic(kernel, B):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r05_pmc_traffic.json, written by
    tools/pmc_step.sh: 2*FETCH_SIZE + WRITE_SIZE, the gfx950 halving of FETCH_SIZE calibrated in the same run on
    bn_stats_v4), scaled from the batch the passes ran at; None when the file has no record of that kernel."""
    if not PMC_JSON.exists():
        return None
    rec = json.loads(PMC_JSON.read_text()).get("kernels", {}).get(kernel)
    if not rec or rec.get("bytes_per_launch") is None:
        return None
    return round(rec["bytes_per_launch"] * B / rec["B"])


class Cv1Chain:
    """The all-pixel cost volume's fused pi-stage node (i2pnet_amd.fused._CvPiTail: pair layer -> 128->64 -> 64->64,
    position encoding, 64+64->128 -> 128->64, softmax-weighted sum) on the tensors' real shapes at per-GPU batch B:
    what cost_volume1 runs inside the training step (PPBackbone_center.py:383-433).  Also the PMC workload of
    tools/pmc_step.py."""

    N, M, C = 228, 468, 128

    def __init__(self, B, device, seed=0, N=None):
        from i2pnet_amd import modules
        from i2pnet_amd.config import I2PNetConfig as cfg
        if N is not None:
            self.N = N
        from i2pnet_amd.model import RegNet_v2
        torch.manual_seed(seed)
        self.B, self.device = B, device
        self.cv = RegNet_v2(cfg=cfg).cost_volume1.to(device)       # the module the step runs, seeded weights
        g = torch.Generator(device=device).manual_seed(seed)
        rnd = lambda *s: torch.randn(*s, generator=g, device=device)
        N, M, C = self.N, self.M, self.C
        first = self.cv.mlp1_convs[0]
        c1 = first.out_channels
        ce = self.cv.pi_encoding.out_channels
        self.inputs = [rnd(B, N, C), rnd(B, M, C), rnd(B, N, c1), rnd(B, M, c1), rnd(c1, C) / C ** 0.5, rnd(B, N, ce), rnd(B, M, ce)]
        for t in self.inputs:
            t.requires_grad_(True)
        self.g_out = rnd(B, N, self.cv.mlp1_convs[-1].out_channels) * 1e-3
        self.rows = B * N * M
        self.modules = modules

    def forward(self):
        from i2pnet_amd.fused import cv_pi_tail
        cv = self.cv
        f, gk, bn, bk, w1, en, ek = self.inputs
        return cv_pi_tail(f, gk, bn, bk, w1, en, ek, cv.mlp1_convs[0], list(cv.mlp1_convs)[1:], cv.pi_encoding,
                          list(cv.mlp2_convs))

    def time_us(self, iters=10, warm=20):
        """(forward, backward) device time of the node, events on the launch stream around each half"""
        from i2pnet_amd import ops
        ev = lambda: torch.cuda.Event(enable_timing=True)
        tf = tb = 0.0
        for i in range(warm + iters):
            ops.begin_step(self.device)                   # the step's zero arena (one memset instead of a fill per accumulator)
            try:
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
                out = self.forward()
                e1.record()
                out.backward(self.g_out)
                e2.record()
            finally:
                ops.end_step(self.device)
            e2.synchronize()
            for t in self.inputs + list(self.cv.parameters()):
                t.grad = None
            if i >= warm:
                tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        return tf / iters * 1e3, tb / iters * 1e3


    def graph_time_us(self, replays=20):
        """(forward, backward) device time of the node as the training step runs it: replays of captured hipGraphs (the step's
        zero-arena memset alone; + forward; + backward; the differences are the two halves), events around `replays` back-to-back
        launches.  The eager
        figure of `time_us` carries the host's launch gaps between the node's ~30 kernels, which the captured step does not
        have (profiles/*_step_sequence.txt: gap 0.0 between them)."""
        from i2pnet_amd import ops
        ops.chain_error_words(self.device)                     # (registered outside any capture)

        def run(with_backward):
            ops.begin_step(self.device)
            try:
                out = self.forward() if with_backward is not None else None
                if with_backward:
                    out.backward(self.g_out)
            finally:
                ops.end_step(self.device)
            return out
        times = []
        for with_backward in (None, False, True):              # None: the zero-arena memset of begin_step alone (once per STEP, not per node)
            for t in self.inputs + list(self.cv.parameters()):
                t.grad = None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    run(with_backward)
                    for t in self.inputs + list(self.cv.parameters()):
                        t.grad = None
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                keep = run(with_backward)                      # noqa: F841 — keeps the graph's outputs alive
            for _ in range(5):
                graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                graph.replay()
            e1.record(); e1.synchronize()
            times.append(e0.elapsed_time(e1) / replays * 1e3)
            del graph, keep
        for t in self.inputs + list(self.cv.parameters()):
            t.grad = None
        return times[1] - times[0], times[2] - times[1]


# BN-exact no-recompute traffic and flops of the cost-volume pi-stage per sample and forward (cv1, fp32)
CV1_GB_PER_SAMPLE_FWD = 0.437
CV1_GFLOP_PER_SAMPLE_FWD = 15.1


MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (never the 2:1-sparsity figure)


def chain_roofline(B, device, bf16=False, N=228):
    """cv1 forward and backward device time (hipGraph replays of the fused node as the step launches it, `Cv1Chain.graph_time_us`;
    the eager event timing is reported beside it) against SURVEY.md §8(d)'s denominators:
    0.437 GB and 15.1 GFLOP per sample and forward; the backward does twice the flops (dgrad + wgrad of every layer) and, BN-exact
    and without recomputation, reads each of the six pre-BN tensors and its gradient once and writes each gradient once
    (3 x 512 channels x 4 B per pair = 0.655 GB per sample).  bf16=True (the caller has ops.set_precision("bf16") active): the same
    node in bf16 storage — half the bytes (0.2185 GB per sample), the same flops on the bf16 MFMA roof: the HBM-bound regime the
    north star's ">= 40 % of HBM on the fused grouping + cost-volume kernel" speaks about (VERDICT r3 missing #2)."""
    ch = Cv1Chain(B, device, N=N)
    e_f, e_b = ch.time_us()
    try:
        t_f, t_b = ch.graph_time_us()
        timed_as = ("replays of captured hipGraphs, as the training step launches the node (the step's zero-arena memset alone; + forward; "
                    "+ backward; each half = the difference), events around 20 back-to-back replays after 5 warm-up replays; *_eager_us: events around "
                    "cv_pi_tail(...) and its .backward() with eager launches (host launch gaps included), 10 iterations after 20")
    except Exception as e:                                     # noqa: BLE001 — report the eager figure and say so
        print("bench.py: hipGraph capture of the cv1 node failed (%s: %s); chain roofline from eager launches" % (type(e).__name__, e),
              file=sys.stderr, flush=True)
        t_f, t_b = e_f, e_b
        timed_as = "torch.cuda events on the launch stream around cv_pi_tail(...) and around its .backward(), eager launches, 10 iterations after 20 warm-up iterations"
    gb = CV1_GB_PER_SAMPLE_FWD * (0.5 if bf16 else 1.0) * (N / 228.0)       # (per-sample figures of SURVEY 8(d) are for 228 points)
    fwd_b, fwd_f = gb * 1e9 * B, CV1_GFLOP_PER_SAMPLE_FWD * 1e9 * B * (N / 228.0)
    bwd_b, bwd_f = 1.5 * fwd_b, 2.0 * fwd_f
    mfma_peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS
    return {"node": "cost_volume1 pi-stage (_CvPiTail): pair layer, 128->64, 64->64, position encoding, 64+64->128, 128->64, softmax-weighted sum; "
                    "batch %d, %d point x pixel pairs" % (B, ch.rows),
            "forward_us": round(t_f, 1), "backward_us": round(t_b, 1), "forward_eager_us": round(e_f, 1), "backward_eager_us": round(e_b, 1),
            "forward": {"hbm_GBps": round(fwd_b / t_f / 1e3, 1), "hbm_frac": round(fwd_b / t_f / 1e3 / HBM_PEAK_GBS, 4),
                        "mfma_TFLOPs": round(fwd_f / t_f / 1e6, 1), "mfma_frac": round(fwd_f / t_f / 1e6 / mfma_peak, 4),
                        "bytes": fwd_b, "flop": fwd_f},
            "backward": {"hbm_GBps": round(bwd_b / t_b / 1e3, 1), "hbm_frac": round(bwd_b / t_b / 1e3 / HBM_PEAK_GBS, 4),
                         "mfma_TFLOPs": round(bwd_f / t_b / 1e6, 1), "mfma_frac": round(bwd_f / t_b / 1e6 / mfma_peak, 4),
                         "bytes": bwd_b, "flop": bwd_f},
            "storage": "bf16" if bf16 else "fp32", "mfma_peak_TFLOPs": mfma_peak,
            "denominators": "SURVEY.md 8(d): %s GB + 15.1 GFLOP per sample and forward (%s activations); backward 1.5x the bytes, 2x the flops"
                            % (("0.2185", "bf16") if bf16 else ("0.437", "fp32")),
            "timed_as": timed_as}


def _image_encoder_entry(B, device):
    """the hand-written kernels of the image encoder's blocks 1-5 at the step's shapes (csrc/image_first.hip, csrc/image_conv16.hip;
    DESIGN.md section 4): event timings on the launch stream, fp32 MFMA rate against 157.3 TFLOP/s and algorithmic bytes against 8 TB/s"""
    from i2pnet_amd import ops
    hip = ops.hip_backend()
    g = torch.Generator(device=device).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g, device=device)
    H, W = 375, 1242
    x = rnd(B, 3, H, W)
    w1 = (rnd(16, 3, 3, 3) * 0.3).contiguous(memory_format=torch.channels_last)
    gam, bet = rnd(16), rnd(16) * 0.2
    out, arg, mi, gram = hip.img_first_forward(x, w1, gam, bet, 1e-5, 0.1, 2)
    go = torch.randn_like(out)
    _event_time_us(lambda: hip.img_first_forward(x, w1, gam, bet, 1e-5, 0.1, 2), 30)
    t1f = _event_time_us(lambda: hip.img_first_forward(x, w1, gam, bet, 1e-5, 0.1, 2), 15)
    t1b = _event_time_us(lambda: hip.img_first_backward(go, arg, x, w1, gam, bet, 0.1, 2, mi, gram), 15)
    H2, W2 = out.shape[1], out.shape[2]
    a = rnd(B, H2, W2, 16)
    w = (rnd(16, 16, 3, 3) * 0.2).contiguous(memory_format=torch.channels_last)
    y = hip.img_conv16(a, w)
    o2, arg2, mi2 = hip.img_block_forward(y, gam, bet, 1e-5, 0.1, 1)
    gin = torch.randn_like(o2)
    _event_time_us(lambda: hip.img_conv16(a, w), 30)
    tcf = _event_time_us(lambda: hip.img_conv16(a, w), 20)
    tcw = _event_time_us(lambda: hip.img_conv16_wgrad(a, y, w), 20)
    tcb = _event_time_us(lambda: hip.img_conv16_tail_backward(gin, arg2, y, mi2, gam, bet, 0.1, w), 20)
    tb = a.numel() * 4
    flop = 2.0 * 144 * 16 * a.numel() / 16
    mf = lambda t: round(flop / t / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)
    return {"kernel": "image encoder blocks 1-5 (csrc/image_first.hip, csrc/image_conv16.hip): first block without its conv output; 3x3 "
                      "convolutions of the 188x621 stage on MFMA",
            "first_block": {"forward_us (Gram + coefficients + conv/BN/act/pool)": round(t1f, 1), "backward_us (sparse pass + reduction)": round(t1b, 1),
                            "algorithmic_bytes_forward": int(2 * x.numel() * 4 + out.numel() * 5), "algorithmic_bytes_backward": int(x.numel() * 4 + out.numel() * 5),
                            "hbm_frac_forward": round((2 * x.numel() * 4 + out.numel() * 5) / t1f / 1e3 / HBM_PEAK_GBS, 4),
                            "hbm_frac_backward": round((x.numel() * 4 + out.numel() * 5) / t1b / 1e3 / HBM_PEAK_GBS, 4),
                            "replaces": "channels_last copy + MIOpen zero-fill + igemm forward + statistics + pooling (310 us); BN-backward statistics "
                                        "+ dy + MIOpen weight gradient (397 us)"},
            "conv16_16_at_%dx%dx%d" % (B, H2, W2): {
                "bound": "mfma", "flop": flop, "forward_us (no sums: host-timed wrapper without the zero-fill)": round(tcf, 1), "forward_mfma_frac": mf(tcf),
                "weight_gradient_us (pass + fixed-order reduction)": round(tcw, 1), "weight_gradient_mfma_frac": mf(tcw),
                "tail_backward_us (statistics pass + un-pool/BN-backward/input-gradient kernel)": round(tcb, 1),
                "bytes_per_tensor": tb, "replaces": "MIOpen igemm: 101 us forward (+ 17 statistics), 79 + 18 zero-fill input gradient, 130 + 5 weight gradient"}}


def _kernel_only_us(fn, iters):
    """average duration of the headline kernel inside `fn()` — bracketed by the library's own events (i2p_ktime_*), see common.h"""
    from i2pnet_amd import _lib
    _lib.helper("i2p_ktime_enable", 1)
    try:
        tot = 0.0
        for _ in range(iters):
            fn()
            t = float(_lib.helper("i2p_ktime_last_us"))
            if t < 0:
                return float("nan")
            tot += t
        return tot / iters
    finally:
        _lib.helper("i2p_ktime_enable", 0)


def kernel_rooflines(B, device):
    """Live timings (events on the launch stream) of hand-written kernels AS THE fp32 STEP RUNS THEM: every kernel named here
    is an instantiation that appears in the step's rocprofv3 table (profiles/r03_*_steady_kernel_stats.csv) on the same shapes.

    * `roofline`: wreg_bwd_fused_kernel<64,128> — the backward of a 128->64 layer of the all-pixel cost volume on [B*228*468] rows
      (mlp1[1] and mlp2[1] of cost_volume1: two launches per step, the largest per-step time of the hand-written kernels) as ONE
      pass (csrc/mlp_wreg_fused.hip): it reads gz [rows,64], the layer's pre-BN output y [rows,64] (BN backward of the layer behind
      formed on load) and the pre-BN input x [rows,128] ONCE, writes dL/dz_in [rows,128] and keeps the [64][128] weight gradient in
      registers: rows*(2*64+2*128)*4 B against 4*rows*128*64 flop => HBM-bound (164 us at 8 TB/s, 178 us at 157.3 TFLOP/s: close
      to the ridge).  `traffic` = PMC bytes per launch from profiles/r06_pmc_traffic.json (tools/pmc_step.sh r06).
    * other_kernels: the forward of the same layer (wreg_fwd_kernel<128,64,true,false>), the two-source 64+64->128 forward
      (wreg_fwd_kernel<128,128,true,true>), the factored first layer (wreg_pair_fwd_kernel<128,128>), level-1
      fused_conv_select_k and the fused level-1 grouping on the three input densities.
    * chain: the whole cost_volume1 pi-stage node against SURVEY.md 8(d)'s per-sample bytes / flops (chain_roofline).
    """
    from i2pnet_amd import _lib, ops, projectpn as P, synth
    hip = ops.hip_backend()
    N, M = 228, 468
    CI, CO = 128, 64
    rows = B * N * M
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=device)
    x = rnd(rows, CI); w = rnd(CO, CI) / CI ** 0.5
    gam_i = torch.ones(CI, device=device); bet_i = torch.zeros(CI, device=device)
    gam_o = torch.ones(CO, device=device); bet_o = torch.zeros(CO, device=device)
    sx = hip.bn_stats(x)
    in_coef, in_mi = hip.bn_finalize(rows, sx, gam_i, bet_i, 1e-5)
    y = torch.empty(rows, CO, device=device)
    sy = torch.zeros(ops.BN_REPLICAS * 2 * CO, dtype=torch.float64, device=device)
    st = torch.cuda.current_stream().cuda_stream
    fwd_call = lambda: _lib.call("i2p_lin_fwd", rows, CI, CO, x.data_ptr(), in_coef.data_ptr(), 0.1, w.data_ptr(),
                                 y.data_ptr(), sy.data_ptr(), stream=st)
    # --- forward: exactly one launch per call, buffers preallocated -------------------------------------------
    _event_time_us(fwd_call, 150)                                  # (clocks ramp for ~50 ms after idle)
    t_fwd = _event_time_us(fwd_call, 25)
    flop = 2.0 * rows * CI * CO
    fwd_bytes = rows * (CI + CO) * 4 + CI * CO * 4
    fwd = {"kernel": "wreg_fwd_kernel<128,64,true,false> (cost-volume 128->64 layer forward, 2 launches per step: weights stationary in "
                     "registers, BN+act of the layer in front between the MFMAs, fp64-reduced BN statistics)",
           "bound": "hbm", "achieved": round(fwd_bytes / t_fwd / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(fwd_bytes / t_fwd / 1e3 / HBM_PEAK_GBS, 4), "avg_kernel_us": round(t_fwd, 1),
           "traffic": _pmc_traffic("wreg_fwd_kernel<128, 64, true, false>", B), "bytes_per_launch_algorithmic": fwd_bytes,
           "mfma_TFLOPs": round(flop / t_fwd / 1e6, 1), "mfma_frac": round(flop / t_fwd / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)}
    sy.zero_()
    fwd_call()
    out_coef, out_mi = hip.bn_finalize(rows, sy, gam_o, bet_o, 1e-5)
    gz = rnd(rows, CO)
    ods = torch.zeros(ops.BN_REPLICAS * 2 * CO, dtype=torch.float64, device=device)
    _event_time_us(lambda: hip.lin_backward(gz, y, out_coef, out_mi, ods, x, in_coef, in_mi, 0.1, w), 40)
    t_entry = _event_time_us(lambda: hip.lin_backward(gz, y, out_coef, out_mi, ods, x, in_coef, in_mi, 0.1, w), 20)
    # the kernel ALONE: events recorded by the launcher around wreg_bwd_fused_kernel itself (i2p_ktime_*), not around the entry's
    # coefficient and slab-reduction launches — the figure the rocprofv3 table's per-size row can be compared with directly
    t_bwd = _kernel_only_us(lambda: hip.lin_backward(gz, y, out_coef, out_mi, ods, x, in_coef, in_mi, 0.1, w), 20)
    dg_bytes = rows * (2 * CO + 2 * CI) * 4 + CI * CO * 4
    dgrad = {"kernel": "wreg_bwd_fused_kernel<64,128> (cost-volume 128->64 layer backward in ONE pass, 2 launches per step at this size: input "
                       "gradient + weight gradient from a single read of gz, y, x; W and the dW accumulators stationary in registers, BN backward "
                       "of the layer behind formed on load, operands of the weight gradient transposed through wave-private LDS tiles, activation "
                       "derivative + BN-backward statistics in the store phase)",
             "bound": "hbm", "achieved": round(dg_bytes / t_bwd / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(dg_bytes / t_bwd / 1e3 / HBM_PEAK_GBS, 4), "traffic": _pmc_traffic("wreg_bwd_fused_kernel<64, 128, false>", B),
             "avg_kernel_us": round(t_bwd, 1), "bytes_per_launch_algorithmic": dg_bytes,
             "mfma_TFLOPs": round(2 * flop / t_bwd / 1e6, 1), "mfma_frac": round(2 * flop / t_bwd / 1e6 / MFMA_F32_PEAK_TFLOPS, 4),
             "entry_us (kernel + BN-coefficient launch + 256-slab reduction of dW)": round(t_entry, 1),
             "timed_as": "the kernel alone: HIP events recorded by its launcher on the launch stream directly before and after "
                         "wreg_bwd_fused_kernel (i2p_ktime_enable / i2p_ktime_last_us), average of 20 launches with warm clocks; "
                         "compare with the (kernel, blocks = 256, 853632-row) row of profiles/r06_*_steady_kernel_stats.csv",
             "replaces": "wreg_dgrad_kernel<64,128,false> + wreg_wgrad_kernel<64,128,true,false> (two reads of the same tensors: 430 us)"}
    del x, y, gz
    # --- two-source forward (position encoding 64 + mlp1 output 64 -> 128), 1 launch per step ------------------------
    C2 = 64
    xa, xb = rnd(rows, C2), rnd(rows, C2)
    g2 = torch.ones(C2, device=device); b2 = torch.zeros(C2, device=device)
    ca, _ = hip.bn_finalize(rows, hip.bn_stats(xa), g2, b2, 1e-5)
    cb, _ = hip.bn_finalize(rows, hip.bn_stats(xb), g2, b2, 1e-5)
    w2 = rnd(128, 2 * C2) / (2 * C2) ** 0.5
    t_2s = _event_time_us(lambda: hip.lin_forward_2src(xa, ca, 0.1, xb, cb, 0.1, w2), 20)
    flop2 = 2.0 * rows * 128 * 128
    b2s = rows * (2 * C2 + 128) * 4
    two = {"kernel": "wreg_fwd_kernel<128,128,true,true> (64+64->128 two-source layer forward, no concatenation; 1 launch per step)",
           "bound": "mfma", "achieved": round(flop2 / t_2s / 1e6, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(flop2 / t_2s / 1e6 / MFMA_F32_PEAK_TFLOPS, 4), "avg_kernel_us": round(t_2s, 1),
           "traffic": _pmc_traffic("wreg_fwd_kernel<128, 128, true, true>", B), "bytes_per_launch_algorithmic": b2s,
           "hbm_GBps_algorithmic": round(b2s / t_2s / 1e3, 1)}
    # --- the same layer's backward in ONE pass (round 6): wreg_bwd_fused_kernel<128,64,TWO>, a wave per (strip, source) -------
    mia = hip.bn_finalize(rows, hip.bn_stats(xa), g2, b2, 1e-5)[1]; mib = hip.bn_finalize(rows, hip.bn_stats(xb), g2, b2, 1e-5)[1]
    y2s, s2s = hip.lin_forward_2src(xa, ca, 0.1, xb, cb, 0.1, w2)
    g128 = torch.ones(128, device=device); b128 = torch.zeros(128, device=device)
    oc2, om2 = hip.bn_finalize(rows, s2s, g128, b128, 1e-5)
    gz2 = rnd(rows, 128) * 0.1; ead = rnd(rows, C2) * 0.1
    ods2 = hip.bn_act_backward_stats(gz2, y2s, om2, g128, b128, 0.1)
    run2 = lambda: hip.lin_backward_2src(gz2, y2s, oc2, om2, ods2, xa, ca, mia, 0.1, xb, cb, mib, 0.1, ead, w2)
    _event_time_us(run2, 30)
    t_e2 = _event_time_us(run2, 15)
    t_b2 = _kernel_only_us(run2, 15)
    b2b = rows * (2 * 128 + 2 * 128 + C2) * 4
    two_bwd = {"kernel": "wreg_bwd_fused_kernel<128,64,TWO> (64+64->128 two-source layer backward in ONE pass, 1 launch per step at this size: a wave "
                         "owns (16-row strip, source) — W and dW columns of its 64 input channels in registers, g^y for all 128 output channels "
                         "formed in the registers gz arrived in; both input gradients, both sources' BN-backward statistics and dW from one "
                         "read of gz, y, xa, xb, e_add)",
               "bound": "mfma", "achieved": round(2 * flop2 / t_b2 / 1e6, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": round(2 * flop2 / t_b2 / 1e6 / MFMA_F32_PEAK_TFLOPS, 4), "avg_kernel_us": round(t_b2, 1),
               "traffic": _pmc_traffic("wreg_bwd_fused_kernel<128, 64, true>", B), "bytes_per_launch_algorithmic": b2b,
               "hbm_GBps_algorithmic": round(b2b / t_b2 / 1e3, 1), "entry_us (kernel + 256-slab reduction of dW)": round(t_e2, 1),
               "replaces": "wreg_dgrad_kernel<128,128,true> + wreg_wgrad_kernel<128,128,true,true> (two reads of gz / y / xa / xb: 705-720 us)"}
    del xa, xb, y2s, gz2, ead
    # --- pair-mode forward (first cost-volume layer) -----------------------------------------------------
    C = 128
    w = rnd(C, C) / C ** 0.5
    f = rnd(B, N, C); gk = rnd(B, M, C); bn = rnd(B, N, C); bk = rnd(B, M, C)
    t_pf = _event_time_us(lambda: hip.pair_lin_forward(f, gk, bn, bk, w), 10)
    flop_pf = 2.0 * rows * C * C
    pf = {"kernel": "wreg_pair_fwd_kernel<128,128> (first cost-volume layer forward: product formed from lane-private / wave-shared LDS rows, only y touches HBM)", "bound": "mfma",
          "achieved": round(flop_pf / t_pf / 1e6, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": round(flop_pf / t_pf / 1e6 / MFMA_F32_PEAK_TFLOPS, 4), "avg_kernel_us": round(t_pf, 1),
          "traffic": _pmc_traffic("wreg_pair_fwd_kernel<128, 128>", B), "hbm_GBps_on_output": round(rows * C * 4 / t_pf / 1e3, 1)}
    # --- level-1 neighbour selection -----------------------------------------------------------------------
    raw = synth.lidar_scan(B, 8192, torch.Generator(device=device).manual_seed(0), device, layout="centre")
    img, _, _ = hip.project_seq(raw, [], 64, 1800, 2.0, -24.8)
    idx = P.get_stride_idx_cuda(B, 16, 225, 4, 8, device)
    rhw = torch.arange(135, dtype=torch.int32, device=device)
    sel = torch.zeros(3, B, 3600, 32, 1, dtype=torch.long, device=device)
    mask = torch.zeros(B, 3600, 32, 1, device=device)
    unused = torch.zeros(1, device=device)
    run_sel = lambda: hip.fused_conv_select_k(img, img, idx, rhw, 64, 1800, 3600, 9, 15, 32, 3, 0.75, 1, 1,
                                              sel[0], sel[1], sel[2], unused, unused, mask, 64, 1800)
    _event_time_us(run_sel, 200)                                   # (clocks ramp for ~50 ms after idle: tools/time_fcsk.py does the same)
    t_sel = _event_time_us(run_sel, 100)
    bytes_sel = B * (64 * 1800 * 12 + 3600 * 8 + 3600 * 32 * 28)
    selk = {"kernel": "fcsk_kernel<9> (fused_conv_select_k, level 1, all 3600 queries live)", "bound": "hbm",
            "achieved": round(bytes_sel / t_sel / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bytes_sel / t_sel / 1e3 / HBM_PEAK_GBS, 4), "avg_kernel_us": round(t_sel, 2),
            "bytes_per_launch": bytes_sel}
    # --- level-1 grouping as one kernel (selection + gather + feature build, window strip staged in LDS) against the
    #     unfused chain it replaces, on the three input densities of SURVEY.md §0.5 -----------------------------------
    from i2pnet_amd import modules
    grp = []
    net = modules.ProjectPointNet(64, 1800, 16, 225, 4, 8, [9, 15], 32, 0.75, 12, [16, 16, 32], use_trans=True).to(device)
    for name, npts, layout, zero_rows in (("8192-pt scan (7 % occupancy: most level-1 queries exit early)", 8192, "scan", 0),
                                          ("8192-pt centre-aligned (all 3600 queries live)", 8192, "centre", 0),
                                          ("150000-pt scan (120000 real + 30000 zero rows, the reference loader's padding)", 150000, "scan", 30000)):
        cloud = synth.lidar_scan(B, npts, torch.Generator(device=device).manual_seed(1), device, layout=layout, zero_rows=zero_rows,
                                 beams=64)
        im, _, _ = hip.project_seq(cloud, [], 64, 1800, 2.0, -24.8)
        occ = float((im != 0).any(-1).float().mean())
        _event_time_us(lambda: hip.sa_l1_group(im, im, 16, 225, 4, 8, 9, 15, 32, 0.75), 200)      # warm clocks, as tools/time_sa_l1.py
        t_f = _event_time_us(lambda: hip.sa_l1_group(im, im, 16, 225, 4, 8, 9, 15, 32, 0.75), 100)
        feats = {}
        def chain(fused):
            modules.USE_FUSED_GROUP = fused
            orig = net._mlp_max
            net._mlp_max = lambda x, B_: x                     # stop in front of the MLP: grouping front end only
            try:
                with torch.no_grad():
                    feats[fused] = net.forward_center(im, im, None, raw_feat_point=True)[2]
            finally:
                net._mlp_max = orig; modules.USE_FUSED_GROUP = True
        t_u = _event_time_us(lambda: chain(False), 10)
        by = B * (2 * 64 * 1800 * 12 + 3600 * 32 * 48)
        grp.append({"input": name, "occupancy": round(occ, 3), "fused_us": round(t_f, 1), "unfused_chain_us": round(t_u, 1),
                    "GBps_algorithmic": round(by / t_f / 1e3, 1), "frac_of_hbm_peak": round(by / t_f / 1e3 / HBM_PEAK_GBS, 4)})
    group = {"kernel": "sa_l1_kernel<9> (level-1 selection + gather + feature build in one launch; unfused = fused_conv_select_k + 2 row "
                       "gathers + subtract + norm + cat, host-timed eager launches)", "bound": "hbm (in practice L2 / LDS / issue)",
             "bytes_per_launch": B * (2 * 64 * 1800 * 12 + 3600 * 32 * 48), "cases": grp}
    # (new entries go to the END: the reviews of rounds 4-5 cite other_kernels[3..5] by index)
    dgrad["other_kernels"] = [fwd, two, pf, selk, group, _mlp_chain_entry(B, device), _image_encoder_entry(B, device), two_bwd]
    del f, gk, bn, bk
    torch.cuda.empty_cache()
    dgrad["chain"] = chain_roofline(B, device)
    return dgrad


def kernel_rooflines_bf16(B, device, N=228):
    """configs[2] / configs[4]: live timings of the bf16-storage cost-volume kernels on the step's own tensors
    (rows = B x N x 468; N = 228 level-3 points for KITTI, 171 for nuScenes).  All HBM-bound (the bf16 MFMA roof is ~300 flop/B).

    * `roofline`: bwd_fused_bf16_kernel<128> — the backward of a 128->64 layer of the all-pixel cost volume (mlp1[1] and mlp2[1] of
      cost_volume1: two launches per step, the largest per-step time among the hand-written kernels in
      profiles/r05_bf16_c2_steady_kernel_stats.csv) in ONE pass (csrc/mlp_bwd_fused_bf16.hip): gz, y [rows,64] and x [rows,128] read
      once, dL/dz_in [rows,128] written once = rows*(2*64+2*128)*2 B.  Timed as the kernel alone (i2p_ktime_*: events recorded by the
      launcher around the kernel).  `traffic`: PMC bytes per launch from profiles/r05_pmc_bf16_traffic.json (tools/pmc_kernels.sh).
    * other_kernels: the 64->64 one-pass backward, the two-source 64+64->128 one-pass backward, the pair-layer backward (second
      generation) and forward, the 128->128 / 128->64 layer forwards.
    * chain: the whole cost_volume1 pi-stage node in bf16 storage against SURVEY 8(d)'s bytes halved (0.2185 GB per sample)."""
    from i2pnet_amd import _lib, ops
    hip = ops.hip_backend()
    M, C = 468, 128
    rows = B * N * M
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=device)
    bf = torch.bfloat16
    gam = lambda c: torch.ones(c, device=device)
    bet = lambda c: torch.zeros(c, device=device)
    pmc = json.loads(PMC_BF16_JSON.read_text()) if PMC_BF16_JSON.exists() else {}

    def traffic_of(name):
        rec = pmc.get(name)
        return round(rec["bytes_per_row"] * rows) if rec else None

    def entry(kernel, nbytes, t, **extra):
        d = {"kernel": kernel, "bound": "hbm", "achieved": round(nbytes / t / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(nbytes / t / 1e3 / HBM_PEAK_GBS, 4), "avg_kernel_us": round(t, 1), "bytes_per_launch_algorithmic": nbytes}
        d.update(extra)
        return d

    def layer(cin, cout):
        """a genuine pre-BN bf16 tensor x with its statistics, the layer's output y, its BN, a gradient and its statistics"""
        x0 = rnd(rows, cin).to(bf)
        w0 = rnd(cin, cin) / cin ** 0.5
        x, s0 = hip.lin_forward(x0, None, 1.0, w0, out_dtype=bf)
        del x0
        in_coef, in_mi = hip.bn_finalize(rows, s0, gam(cin), bet(cin), 1e-5)
        w = rnd(cout, cin) / cin ** 0.5
        y, sy = hip.lin_forward(x, in_coef, 0.1, w, out_dtype=bf)
        out_coef, out_mi = hip.bn_finalize(rows, sy, gam(cout), bet(cout), 1e-5)
        gz = (rnd(rows, cout) * 0.1).to(bf)
        ods = hip.bn_act_backward_stats_bf16(gz, y, out_coef, out_mi, 1.0)
        return x, in_coef, in_mi, w, y, out_coef, out_mi, gz, ods

    # ---- headline: 128 -> 64 one-pass backward --------------------------------------------------------------------------------
    x, ic, im, w, y, oc, om, gz, ods = layer(128, 64)
    run = lambda: hip.lin_backward(gz, y, oc, om, ods, x, ic, im, 0.1, w)
    _event_time_us(run, 150)                                        # (clocks ramp for ~50 ms after idle)
    t_entry = _event_time_us(run, 20)
    t_k = _kernel_only_us(run, 20)
    nb = rows * (2 * 64 + 2 * 128) * 2
    head = entry("bwd_fused_bf16_kernel<128> (cost-volume 128->64 layer backward in ONE pass, 2 launches per step: input gradient + "
                 "weight gradient from a single read of gz, y, x; 32-row strips staged twice through wave-private LDS images — row-major "
                 "for the input gradient, transposed for the weight gradient —, v_mfma_f32_32x32x16_bf16, BN backward on load, activation "
                 "derivative + BN-backward statistics in the store phase)", nb, t_k,
                 traffic=traffic_of("bwd_fused_bf16_kernel<128>"),
                 **{"entry_us (kernel + BN-coefficient launch + 256-slab reduction of dW)": round(t_entry, 1),
                    "timed_as": "the kernel alone: HIP events recorded by its launcher directly around bwd_fused_bf16_kernel "
                                "(i2p_ktime_enable / i2p_ktime_last_us), average of 20 launches with warm clocks",
                    "replaces": "rg_dgrad_kernel<4> + wreg_wgrad_bf16_kernel<64,128> (two reads of the same tensors: 551 us at batch 16)"})
    t_f = _event_time_us(lambda: hip.lin_forward(x, ic, 0.1, w, out_dtype=bf), 20)
    others = [entry("rg_fwd_kernel<2,true,false> (128->64 layer forward, 2 launches per step)", rows * (128 + 64) * 2, t_f)]
    del x, y, gz
    x, ic, im, w, y, oc, om, gz, ods = layer(64, 64)
    run = lambda: hip.lin_backward(gz, y, oc, om, ods, x, ic, im, 0.1, w)
    _event_time_us(run, 20)
    others.append(entry("bwd_fused_bf16_kernel<64> (64->64 layer backward in one pass, 1 launch per step)", rows * 4 * 64 * 2, _kernel_only_us(run, 20),
                        traffic=traffic_of("bwd_fused_bf16_kernel<64>")))
    del x, y, gz
    # ---- two-source 64 + 64 -> 128 ---------------------------------------------------------------------------------------------
    w64 = rnd(64, 64) / 8.0
    xa, sa = hip.lin_forward(rnd(rows, 64).to(bf), None, 1.0, w64, out_dtype=bf)
    xb, sb = hip.lin_forward(rnd(rows, 64).to(bf), None, 1.0, w64, out_dtype=bf)
    ca, ma = hip.bn_finalize(rows, sa, gam(64), bet(64), 1e-5)
    cb, mb = hip.bn_finalize(rows, sb, gam(64), bet(64), 1e-5)
    w2 = rnd(128, 128) / 128 ** 0.5
    y2, s2 = hip.lin_forward_2src(xa, ca, 0.1, xb, cb, 0.1, w2)
    t_2f = _event_time_us(lambda: hip.lin_forward_2src(xa, ca, 0.1, xb, cb, 0.1, w2), 20)
    others.append(entry("rg_fwd_kernel<4,true,false> (64+64->128 two-source layer forward, 1 launch per step)", rows * 256 * 2, t_2f))
    oc2, om2 = hip.bn_finalize(rows, s2, gam(128), bet(128), 1e-5)
    gz2 = (rnd(rows, 128) * 0.1).to(bf); ea = (rnd(rows, 64) * 0.1).to(bf)
    ods2 = hip.bn_act_backward_stats_bf16(gz2, y2, oc2, om2, 1.0)
    run2 = lambda: hip.lin_backward_2src(gz2, y2, oc2, om2, ods2, xa, ca, ma, 0.1, xb, cb, mb, 0.1, ea, w2)
    _event_time_us(run2, 20)
    others.append(entry("bwd_fused2_bf16_kernel (64+64->128 backward in one pass: dW in all 256 accumulator registers, input gradient on "
                        "v_mfma_f32_16x16x32_bf16; + coefficient and reduction launches)", rows * (2 * 128 + 3 * 64 + 128) * 2, _event_time_us(run2, 20),
                        traffic=traffic_of("bwd_fused2_bf16_kernel")))
    del xa, xb, y2, gz2, ea
    # ---- pair layer --------------------------------------------------------------------------------------------------------------
    w = rnd(C, C) / C ** 0.5
    f = rnd(B, N, C); gk = rnd(B, M, C); bn = rnd(B, N, C); bk = rnd(B, M, C)
    t_pf = _event_time_us(lambda: hip.pair_lin_forward(f, gk, bn, bk, w, out_dtype=bf), 10)
    others.append(entry("pair_fwd3_bf16_kernel (first cost-volume layer forward: 32-pixel strip shared by the four waves of a block, output bf16)", rows * C * 2, t_pf))
    y1, s1 = hip.pair_lin_forward(f, gk, bn, bk, w, out_dtype=bf)
    c1, m1 = hip.bn_finalize(rows, s1, gam(C), bet(C), 1e-5)
    gz1 = (rnd(rows, C) * 0.1).to(bf)
    ds1 = hip.bn_act_backward_stats_bf16(gz1, y1, c1, m1, 1.0)
    runp = lambda: hip.pair_lin_backward(gz1, f, gk, w, y=y1, out_coef=c1, out_mi=m1, out_dsums=ds1)
    _event_time_us(runp, 10)
    others.append(entry("pair_bwd2_bf16_kernel (first cost-volume layer backward, second generation; + coefficient and reduction launches)",
                        rows * C * 2 * 2, _event_time_us(runp, 10), traffic=traffic_of("pair_bwd2_bf16_kernel")))
    head["other_kernels"] = others
    del y1, gz1, f, gk, bn, bk
    torch.cuda.empty_cache()
    head["chain"] = chain_roofline(B, device, bf16=True, N=N)
    return head


def _oracle_op_times():
    """microseconds of the scalar (single-thread) CPU restatements of the reference's CUDA operators at the shapes of
    one KITTI-shaped sample: what `cpu_baseline`'s step spends outside PyTorch-CPU's GEMMs."""
    from i2pnet_amd import projectpn as P, synth
    from oracle import oracle
    cpu = oracle.backend()
    out = {}

    def t(fn, reps=1):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return round((time.perf_counter() - t0) / reps * 1e6, 1)
    raw = synth.lidar_scan(1, 8192, torch.Generator().manual_seed(0), "cpu", layout="centre")
    out["project_seq 8192 pts -> 64x1800"] = t(lambda: cpu.project_seq(raw, [], 64, 1800, 2.0, -24.8), 3)
    img, _, _ = cpu.project_seq(raw, [], 64, 1800, 2.0, -24.8)
    idx = P.get_stride_idx_cuda(1, 16, 225, 4, 8, "cpu")
    rhw = torch.arange(135, dtype=torch.int32)
    sel = torch.zeros(3, 1, 3600, 32, 1, dtype=torch.long); mask = torch.zeros(1, 3600, 32, 1); un = torch.zeros(1)
    out["fused_conv_select_k level 1 (3600 queries x 135 cells, K=32)"] = t(
        lambda: cpu.fused_conv_select_k(img, img, idx, rhw, 64, 1800, 3600, 9, 15, 32, 3, 0.75, 1, 1, sel[0], sel[1], sel[2], un, un, mask, 64, 1800))
    pts = synth.lidar_scan(1, 8192, torch.Generator().manual_seed(1), "cpu")
    fidx = torch.zeros(1, 2048, dtype=torch.int32)
    out["furthest_point_sampling 8192 -> 2048"] = t(lambda: cpu.furthest_point_sampling_wrapper(1, 8192, 2048, pts, torch.full((1, 8192), 1e10), fidx))
    q = torch.rand(1, 228, 3); k = torch.rand(1, 468, 3); kid = torch.zeros(1, 228, 32, dtype=torch.int32)
    out["knn 228 x 468, k=32"] = t(lambda: cpu.knn(k, q, 32, kid), 3)
    feat = torch.rand(1, 3600, 32); hh = torch.zeros(1, 904 * 16, dtype=torch.long); ww = torch.randint(0, 3600, (1, 904 * 16))
    go = torch.rand(1, 904 * 16, 32); gf = torch.zeros(1, 3600, 32)
    out["gather_rows_grad level 2 (14464 rows x 32 ch)"] = t(lambda: cpu.gather_rows_grad(go, hh, ww, 3600, gf), 3)
    return out


def cpu_baseline(cfg, batch_size=8, thread_counts=(3, 8, 16, 32, 64, 128)):
    """The same training step on the host: PyTorch-CPU model + CPU oracle operators (`kind: port`: the reference has no
    CPU path of its own), at configs[1]'s own shape (batch 8, 375x1242 + 8192 points) — one
    timed step per torch thread count after one shared warm-up step, `value` = the HOST'S BEST (the round-5
    number was 128 threads at batch 2, four times slower than 3 threads).  `threads_3` stays: the reference sets
    OMP_NUM_THREADS=3 (train20v2learn_wandb_proj.py:22).  Bounded: a thread count whose step exceeds 40 s ends the sweep
    (more threads only get slower on this step: hundreds of small eager ops); plus the per-operator times of the scalar
    oracle kernels."""
    from i2pnet_amd import modules, ops, synth
    from i2pnet_amd.train import Trainer
    from oracle import oracle
    prev = ops.set_backend(oracle.backend())
    # the reference's CPU path = eager PyTorch-CPU (multi-threaded GEMM / BN) + CPU restatements of its
    # CUDA operators; the oracle's scalar restatements of OUR fused kernels are checkers, not a baseline
    fused = (modules.USE_FUSED_MLP, modules.USE_FUSED_BN, modules.USE_FUSED_IMG)
    modules.USE_FUSED_MLP = modules.USE_FUSED_BN = modules.USE_FUSED_IMG = False
    threads0 = torch.get_num_threads()
    ncpu = os.cpu_count() or 1
    counts = sorted({min(n, ncpu) for n in thread_counts})
    table = {}
    try:
        tr = Trainer(cfg=cfg, device="cpu")
        batch = synth.make_batch(batch_size, 8192, 375, 1242, seed=0)
        torch.set_num_threads(min(16, ncpu))
        tr.step(batch)                                   # warm-up (allocator, thread pools), shared by every leg
        for n in counts:
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            tr.step(batch)
            dt = time.perf_counter() - t0
            table[str(n)] = round(batch_size / dt, 4)
            if dt > 40.0:
                break
        torch.set_num_threads(1)
        per_op = _oracle_op_times()
    finally:
        torch.set_num_threads(threads0)
        ops.set_backend(prev)
        modules.USE_FUSED_MLP, modules.USE_FUSED_BN, modules.USE_FUSED_IMG = fused
    best = max(table, key=table.get)
    return {"value": table[best], "unit": "samples/s", "cores": int(best),
            "kind": "port", "sample": f"1 training step at batch {batch_size} (configs[1]'s shapes) per thread count after one shared warm-up step; "
                                      f"best of the sweep; host cpu_count={ncpu}, torch default threads {threads0}",
            "threads_sweep": table,
            "threads_3": {"value": table.get("3"), "unit": "samples/s", "cores": 3,
                          "sample": f"the same step at batch {batch_size} with 3 torch threads (the reference's OMP_NUM_THREADS=3)"},
            "oracle_op_us_1_thread": per_op}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher.  Re-executes this
    file under `torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1, a free port), passes the
    children's output through (rank 0 prints the one JSON line) and exits with the launcher's status."""
    import subprocess
    if not args.selftest_cpu:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible: refusing to run fewer ranks "
                  f"than asked for", file=sys.stderr, flush=True)
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1) // 2)))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def _selftest_cpu(args):
    """Launcher / timing / JSON plumbing of the N > 1 path on CPU ranks (gloo), for tests/test_bench_launcher.py:
    the same Trainer step on a reduced shape with the CPU oracle as operator backend.  NOT a measurement: the
    metric is named `selftest` and the mode needs I2P_BENCH_SELFTEST=1."""
    if os.environ.get("I2P_BENCH_SELFTEST") != "1":
        print("bench.py: --selftest-cpu is a test hook (set I2P_BENCH_SELFTEST=1)", file=sys.stderr)
        sys.exit(2)
    from i2pnet_amd import ops, synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.train import Trainer, init_distributed
    from oracle import oracle
    ops.set_backend(oracle.backend())
    torch.set_num_threads(2)
    rank, local_rank, world = init_distributed("gloo")
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr); sys.exit(2)
    tr = Trainer(cfg=cfg, device="cpu", world_size=world, local_rank=local_rank)
    batch = synth.make_batch(1, 2048, 160, 512, seed=1000 + rank)
    for _ in range(args.warmup):
        tr.step(batch)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _, _ = tr.step(batch)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "selftest (CPU ranks, gloo, oracle operators: launcher plumbing only)",
                          "value": round(world * args.steps / float(t), 4), "unit": "samples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(t) / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "selftest", "parallelism": f"dp{world}",
                                                          "final_loss": round(float(loss), 4)}}), flush=True)
    if dist.is_initialized():
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (configs[1]: 8, configs[2]: 16)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 4],
                    help="BASELINE.json configs[i]: 1 = KITTI-shaped fp32 batch 8 (default, the driver's line); "
                         "2 = KITTI-shaped bf16 activations batch 16; 4 = nuScenes range image (21x1800, 16384 points) bf16")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--layout", default="scan", choices=["scan", "centre"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="after the default (--config 1) line also measure configs[2] and configs[4] in the same process and report "
                         "them under `other_configs` (1 = at N=1 only, 2 = at any N, 0 = never)")
    ap.add_argument("--graph", type=int, default=1, help="capture the step in hipGraphs (0 = eager)")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "tree"],
                    help="tree: every rank reads a synthetic KITTI odometry tree (written to a temp dir) through the real input pipeline "
                         "(i2pnet_amd.data: file reads, pinned staging, device-side sample build on a copy stream, Prefetcher) — a "
                         "loader-inclusive number on the reference loader's own shapes (160x512 crop, 150 000-row clouds); its own metric name, never `value` of the BASELINE metric")
    ap.add_argument("--loader-line", type=int, default=1, help="after the default line also time the step fed by the input pipeline "
                    "(--data tree workload) and report it as `loader_inclusive` (1 = at N=1 only: an untested failure of the loader path must never "
                    "cost a multi-GPU record; 2 = at any N; 0 = skip)")
    ap.add_argument("--loader-steps", type=int, default=60)
    ap.add_argument("--no-dp-proxy", action="store_true", help="skip the two-graph + 1-rank all-reduce proxy measurement")
    ap.add_argument("--no-pin", action="store_true", help="do not pin ranks to host-core groups")
    ap.add_argument("--no-finddb-warmup", action="store_true", help="every rank runs MIOpen's find itself")
    ap.add_argument("--selftest-cpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    dflt = _CONFIG_DEFAULTS[args.config]
    args.batch = dflt[0] if args.batch is None else args.batch
    args.points = dflt[1] if args.points is None else args.points
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(args, sys.argv[1:])
    if args.selftest_cpu:
        return _selftest_cpu(args)

    from i2pnet_amd import ops
    from i2pnet_amd.train import dist_env, init_distributed

    process_setup()
    _, lr0, w0 = dist_env()
    pinned = None if args.no_pin else _pin_cores(lr0, w0)            # before any helper thread exists (RCCL watchdog / proxy, MIOpen find, OpenMP pool inherit it)
    rank, local_rank, world = init_distributed("nccl")
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr, flush=True)
        sys.exit(2)
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no GPU (device_count={torch.cuda.device_count()})", file=sys.stderr, flush=True)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    res = _run_workload(args.config, args, rank, local_rank, world, device)
    line = None
    if rank == 0:
        bf16 = args.config in (2, 4)
        line = _json_line(res, args, world)
        line["config"]["rccl_ranks"] = dist.get_world_size() if dist.is_initialized() else 1
        line["config"]["host_cores_per_rank"] = pinned
        sol = solvers_from_find_db()
        counts = {}
        for v in sol.values():
            counts[v] = counts.get(v, 0) + 1
        line["config"]["miopen"] = {"find_mode": "cudnn.benchmark=True (miopenFind*, exhaustive)", "env": miopen_env(),
                                    "solvers": counts, "problems": len(sol)}
        # which 3x3 convolutions of the 15-block image encoder are NOT on MIOpen (the switches restore MIOpen)
        hip_first, hip_conv = os.environ.get("I2P_NO_IMG_FIRST") != "1", os.environ.get("I2P_NO_CONV16") != "1"
        line["config"]["image_encoder"] = {
            "block 1 (3->16, conv + BN + LeakyReLU + pool)": "csrc/image_first.hip" if hip_first else "MIOpen + csrc/image_block.hip",
            "blocks 2-5 convolutions (16->16 x3, 16->32; fp32 tier)": "csrc/image_conv16.hip" if hip_conv else "MIOpen",
            "blocks 6-15 convolutions": "MIOpen", "BN + LeakyReLU + MaxPool tails": "csrc/image_block.hip",
            "switches": {k: os.environ[k] for k in ("I2P_NO_IMG_FIRST", "I2P_NO_CONV16", "I2P_NO_CONV32", "I2P_NO_TAIL_BWD") if k in os.environ}}
        if args.data == "synthetic":
            prev = ops.set_precision("bf16" if bf16 else "fp32")
            line["roofline"] = kernel_rooflines_bf16(args.batch, device, N=171 if args.config == 4 else 228) if bf16 else kernel_rooflines(args.batch, device)
            ops.set_precision(prev)
    # the bf16 workloads (configs[2], configs[4]) in the same process after the default line, so that the driver's record
    # carries them too (N = 1 only: the scaling runs stay short)
    others = []
    if args.config == 1 and args.other_configs and (world == 1 or args.other_configs > 1) and args.data == "synthetic":
        for c in (2, 4):
            a2 = argparse.Namespace(**vars(args))
            a2.config, a2.batch, a2.points = c, *_CONFIG_DEFAULTS[c]
            r2 = _run_workload(c, a2, rank, local_rank, world, device)
            if rank == 0:
                l2 = _json_line(r2, a2, world)
                prev = ops.set_precision("bf16")
                l2["roofline"] = kernel_rooflines_bf16(a2.batch, device, N=171 if c == 4 else 228)
                ops.set_precision(prev)
                others.append(l2)
    # the same step fed by the real input pipeline (every rank its own synthetic KITTI tree through data.Prefetcher: the reference
    # loader's shapes, file reads + staging + device-side build inside the timed region) — its own metric, never `value`; by default at
    # N = 1 only (`--loader-line 2` or `--data tree` for N > 1: VERDICT r4 item 9).  At N = 1 a failure here is reported, not raised.
    loader_line = None
    if args.config == 1 and args.data == "synthetic" and (args.loader_line > 1 or (args.loader_line and world == 1)):
        a3 = argparse.Namespace(**vars(args))
        a3.data, a3.steps, a3.warmup = "tree", min(args.loader_steps, 200), 10
        try:
            r3 = _run_workload(1, a3, rank, local_rank, world, device)
            if rank == 0:
                l3 = _json_line(r3, a3, world)
                loader_line = {k: l3[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "data")}
                loader_line["workload"] = l3["config"]["workload"]; loader_line["per_gpu_batch"] = l3["config"]["per_gpu_batch"]
        except Exception as e:                                  # noqa: BLE001
            loader_line = {"error": "%s: %s" % (type(e).__name__, e)}
            if world > 1:
                raise                                            # (ranks must not diverge: with N > 1 a one-sided failure would hang the others)
    if rank == 0 and world == 1 and not args.no_dp_proxy and not dist.is_initialized() and args.data == "synthetic":
        # what ONE GPU can say about N > 1: the data-parallel step structure — graph A, RCCL all-reduce of the flat
        # gradient, graph B — with a 1-rank group, against the single-graph step measured above
        line["dp_proxy"] = dp_proxy(args, device, line["ms_per_step"])
    if rank == 0:
        if loader_line is not None:
            line["loader_inclusive"] = loader_line
        if others:
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline and args.data == "synthetic":
            from i2pnet_amd.config import I2PNetConfig, I2PNetConfigNuScenes
            prev = ops.set_precision("fp32")
            line["cpu_baseline"] = cpu_baseline(I2PNetConfigNuScenes if args.config == 4 else I2PNetConfig)
            ops.set_precision(prev)
    if dist.is_available() and dist.is_initialized():        # also the forced 1-rank group of tools/ddp_smoke.sh
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which a pipe buffers until exit: flush it first so that the JSON
        # line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


_CONFIG_DEFAULTS = {1: (8, 8192), 2: (16, 8192), 4: (8, 16384)}


def _pin_cores(local_rank, world):
    """One contiguous group of host cores per rank: the rank's launch thread, the RCCL proxy / watchdog
    threads and MIOpen's find threads stay off the other ranks' cores.  Groups are cut from the cores this process may
    run on (cgroup-aware); with one rank nothing is changed.  Returns the number of cores of the group (None = not pinned)."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, 8)))
        return per
    except OSError:
        return None


def _run_workload(config, args, rank, local_rank, world, device):
    """W warm-up steps, then K timed steps (barrier + synchronize on both sides, MAX over ranks) of BASELINE configs[config]."""
    from i2pnet_amd import ops, synth
    from i2pnet_amd.config import I2PNetConfig, I2PNetConfigNuScenes
    from i2pnet_amd.train import Trainer
    cfg = I2PNetConfigNuScenes if config == 4 else I2PNetConfig
    bf16 = config in (2, 4)
    prev_prec = ops.set_precision("bf16" if bf16 else "fp32")   # bf16 storage of the fused chains' activations / gradients (configs[2], [4])
    try:
        use_graph = bool(args.graph)       # N>1: graph A (fwd+bwd) -> eager RCCL all-reduce of the flat gradient -> graph B (clip+Adam)
        if world > 1:
            # MIOpen's exhaustive find runs inside the warm-up steps of capture(): let rank 0 search first and fill the
            # user find-db, the other ranks then read its records instead of 8 processes benchmarking concurrently on one host
            _find_db_warmup(cfg, args, config, rank, device)
        tr = Trainer(cfg=cfg, device=device, world_size=world, local_rank=local_rank, capturable=use_graph)
        feed = None
        if getattr(args, "data", "synthetic") == "tree":
            # the real input pipeline under the step: per rank its own tree (2 x batch frames), read again every epoch
            import itertools
            import tempfile
            from i2pnet_amd import data as D
            tmp = tempfile.mkdtemp(prefix="i2p_tree_r%d_" % rank)
            synth.write_kitti_tree(tmp, frames=2 * args.batch, seqs=(0,), seed=100 + rank)
            ds = D.KittiOdometryFiles(tmp, "train")
            builder = D.DeviceSampleBuilder(device, mode="train")

            pf = D.Prefetcher(ds, builder, args.batch)       # ONE prefetcher: its pinned staging slots and copy stream live across epochs

            feed = pf.cycle()                                # endless stream over the tree: no pipeline restart at an epoch end
            batch = next(feed)
        else:
            batch = synth.make_batch(args.batch, args.points, 375, 1242, seed=1000 + rank, device=device, layout=args.layout,
                                     beams=32 if config == 4 else 64, fup=cfg.fup, fdown=cfg.fdown)
        graph_live = tr.capture(batch) if use_graph else False
        if use_graph and not graph_live:       # an eager step is 3-4x slower: never report it as the captured number
            print("bench.py: hipGraph capture failed (see the message above); run with --graph 0 for an eager measurement",
                  file=sys.stderr, flush=True)
            sys.exit(3)

        def sync():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        nxt = (lambda: next(feed)) if feed is not None else (lambda: batch)
        for _ in range(args.warmup):
            tr.step(nxt())
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _, _ = tr.step(nxt())
        sync()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert torch.isfinite(loss).all()
        from i2pnet_amd import ops as _ops
        if _ops.chain_errors():                 # a grid barrier of the one-launch MLP chains timed out: the steps computed garbage
            print("bench.py: chain-kernel grid barrier timed out (%d) — the grid was not resident; I2P_NO_CHAIN=1 runs the layer kernels" %
                  _ops.chain_errors(), file=sys.stderr, flush=True)
            sys.exit(4)
        out = {"dt": dt, "loss": float(loss), "graph_live": graph_live}
        del tr, batch
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return out
    finally:
        ops.set_precision(prev_prec)


def dp_proxy(args, device, single_graph_ms):
    """The N > 1 step structure on one GPU: the same workload with the step split into hipGraph A (forward, loss, backward, pack) ->
    RCCL all-reduce of the flat gradient on a 1-RANK process group -> hipGraph B (average, clip, Adam), i2pnet_amd/train.py.  What it
    measures: the cost of the graph split and of launching / running the collective kernel (no xGMI traffic: one rank); what it
    cannot: link time of the 3.4 MB ring all-reduce (~40 us at 7 x 153 GB/s) and stragglers.  `implied_ceiling` =
    t1 / (t1 + overhead) = the scaling efficiency this structure allows before any link time.  No scaling curve exists (SCALE
    skipped in rounds 1-3)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    prev = os.environ.get("I2P_FORCE_DP")
    os.environ["I2P_FORCE_DP"] = "1"
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        res = _run_workload(args.config, args, 0, device.index or 0, 1, device)
        ranks = dist.get_world_size()
    finally:
        dist.destroy_process_group()
        if prev is None:
            os.environ.pop("I2P_FORCE_DP", None)
        else:
            os.environ["I2P_FORCE_DP"] = prev
    t_dp = res["dt"] / args.steps * 1e3
    over_us = (t_dp - single_graph_ms) * 1e3
    return {"two_graph_allreduce_ms_per_step": round(t_dp, 3), "single_graph_ms_per_step": round(single_graph_ms, 3),
            "dp_overhead_us_per_step": round(over_us, 1), "implied_ceiling": round(single_graph_ms / max(t_dp, single_graph_ms), 4),
            "rccl_ranks": ranks, "steps": args.steps,
            "what": "graph A -> 1-rank RCCL all-reduce of the flat gradient -> graph B vs the single captured graph, same process and workload"}


def _find_db_warmup(cfg, args, config, rank, device):
    """rank 0 runs one eager forward+backward of the image encoder's shapes (MIOpen exhaustive find -> user find-db on
    disk), the others wait at a barrier and then find the records.  Skipped with --no-finddb-warmup."""
    if args.no_finddb_warmup:
        return
    if rank == 0:
        from i2pnet_amd.model import RegNet_v2
        net = RegNet_v2(cfg=cfg).to(device).train()
        rgb = torch.rand(args.batch, 3, 375, 1242, device=device) * 255.0
        x = net.RGB_net3(net.RGB_net2(net.RGB_net1(rgb.contiguous(memory_format=torch.channels_last))))
        x.sum().backward()
        torch.cuda.synchronize()
        del net, rgb, x
        torch.cuda.empty_cache()
    dist.barrier()


def _json_line(res, args, world):
    bf16 = args.config in (2, 4)
    global_batch = args.batch * world
    dt = res["dt"]
    if getattr(args, "data", "synthetic") == "tree":
        # NOT the BASELINE metric's workload: the reference loader's own shapes, the input pipeline inside the timed region
        return {"metric": "train samples/sec, loader-inclusive (reference loader shapes)", "value": round(global_batch * args.steps / dt, 3),
                "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32",
                "data": "synthetic KITTI odometry tree on local disk (i2pnet_amd.synth.write_kitti_tree), read through i2pnet_amd.data",
                "config": {"workload": "KITTI loader shapes: 160x512 crop of the x0.5 image, 150 000-row cloud (120 000 points + padding), "
                                       + ("bf16 storage mode, " if bf16 else "fp32 ") +
                                       "forward+loss+backward+clip+Adam; file reads, point shuffle, pinned staging + H2D under the step, the batch's "
                                       "device-side build (two launches) in order with it (data.Prefetcher)",
                           "per_gpu_batch": args.batch, "global_batch": global_batch, "parallelism": f"dp{world}",
                           "hipgraph": res["graph_live"], "final_loss": round(res["loss"], 4)}}
    return {
        "metric": "train samples/sec (img+8192-pt pair)", "value": round(global_batch * args.steps / dt, 3),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": {1: "configs[1]: synthetic KITTI-shaped batch, 375x1242 RGB + %d-pt cloud (%s layout), "
                                   "fp32 forward+loss+backward+clip+Adam",
                                2: "configs[2]: synthetic KITTI-shaped batch, 375x1242 RGB + %d-pt cloud (%s layout), bf16 storage of "
                                   "the fused chains' and the image encoder's activations/gradients + bf16 MFMA point-MLP and MIOpen "
                                   "bf16 convolutions (fp32 accumulate, parameters, BN statistics fp64), forward+loss+backward+clip+Adam",
                                4: "configs[4]: synthetic nuScenes-shaped batch (21x1800 range image), 375x1242 RGB + %d-pt cloud "
                                   "(%s layout), bf16 storage (fused chains + image encoder) + bf16 MFMA point-MLP, forward+loss+backward+clip+Adam"}[args.config]
                               % (args.points, args.layout),
                   "per_gpu_batch": args.batch, "global_batch": global_batch,
                   "parallelism": f"dp{world}", "hipgraph": res["graph_live"], "final_loss": round(res["loss"], 4)},
    }


if __name__ == "__main__":
    main()
