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


def level1_select_roofline(B, device, iters=50):
    """fused_conv_select_k at level 1 (64x1800 image, 3600 queries, 9x15 window, K=32).
    Algorithmic bytes per sample: image 64*1800*12 + idx_n2 3600*8 read,
    3600*32*(3*8+4) written = 1.41 MB + 3.23 MB = 4.64 MB."""
    from i2pnet_amd import ops, projectpn as P, synth
    hip = ops.hip_backend()
    raw = synth.lidar_scan(B, 8192, torch.Generator(device=device).manual_seed(0), device, layout="centre")
    img, _, _ = hip.project_seq(raw, [], 64, 1800, 2.0, -24.8)
    idx = P.get_stride_idx_cuda(B, 16, 225, 4, 8, device)
    rhw = torch.arange(135, dtype=torch.int32, device=device)
    sel = torch.zeros(3, B, 3600, 32, 1, dtype=torch.long, device=device)
    mask = torch.zeros(B, 3600, 32, 1, device=device)
    unused = torch.zeros(1, device=device)

    def launch():
        hip.fused_conv_select_k(img, img, idx, rhw, 64, 1800, 3600, 9, 15, 32, 3, 0.75, 1, 1, sel[0], sel[1], sel[2],
                                unused, unused, mask, 64, 1800)
    for _ in range(5):
        launch()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()                    # same stream the kernel is launched on (torch's current stream)
    for _ in range(iters):
        launch()
    end.record(); end.synchronize()
    avg_s = start.elapsed_time(end) / iters * 1e-3
    bytes_per_launch = B * (64 * 1800 * 12 + 3600 * 8 + 3600 * 32 * 28)
    achieved = bytes_per_launch / avg_s / 1e9
    live = float((mask.view(B, 3600, 32)[:, :, 0] > 0).float().mean())
    return {"kernel": "fcsk_kernel<9> (fused_conv_select_k, level 1)", "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "avg_kernel_us": round(avg_s * 1e6, 2), "bytes_per_launch": bytes_per_launch,
            "live_query_frac": round(live, 3)}


def cpu_baseline(cfg, batch_size=2, steps=2):
    """the same training step on the host: PyTorch-CPU model + CPU oracle operators."""
    from i2pnet_amd import ops, synth
    from i2pnet_amd.train import Trainer
    from oracle import oracle
    prev = ops.set_backend(oracle.backend())
    try:
        tr = Trainer(cfg=cfg, device="cpu")
        batch = synth.make_batch(batch_size, 8192, 375, 1242, seed=0)
        tr.step(batch)                                   # warm-up (allocator, thread pools)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(batch)
        dt = time.perf_counter() - t0
    finally:
        ops.set_backend(prev)
    return {"value": round(batch_size * steps / dt, 4), "unit": "samples/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{steps} training steps at batch {batch_size} (same shapes), host cpu_count={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (configs[1]: 8)")
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--layout", default="scan", choices=["scan", "centre"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="capture the step in one hipGraph (single-GPU runs)")
    args = ap.parse_args()

    from i2pnet_amd import synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.train import Trainer, init_distributed

    rank, local_rank, world = init_distributed("nccl")
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    use_graph = bool(args.graph) and world == 1
    tr = Trainer(cfg=cfg, device=device, world_size=world, local_rank=local_rank, capturable=use_graph)
    batch = synth.make_batch(args.batch, args.points, 375, 1242, seed=1000 + rank, device=device, layout=args.layout)
    graph_live = tr.capture(batch) if use_graph else False

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _, _ = tr.step(batch)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    assert torch.isfinite(loss).all()

    if rank == 0:
        global_batch = args.batch * world
        line = {
            "metric": "train samples/sec (img+8192-pt pair)", "value": round(global_batch * args.steps / dt, 3),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic KITTI-shaped batch, 375x1242 RGB + %d-pt cloud (%s layout), "
                                   "fp32 forward+loss+backward+clip+Adam" % (args.points, args.layout),
                       "per_gpu_batch": args.batch, "global_batch": global_batch,
                       "parallelism": f"dp{world}", "hipgraph": graph_live, "final_loss": round(float(loss), 4)},
            "roofline": level1_select_roofline(args.batch, device),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
