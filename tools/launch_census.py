"""Where do the small launches of one training step come from?  torch.profiler over one eager step:
kernel launches and device time per aten op, and per python source line (forward) / autograd node (backward).

    python tools/launch_census.py [--batch 8]
"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import synth  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.train import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    import bench
    bench.process_setup()                 # the bench's MIOpen mode (exhaustive find in the warm-up steps): the launch count of the MEASURED configuration
    tr = Trainer(cfg=cfg, device=dev)
    batch = synth.make_batch(a.batch, 8192, 375, 1242, seed=1, device=dev)
    from torch.profiler import record_function

    def wrap(name, mod):
        fwd = mod.forward

        def f(*x, **k):
            with record_function("M:" + name):
                return fwd(*x, **k)
        mod.forward = f
    for name, mod in tr.net.named_children():
        wrap(name, mod)
    fc = tr.net.LiDAR_lv1.forward_center

    def fc_wrapped(*x, **k):
        with record_function("M:LiDAR_lv1"):
            return fc(*x, **k)
    tr.net.LiDAR_lv1.forward_center = fc_wrapped
    import i2pnet_amd.warp as W
    for fn in ("mul_q", "inv_q", "warp_quat_xyz"):
        orig = getattr(W, fn)

        def g(*x, _o=orig, _n=fn, **k):
            with record_function("M:warp." + _n):
                return _o(*x, **k)
        setattr(W, fn, g)
    import i2pnet_amd.loss as L
    import i2pnet_amd.train as T
    og = L.Get_loss

    def gl(*x, **k):
        with record_function("M:Get_loss"):
            return og(*x, **k)
    T.Get_loss = gl
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(batch)
        torch.cuda.synchronize()
    evs = prof.events()
    # device kernels attributed to the innermost CPU op that launched them
    by_op = collections.defaultdict(lambda: [0, 0.0])
    by_line = collections.defaultdict(lambda: [0, 0.0])

    def owner(e):
        p = e
        while p is not None:
            if p.name.startswith("M:"):
                return p.name[2:]
            p = p.cpu_parent
        return None
    seq_owner = {}
    for e in evs:
        if e.device_type == torch.autograd.DeviceType.CPU and e.sequence_nr is not None and e.sequence_nr >= 0 \
                and "Backward" not in e.name:
            o = owner(e)
            if o is not None:
                seq_owner.setdefault(e.sequence_nr, o)
    by_mod = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    by_mod_op = collections.defaultdict(lambda: collections.Counter())
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        # only leaf ops (those whose children launched nothing) to avoid double counting
        if any(c.kernels for c in e.cpu_children):
            continue
        n = len(e.kernels); t = sum(k.duration for k in e.kernels)
        by_op[e.name][0] += n; by_op[e.name][1] += t
        line = next((s for s in (e.stack or []) if "/i2pnet_amd/" in s), None)
        if line is None:
            p = e.cpu_parent
            while p is not None and not (p.name.endswith("Backward0") or p.name.endswith("Backward") or "Backward" in p.name):
                p = p.cpu_parent
            line = "bwd:" + p.name if p is not None else "?"
        else:
            line = line.split("/i2pnet_amd/")[-1]
        by_line[line][0] += n; by_line[line][1] += t
        o = owner(e)
        if o is not None:
            by_mod[o][0] += n; by_mod[o][1] += t
            by_mod_op[o]["fwd " + e.name] += n
        else:
            p = e
            while p is not None and not (p.sequence_nr is not None and p.sequence_nr >= 0 and "Backward" in p.name):
                p = p.cpu_parent
            o = seq_owner.get(p.sequence_nr, "(model.forward glue)") if p is not None else "(other: optimizer, packing)"
            by_mod[o][2] += n; by_mod[o][3] += t
            by_mod_op[o]["bwd " + (p.name.replace("autograd::engine::evaluate_function: ", "") if p is not None else e.name)] += n
    tot_n = sum(v[0] for v in by_op.values()); tot_t = sum(v[1] for v in by_op.values())
    print(f"# launches {tot_n}  device time {tot_t / 1e3:.2f} ms")
    print("## by aten op")
    for k, (n, t) in sorted(by_op.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{n:5d} {t / 1e3:8.3f} ms  {k[:100]}")
    print("## by module: fwd launches, fwd ms, bwd launches, bwd ms")
    for k, v in sorted(by_mod.items(), key=lambda kv: -(kv[1][0] + kv[1][2])):
        print(f"{v[0]:5d} {v[1] / 1e3:8.3f}  {v[2]:5d} {v[3] / 1e3:8.3f}  {k}")
    print("## per module: launches by op / autograd node")
    for k, v in sorted(by_mod.items(), key=lambda kv: -(kv[1][0] + kv[1][2])):
        print(f"[{k}] " + ", ".join(f"{name} x{c}" for name, c in by_mod_op[k].most_common(40)))
    print("## by source line / autograd node")
    for k, (n, t) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print(f"{n:5d} {t / 1e3:8.3f} ms  {k[:140]}")


if __name__ == "__main__":
    main()
