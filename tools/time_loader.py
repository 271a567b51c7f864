"""Where a loader-inclusive step spends its time (bench.py --data tree): per-step host time of `next(feed)` and `Trainer.step`
and the device time between consecutive steps, over a persistent Prefetcher on a synthetic KITTI tree.

    python tools/time_loader.py [copy|side|inline] [steps] [cycle]

side   = the whole device build of batch i+1 on a side stream under step i
copy   = only the host-to-device copies on the side stream, the build's kernels on the step's stream
inline = no Prefetcher stream at all: read, copy and build between two steps"""
import sys, time, tempfile, torch
sys.path.insert(0, ".")
from i2pnet_amd import synth, data as D
from i2pnet_amd.config import I2PNetConfig as cfg
from i2pnet_amd.train import Trainer
mode = sys.argv[1] if len(sys.argv) > 1 else "copy"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
CYCLE = len(sys.argv) > 3 and sys.argv[3] == "cycle"          # one endless stream instead of an iterator per epoch
B = 8
tmp = tempfile.mkdtemp()
synth.write_kitti_tree(tmp, frames=4 * B, seqs=(0,), seed=1)
ds = D.KittiOdometryFiles(tmp, "train")
dev = torch.device("cuda", 0)
builder = D.DeviceSampleBuilder(dev, mode="train")
pf = D.Prefetcher(ds, builder, B, mode=mode) if mode != "inline" else None
if pf is not None:
    pf.trace = []


def epochs():
    while True:
        if pf is not None:
            if CYCLE:
                yield from pf.cycle()
            for b in pf:
                yield b
        else:
            for s in range(0, len(ds) - B + 1, B):
                yield builder([ds[i] for i in range(s, s + B)])


SLOW = []


def watch(obj, name):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); d = time.perf_counter() - t
        if d > 0.02:
            SLOW.append((name, 1e3 * d, time.perf_counter()))
        return r
    setattr(obj, name, timed)


for nm in ("affine_f64", "resize_linear_u8"):
    watch(D, nm)
for nm in ("_upload", "_crop_rgb", "perturbation", "draw_perm"):
    watch(builder, nm)
for nm in ("randn_like", "zeros", "empty", "clamp"):
    watch(torch, nm)
for nm in ("to", "__getitem__", "__setitem__"):
    watch(torch.Tensor, nm)
feed = epochs()
tr = Trainer(cfg=cfg, device=dev, capturable=True)
batch = next(feed)
tr.capture(batch)
tn = ts = 0.0
marks = []
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
evs[0].record()
t_all = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter(); b = next(feed); t1 = time.perf_counter(); tr.step(b); t2 = time.perf_counter()
    marks.append((t0, t1))
    evs[i + 1].record()
    if i >= 8:
        tn += t1 - t0; ts += t2 - t1
    if i < 14:
        print(f"[{mode}] step {i}: next {1e3 * (t1 - t0):7.2f} ms   step() {1e3 * (t2 - t1):7.2f} ms", flush=True)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
gaps = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
print(f"[{mode}] device time between step ends: " + " ".join(f"{g:.1f}" for g in gaps))
n = steps - 8
print(f"[{mode}] {steps} steps in {1e3 * t_all:.0f} ms = {1e3 * t_all / steps:.2f} ms/step; host means over the last {n}: "
      f"next(feed) {1e3 * tn / n:.2f} ms, Trainer.step {1e3 * ts / n:.2f} ms")
tr.check_chain_errors(sync=True)
if pf is not None:
    base = marks[0][0]
    for i, (a, b) in enumerate(marks[:20]):
        print(f"main step {i}: next from {1e3 * (a - base):8.1f} to {1e3 * (b - base):8.1f}")
    for k, t0, sy, ld in pf.trace:
        if t0 >= base and (t0 - base) < marks[min(19, len(marks) - 1)][1] - base:
            print(f"reader batch {k}: starts {1e3 * (t0 - base):8.1f}  slot wait {sy:6.1f}  read + shuffle + stage {ld:6.1f}")
for name, ms, at in SLOW[:60]:
    print(f"slow call: {name:20s} {ms:7.1f} ms at {1e3 * (at - marks[0][0]):9.1f}")
