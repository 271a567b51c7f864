"""Which python lines issue the ATen launches (copies, fills, adds, cats ...) left in one eager training step?
A TorchDispatchMode logs every non-view aten op with the innermost i2pnet_amd frame (forward code and the python backward of the
custom Functions; C++ autograd nodes run on the autograd thread without a python frame and are listed under '(autograd)').

    python tools/aten_sites.py [--batch 8] [--top 90]
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from i2pnet_amd import synth  # noqa: E402
from i2pnet_amd.config import I2PNetConfig as cfg  # noqa: E402
from i2pnet_amd.train import Trainer  # noqa: E402

VIEWS = {"view", "_unsafe_view", "reshape", "expand", "permute", "slice", "select", "t", "transpose", "unsqueeze", "squeeze",
         "split", "split_with_sizes", "detach", "alias", "as_strided", "empty", "empty_like", "empty_strided", "new_empty",
         "unbind", "narrow", "_reshape_alias", "view_as", "unfold", "chunk", "is_same_size", "sym_size", "stride", "size",
         "new_empty_strided", "lift_fresh", "_local_scalar_dense", "unsafe_split", "resize_"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            site = "(autograd)"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "/i2pnet_amd/" in fr.filename and "tools/" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            self.sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--top", type=int, default=90)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tr = Trainer(cfg=cfg, device=dev)
    batch = synth.make_batch(a.batch, 8192, 375, 1242, seed=1, device=dev)
    for _ in range(2):
        tr.step(batch)
    torch.cuda.synchronize()
    log = Log()
    with log:
        tr.step(batch)
    torch.cuda.synchronize()
    by_op = collections.Counter()
    for (name, _), c in log.sites.items():
        by_op[name] += c
    print("## by op")
    print(", ".join(f"{k} x{v}" for k, v in by_op.most_common(40)))
    print("## by op and site")
    for (name, site), c in log.sites.most_common(a.top):
        print(f"{c:4d}  {name:28s} {site}")


if __name__ == "__main__":
    main()
