"""Builds libi2p_ops.so (HIP, gfx950) in-tree.  `python -m i2pnet_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libi2p_ops.so"
SOURCES = ["fused_conv_select_k.hip", "pointnet2_ops.hip", "projection_ops.hip", "bn_act.hip", "mlp.hip", "cv_softmax.hip", "image_block.hip"]
HEADERS = [CSRC / "common.h", PKG.parent / "include" / "i2p_ops.h"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",        # bit-exact distance order: only the explicit fmaf()s fuse
    "-munsafe-fp-atomics",      # hardware global_atomic_add_f32 for the scatter-add backward kernels
    "-Wall", "-Wno-unused-function",
]


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [CSRC / s for s in SOURCES] + HEADERS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = [hipcc] + FLAGS + [str(CSRC / s) for s in SOURCES] + ["-o", str(LIB)]
    if verbose:
        print("[i2pnet_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
