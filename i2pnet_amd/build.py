"""Builds libi2p_ops.so (HIP, gfx950) in-tree.  `python -m i2pnet_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
Every source is compiled to its own object (in parallel, re-done only when it or a header is newer)
and the objects are linked into the one library: editing one kernel file costs one compile.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
# I2P_BUILD_VARIANT=name [I2P_BUILD_DEFS="-DX -DY"]: a second build of the same ABI next to the product library
# (lib/libi2p_ops_<name>.so, objects in lib/obj_<name>) for A/B timing through I2P_OPS_LIB (i2pnet_amd/_lib.py); never loaded by default
# I2P_BUILD_VARIANT_FILES="a.hip b.hip": only these sources are compiled with the variant's defines; every other object is taken
# from the product build (lib/obj), so a one-kernel ablation build costs one compile + one link
_VARIANT = os.environ.get("I2P_BUILD_VARIANT", "")
_VARIANT_FILES = os.environ.get("I2P_BUILD_VARIANT_FILES", "").split()
LIB = PKG / "lib" / (f"libi2p_ops_{_VARIANT}.so" if _VARIANT else "libi2p_ops.so")
OBJ = PKG / "lib" / (f"obj_{_VARIANT}" if _VARIANT else "obj")
SOURCES = ["fused_conv_select_k.hip", "pointnet2_ops.hip", "projection_ops.hip", "bn_act.hip", "mlp.hip", "cv_softmax.hip",
           "image_block.hip", "image_first.hip", "image_conv16.hip", "mlp_bf16.hip", "bf16_stream.hip", "sa_group.hip", "scatter_det.hip", "mlp_wreg.hip", "mlp_wreg_fused.hip", "mlp_wreg_bf16.hip", "mlp_bwd_fused_bf16.hip", "pair_bwd_bf16.hip", "pair_fwd_bf16.hip", "gemm_tn.hip", "mlp_big.hip", "optim.hip", "glue.hip", "mlp_chain.hip", "loader_build.hip", "deferred.hip", "mlp_wreg_pair_fused.hip"]
# mlp_wreg.hip: one strip = 256 MFMAs with the rest of the wave's work slotted between them, written as ONE fully
# unrolled loop — past clang's default size limit for `#pragma unroll`
EXTRA_FLAGS = {"mlp_wreg.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
               "mlp_wreg_fused.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
               "mlp_wreg_pair_fused.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
               # img1_bwd_kernel: the SLP vectoriser pairs the unrolled elements into v_pk_fma_f32 and spills 1.9 KB per lane doing it
               "image_first.hip": ["-fno-slp-vectorize"]}
if os.environ.get("WREG_ABL"):      # diagnostic build of the ablation switches in mlp_wreg.hip
    EXTRA_FLAGS["mlp_wreg.hip"].append("-DWREG_ABL=" + os.environ["WREG_ABL"])
HEADERS = [CSRC / "common.h", PKG.parent / "include" / "i2p_ops.h"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",        # bit-exact distance order: only the explicit fmaf()s fuse
    "-munsafe-fp-atomics",      # hardware global_atomic_add_f32 where a kernel still accumulates with atomics
    "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
] + (os.environ.get("I2P_BUILD_DEFS", "").split() if _VARIANT else [])


def _sources():
    return [s for s in SOURCES if (CSRC / s).exists()]


def _stale(target, deps):
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(p.stat().st_mtime > t for p in deps)


def needs_build():
    return _stale(LIB, [CSRC / s for s in _sources()] + HEADERS + list(CSRC.glob("*.h")))


def build(force=False, verbose=True):
    if not force and not needs_build() and not (_VARIANT and _VARIANT_FILES):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OBJ.mkdir(parents=True, exist_ok=True)
    hdrs = HEADERS + list(CSRC.glob("*.h"))
    jobs = []
    main_obj = PKG / "lib" / "obj"
    objs = {}
    for s in _sources():
        if _VARIANT and _VARIANT_FILES and s not in _VARIANT_FILES:
            objs[s] = main_obj / (s + ".o")             # (the product build must be current)
            continue
        o = objs[s] = OBJ / (s + ".o")
        if force or _stale(o, [CSRC / s] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", str(CSRC / s), "-o", str(o)])

    def run(cmd):
        if verbose:
            print("[i2pnet_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=str(CSRC))

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(objs[s]) for s in _sources()] + ["-o", str(LIB)])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
