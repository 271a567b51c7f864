"""Loader for the in-tree HIP library.  There is NO fallback: if libi2p_ops.so is missing or
does not load, importing any operator raises."""
import ctypes as C
from pathlib import Path

from . import _abi

import os

# I2P_OPS_LIB: load another build of the same ABI (A/B comparisons and bisection only)
_LIB_PATH = Path(os.environ.get("I2P_OPS_LIB") or (Path(__file__).resolve().parent / "lib" / "libi2p_ops.so"))
_lib = None
_fns = {}
_helpers = {}


class I2POpsError(RuntimeError):
    pass


_ERRORS = {-1: "bad argument", -2: "kernel_size_H*kernel_size_W exceeds 150", -3: "K exceeds 150"}


def lib_path():
    return _LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise I2POpsError(
                f"{_LIB_PATH} not found: build it with `python -m i2pnet_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.i2p_abi_version.restype = C.c_int
        for name in list(_abi.SIGNATURES) + list(_abi.DEVICE_ONLY):
            if os.environ.get("I2P_OPS_LIB") and not hasattr(_lib, name):
                continue                      # older build under comparison: entries it lacks stay unbound
            _fns[name] = _abi.bind(_lib, name, name, with_stream=True)
        for name in _abi.HELPERS:
            if hasattr(_lib, name):
                _helpers[name] = _abi.bind(_lib, name, name, with_stream=False)
    return _lib


def helper(name, *args):
    """plain `int f(ints...)` helpers of the ABI (grid sizes of scratch buffers)"""
    load()
    return _helpers[name](*args)


def call(name, *args, stream=0):
    """Call C-ABI entry `name`; raises I2POpsError on a non-zero status (the reference's
    launchers exit(-1) instead, e.g. fused_conv_go.cu:259-263)."""
    load()
    rc = _fns[name](*args, C.c_void_p(stream))
    if rc != 0:
        what = _ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise I2POpsError(f"{name} failed: {what}")
