"""ctypes loader for oracle/build/libi2p_oracle.so (the CPU restatement of the reference's CUDA
kernels).  TEST INFRASTRUCTURE ONLY: the product package never imports this module.

`backend()` returns an object with the same methods as `i2pnet_amd.ops.hip_backend()` that
operates on contiguous CPU tensors, so one test harness drives both sides.
"""
import ctypes as C
import subprocess
from pathlib import Path

from i2pnet_amd import _abi
from i2pnet_amd.ops import CBackend

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "build" / "libi2p_oracle.so"
_lib = None
_fns = {}


def build(force=False):
    src = _DIR / "i2p_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR)] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        for name in _abi.SIGNATURES:
            _fns[name] = _abi.bind(_lib, name, name + "_cpu", with_stream=False)
        _lib.i2p_opt_n_threads_cpu.argtypes = [C.c_int]
        _lib.i2p_opt_n_threads_cpu.restype = C.c_int
        _lib.i2p_project_cell_cpu.argtypes = [C.c_float] * 3 + [C.c_int] * 2 + [C.c_float] * 2 + [
            C.POINTER(C.c_int)] * 2
        _lib.i2p_project_cell_cpu.restype = None
    return _lib


def _call(name, *args, stream=0):
    load()
    rc = _fns[name](*args)
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed: {rc}")


_backend = None


def backend():
    global _backend
    if _backend is None:
        load()
        _backend = CBackend(_call, "cpu", "oracle")
    return _backend


def opt_n_threads(n):
    return load().i2p_opt_n_threads_cpu(int(n))


def project_cell(x, y, z, H, W, fup, fdown):
    r, c = C.c_int(), C.c_int()
    load().i2p_project_cell_cpu(x, y, z, H, W, fup, fdown, C.byref(r), C.byref(c))
    return r.value, c.value
