"""CPU tests of the drop-in boundary: the HIP library builds for gfx950 without a GPU, loads,
and exports every symbol include/i2p_ops.h declares; the oracle exports the `_cpu` twins."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "i2p_ops.h").read_text()
    return sorted(set(re.findall(r"^int\s+(i2p_\w+)\s*\(", text, flags=re.M)))


def test_header_declares_the_reference_surface():
    names = _declared()
    for need in ["i2p_fused_conv_select_k", "i2p_furthest_point_sampling", "i2p_gather_points",
                 "i2p_gather_points_grad", "i2p_ball_query", "i2p_group_points", "i2p_group_points_grad",
                 "i2p_three_nn", "i2p_three_interpolate", "i2p_three_interpolate_grad"]:
        assert need in names


def test_hip_library_builds_loads_and_exports_all_symbols():
    from i2pnet_amd import _abi, _lib, build
    build.build(verbose=False)
    lib = ctypes.CDLL(str(_lib.lib_path()))
    for name in _declared():
        assert hasattr(lib, name), name
    lib.i2p_abi_version.restype = ctypes.c_int
    assert lib.i2p_abi_version() == 1
    # the ctypes table covers every compute entry the header declares
    helpers = {"i2p_abi_version", "i2p_lin_bwd_grid", "i2p_pair_lin_bwd_grid"} | set(_abi.HELPERS)   # no stream argument: bound separately
    assert set(_abi.SIGNATURES) | set(_abi.DEVICE_ONLY) == set(_declared()) - helpers
    assert not set(_abi.SIGNATURES) & set(_abi.DEVICE_ONLY)
    for name in _abi.HELPERS:                                  # scratch-size helpers (some return long long)
        assert hasattr(lib, name), name


def test_oracle_exports_cpu_twins():
    from oracle import oracle
    from i2pnet_amd import _abi
    lib = oracle.load()
    # every entry of the reference's operator surface and of our fp32 operator layer has a CPU restatement; the
    # device-only entries (bf16 storage formats, deterministic-accumulation variants, the fused level-1 front end)
    # are checked against those through tolerance / equality tests instead (tests/test_bf16_gpu.py, test_ops_gpu.py)
    for name in _abi.SIGNATURES:
        assert hasattr(lib, name + "_cpu"), name


def test_product_backend_rejects_cpu_tensors():
    """No CPU fallback: the HIP backend refuses host tensors loudly."""
    import torch
    from i2pnet_amd import ops
    be = ops.hip_backend()
    x = torch.zeros(1, 2, 2, 3)
    with pytest.raises(RuntimeError):
        be.gather_rows(x.view(1, 4, 3), torch.zeros(1, 1, dtype=torch.long), torch.zeros(1, 1, dtype=torch.long),
                       2, torch.zeros(1, 1, 3))
