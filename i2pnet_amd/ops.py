"""Tensor-level bindings of the C ABI (include/i2p_ops.h).

`CBackend` marshals torch tensors to raw pointers for a table of C functions; the product
instance (`hip_backend()`) is bound to libi2p_ops.so and only accepts contiguous CUDA/HIP
tensors, launching on torch's current stream.  The argument lists are exactly the pybind
surfaces of the reference (`fused_conv_g.cpp:15-67`, `pointnet2/src/pointnet2_api.cpp:10-24`).

`set_backend()` exists so that tests and bench.py's cpu_baseline leg can run the host logic
against the CPU oracle; nothing in this package ever installs a non-HIP backend.
"""
import ctypes as C

import contextlib
import os

import torch

from . import _abi, _lib

_F32, _I32, _I64 = torch.float32, torch.int32, torch.int64
_BF16 = torch.bfloat16
BN_REPLICAS = 32          # I2P_BN_REPLICAS in include/i2p_ops.h

# ---- activation storage precision of the fused layer chains ---------------------------------------------------------
# "fp32" (default; BASELINE configs[1], the reference's precision) or "bf16" (configs[2] / configs[4]): the pre-BN
# [rows, C] tensors of a chain and the gradients between its layers are stored as bf16 and contracted on bf16 MFMA
# (fp32 accumulate, fp64 BN statistics, fp32 parameters / BN / softmax arithmetic) — csrc/mlp_bf16.hip, bf16_stream.hip.
# Chains with fewer rows than BF16_MIN_ROWS stay fp32: they are launch-latency-bound, bf16 buys them nothing and the
# pose heads / mask predictors keep full precision.
_PRECISION = "fp32"
BF16_MIN_ROWS = 32768


def set_precision(name):
    global _PRECISION
    if name not in ("fp32", "bf16"):
        raise ValueError(name)
    prev, _PRECISION = _PRECISION, name
    return prev


def get_precision():
    return _PRECISION


def bf16_rows_ok(rows, device):
    return _PRECISION == "bf16" and torch.device(device).type == "cuda" and rows >= BF16_MIN_ROWS


class _ZeroArena:
    """One pre-zeroed byte buffer per device for the many small accumulators of a training step
    (replicated fp64 BN sums, small gradient buffers): a step clears it with ONE memset
    (`begin_step`) instead of ~150 separate fills of a few KB each, and hands out 256-byte aligned
    slices.  Slices live until the next `begin_step` on that device — i.e. for exactly one
    forward+backward; with no arena active (inference, unit tests) `zeros` is `torch.zeros`."""
    SIZE = 256 << 20
    MAX_REQUEST = 16 << 20          # (larger requests in the arena: no measurable gain in time — A/B on one box, rounds 2 and 5: 730.0 / 732.1 vs
                                    #  730.8 / 727.9 samples/s — but a dozen fill launches fewer per step)

    def __init__(self):
        self.buf = {}
        self.cursor = {}
        self.high = {}                  # bytes ever handed out per device: everything above is still zero

    def begin_step(self, device):
        device = torch.device(device)
        if device not in self.buf:
            self.buf[device] = torch.zeros(self.SIZE, dtype=torch.uint8, device=device)
            self.high[device] = 0
        else:
            self.high[device] = max(self.high[device], self.cursor.get(device) or 0)
            if self.high[device]:
                self.buf[device][:self.high[device]].zero_()      # the memset covers only what was ever used
        self.cursor[device] = 0

    def end(self, device):
        device = torch.device(device)
        if device in self.high:
            self.high[device] = max(self.high[device], self.cursor.get(device) or 0)
        self.cursor.pop(device, None)

    def zeros(self, shape, dtype, device):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        cur = self.cursor.get(device)
        numel = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            numel *= int(d)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        if cur is None or nbytes > self.MAX_REQUEST or cur + nbytes > self.SIZE or numel == 0:
            return torch.zeros(shape, dtype=dtype, device=device)
        self.cursor[device] = (cur + nbytes + 255) & ~255
        return self.buf[device][cur:cur + nbytes].view(dtype).view(shape)


_arena = _ZeroArena()
_ZERO = {}
_CHAIN_OK = {}        # (device index, rows, widths, pool_k) -> does i2p_chain_fwd take it on that device
_CHAIN_ERR = {}       # device -> (device fp32 [4] counter, pinned int32 [1] flag): timed-out grid barriers of the chain kernels (must stay 0)


def _cur_dev():
    return torch.device("cuda", torch.cuda.current_device())


class ChainBarrierTimeout(RuntimeError):
    """a grid barrier of the one-launch MLP chains (csrc/mlp_chain.hip) was abandoned: the launch's grid was not co-resident
    (CU mask, another process on the GPU, two chain launches overlapping) and its results are invalid"""


def _dev_key(device=None):
    """torch.device('cuda', index) of `device` (None / 'cuda' = the current device)"""
    d = torch.device("cuda" if device is None else device)
    if d.type != "cuda":
        return d
    return torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index)


def _register_chain_counter():
    """Error sinks of the chain kernels for the CURRENT device (i2p_chain_set_error_words; the library keeps them per device):
    a device fp32 [4] whose element 0 counts the launches with a timed-out grid barrier (four floats so that the trainer can
    append it to its 16-byte-granular flat gradient: the optimiser kernel skips a poisoned step, and the all-reduce makes
    every rank skip it) and a pinned host word the kernels set at the same moment, read without synchronising."""
    dev = _cur_dev()
    t = torch.zeros(4, dtype=torch.float32, device=dev)
    h = torch.zeros(1, dtype=torch.int32).pin_memory()
    _CHAIN_ERR[dev] = (t, h)
    with torch.cuda.device(dev):
        _lib.helper("i2p_chain_set_error_words", C.c_void_p(t.data_ptr()), C.c_void_p(h.data_ptr()))


def chain_error_words(device=None):
    """(device fp32 [4] counter, pinned host int32 [1] flag) of `device`, registering them on first use; None while a graph
    is being captured and nothing is registered yet (never allocated inside a graph's pool)"""
    dev = _dev_key(device)
    if dev not in _CHAIN_ERR:
        if torch.cuda.is_current_stream_capturing():
            return None
        with torch.cuda.device(dev):
            _register_chain_counter()
    return _CHAIN_ERR[dev]


def chain_error_flag(device=None):
    """True if a chain launch on `device` (None: any device of this process) has reported a timed-out barrier.  Reads the
    host-mapped flags only: no synchronisation, so it sees the launches that have RUN so far."""
    dev = None if device is None else _dev_key(device)
    return any(int(h[0]) != 0 for d, (t, h) in _CHAIN_ERR.items() if dev is None or d == dev)


def chain_errors(device=None):
    """number of chain launches of this process whose grid barrier timed out (synchronises); non-zero means invalid results"""
    dev = None if device is None else _dev_key(device)
    return sum(int(t[0].item()) for d, (t, h) in _CHAIN_ERR.items() if dev is None or d == dev)


def chain_errors_reset(device=None):
    dev = None if device is None else _dev_key(device)
    for d, (t, h) in _CHAIN_ERR.items():
        if dev is None or d == dev:
            t.zero_(); h.zero_()


# ---- deferred weight-gradient reductions (csrc/deferred.hip) ------------------------------------------------------------------
# Trainer: defer_begin() before the step's backward, defer_flush() after it (one launch instead of ~43), defer_end() in the finally.
# While active, every partial-slab buffer a backward allocates is kept alive here until the flush: the caching allocator would
# otherwise hand its memory to the next kernel of the same step.  I2P_NO_DEFER=1 switches the mechanism off (A/B).
_DEFER = {"on": False, "keep": [], "dws": [], "paused": 0}


def defer_begin():
    if os.environ.get("I2P_NO_DEFER") == "1" or get_backend().name != "hip":
        return False
    _lib.helper("i2p_defer_begin")
    _DEFER["on"] = True; _DEFER["keep"] = []; _DEFER["dws"] = []; _DEFER["paused"] = 0
    return True


def defer_keep(*tensors, dw=()):
    """called by the backend's backward entries with their partial / scratch buffers, and with the weight-gradient tensors whose sums
    were recorded: only their ADDRESSES are noted (a reference held here would make autograd's AccumulateGrad clone the gradient instead
    of adopting it) — Trainer checks before the flush that each of them became a parameter's `.grad` untouched (`defer_noted`)"""
    if _DEFER["on"] and not _DEFER["paused"]:
        _DEFER["keep"].extend(t for t in tensors if t is not None)
        _DEFER["dws"].extend(int(t.data_ptr()) for t in ((dw,) if isinstance(dw, torch.Tensor) else dw) if t is not None)


def defer_noted():
    return list(_DEFER["dws"])


def defer_compact(call, cout, cpad, cin):
    """`call()` -> (..., dw [cout, cpad]) is a layer backward whose input rows are zero-padded from cin to cpad columns.  With deferral
    active its slab sum is recorded in the column-compacting form and the [cout, cin] gradient is a dense view of dw's first elements
    (no slice copy, and the sum joins the step's one flush); otherwise the full sum is formed at once and sliced."""
    if _DEFER["on"] and not _DEFER["paused"]:
        _lib.helper("i2p_defer_compact_next", int(cout), int(cpad), int(cin))
        noted = len(_DEFER["dws"])
        out = call()
        if _lib.helper("i2p_defer_compact_next", 0, 0, 0):
            return out[:-1] + (out[-1].view(-1)[:cout * cin].view(cout, cin),)
        # (a reduction that cannot take the compact form — bf16 slabs, another shape — is not recorded under a pending request: it has
        #  run; its full result is sliced, and it is not one of the gradients the flush still owes)
        del _DEFER["dws"][noted:]
        return out[:-1] + (out[-1][:, :cin].contiguous(),)
    with defer_paused():
        out = call()
    return out[:-1] + (out[-1][:, :cin].contiguous(),)


class defer_paused:
    """`with ops.defer_paused():` around a backward call whose weight gradient is read at once (its reduction is launched immediately)"""
    def __enter__(self):
        if _DEFER["on"]:
            _DEFER["paused"] += 1
            _lib.helper("i2p_defer_pause", 1)

    def __exit__(self, *exc):
        if _DEFER["on"]:
            _DEFER["paused"] = max(0, _DEFER["paused"] - 1)
            _lib.helper("i2p_defer_pause", 0)


def defer_flush():
    """sum every recorded weight-gradient reduction on the current stream; -> number of reductions flushed"""
    if not _DEFER["on"]:
        return 0
    n = _lib.helper("i2p_defer_pending")
    if n:
        _lib.call("i2p_defer_flush", stream=torch.cuda.current_stream().cuda_stream)
    _DEFER["keep"] = []; _DEFER["dws"] = []
    return n


def defer_end():
    if _DEFER["on"]:
        _DEFER["on"] = False; _DEFER["keep"] = []; _DEFER["dws"] = []; _DEFER["paused"] = 0
        dropped = _lib.helper("i2p_defer_end")
        if dropped:
            raise RuntimeError(f"{dropped} deferred weight-gradient reductions were never flushed: those gradients are invalid")


def zero_scalar(device, dtype=torch.float32):
    """A permanent read-only 0-d zero on `device` (expand it where a constant zero operand is needed: padding
    channels of a cat, the real part of a pure quaternion) — a fresh `new_zeros(())` is one fill launch each time."""
    device = torch.device(device)
    key = (str(device), dtype)
    if key not in _ZERO:
        _ZERO[key] = torch.zeros((), dtype=dtype, device=device)
    return _ZERO[key]
_CHAINS_OFF = [0]


@contextlib.contextmanager
def chains_off():
    """forward passes issued inside take the layer-by-layer kernels, and so do their backward passes (the choice is made in the
    forward and kept on the autograd node).  For work that shares the GPU with another stream's kernels: a chain launch needs its
    whole grid resident (csrc/mlp_chain.hip)."""
    _CHAINS_OFF[0] += 1
    try:
        yield
    finally:
        _CHAINS_OFF[0] -= 1


begin_step = _arena.begin_step      # called by the trainer at the top of every forward+backward
end_step = _arena.end
zeros = _arena.zeros


class CBackend:
    """Calls `fns[name](*scalars_and_pointers [, stream])`."""

    def __init__(self, call, device_type, name):
        self._call = call            # call(name, *args, stream=...)
        self.device_type = device_type
        self.name = name

    # ---- marshalling ----------------------------------------------------------------------
    def _p(self, t, dtype, what):
        if not isinstance(t, torch.Tensor):
            raise RuntimeError(f"{what} must be a tensor")
        if t.device.type != self.device_type:
            # reference: TORCH_CHECK(x.type().is_cuda(), ...) — fused_conv_g.cpp:11
            raise RuntimeError(f"{what} must be a {self.device_type} tensor (got {t.device})")
        if not t.is_contiguous():
            raise RuntimeError(f"{what} must be contiguous")   # fused_conv_g.cpp:12
        if t.dtype != dtype:
            raise RuntimeError(f"{what} must be {dtype} (got {t.dtype})")
        return C.c_void_p(t.data_ptr())

    def _stream(self):
        if self.device_type == "cuda":
            return torch.cuda.current_stream().cuda_stream
        return 0

    # ---- fused_conv_select_k (fused_conv_g.cpp:15-67) ---------------------------------------
    def fused_conv_select_k(self, xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H,
                            kernel_size_W, K, flag, distance, stride_h, stride_w, select_b_idx,
                            select_h_idx, select_w_idx, valid_idx, valid_in_dis_idx, select_mask,
                            small_h, small_w):
        batch = xyz1.size(0)                                   # fused_conv_g.cpp:50
        self._call(
            "i2p_fused_conv_select_k", int(batch), int(H), int(W), int(npoints),
            int(kernel_size_H), int(kernel_size_W), int(K), int(flag), float(distance),
            int(stride_h), int(stride_w),
            self._p(xyz1, _F32, "xyz1"), self._p(xyz2, _F32, "xyz2"),
            self._p(idx_n2, _I32, "idx_n2"), self._p(random_hw, _I32, "random_hw"),
            self._p(select_b_idx, _I64, "select_b_idx"), self._p(select_h_idx, _I64, "select_h_idx"),
            self._p(select_w_idx, _I64, "select_w_idx"), self._p(valid_idx, _F32, "valid_idx"),
            self._p(valid_in_dis_idx, _F32, "valid_in_dis_idx"),
            self._p(select_mask, _F32, "select_mask"), int(small_h), int(small_w),
            stream=self._stream())

    # ---- pointnet2 (pointnet2_api.cpp:10-24) ------------------------------------------------
    def furthest_point_sampling_wrapper(self, b, n, m, points, temp, idx):
        self._call("i2p_furthest_point_sampling", int(b), int(n), int(m),
                   self._p(points, _F32, "points"), self._p(temp, _F32, "temp"),
                   self._p(idx, _I32, "idx"), stream=self._stream())
        return 1

    def gather_points_wrapper(self, b, c, n, npoints, points, idx, out):
        self._call("i2p_gather_points", int(b), int(c), int(n), int(npoints),
                   self._p(points, _F32, "points"), self._p(idx, _I32, "idx"),
                   self._p(out, _F32, "out"), stream=self._stream())
        return 1

    def gather_points_grad_wrapper(self, b, c, n, npoints, grad_out, idx, grad_points):
        self._call("i2p_gather_points_grad", int(b), int(c), int(n), int(npoints),
                   self._p(grad_out, _F32, "grad_out"), self._p(idx, _I32, "idx"),
                   self._p(grad_points, _F32, "grad_points"), stream=self._stream())
        return 1

    def ball_query_wrapper(self, b, n, m, radius, nsample, new_xyz, xyz, idx):
        self._call("i2p_ball_query", int(b), int(n), int(m), float(radius), int(nsample),
                   self._p(new_xyz, _F32, "new_xyz"), self._p(xyz, _F32, "xyz"),
                   self._p(idx, _I32, "idx"), stream=self._stream())
        return 1

    def group_points_wrapper(self, b, c, n, npoints, nsample, points, idx, out):
        self._call("i2p_group_points", int(b), int(c), int(n), int(npoints), int(nsample),
                   self._p(points, _F32, "points"), self._p(idx, _I32, "idx"),
                   self._p(out, _F32, "out"), stream=self._stream())
        return 1

    def group_points_grad_wrapper(self, b, c, n, npoints, nsample, grad_out, idx, grad_points):
        self._call("i2p_group_points_grad", int(b), int(c), int(n), int(npoints), int(nsample),
                   self._p(grad_out, _F32, "grad_out"), self._p(idx, _I32, "idx"),
                   self._p(grad_points, _F32, "grad_points"), stream=self._stream())
        return 1

    def three_nn_wrapper(self, b, n, m, unknown, known, dist2, idx):
        self._call("i2p_three_nn", int(b), int(n), int(m), self._p(unknown, _F32, "unknown"),
                   self._p(known, _F32, "known"), self._p(dist2, _F32, "dist2"),
                   self._p(idx, _I32, "idx"), stream=self._stream())

    def three_interpolate_wrapper(self, b, c, m, n, points, idx, weight, out):
        self._call("i2p_three_interpolate", int(b), int(c), int(m), int(n),
                   self._p(points, _F32, "points"), self._p(idx, _I32, "idx"),
                   self._p(weight, _F32, "weight"), self._p(out, _F32, "out"),
                   stream=self._stream())

    def three_interpolate_grad_wrapper(self, b, c, n, m, grad_out, idx, weight, grad_points):
        self._call("i2p_three_interpolate_grad", int(b), int(c), int(n), int(m),
                   self._p(grad_out, _F32, "grad_out"), self._p(idx, _I32, "idx"),
                   self._p(weight, _F32, "weight"), self._p(grad_points, _F32, "grad_points"),
                   stream=self._stream())

    # ---- operators that are eager PyTorch in the reference ------------------------------------
    def project_seq(self, xyz, feats, H, W, fup, fdown):
        """xyz [B,N,3], feats list of [B,N,D] -> (xyz_img [B,H,W,3], [feat_img [B,H,W,D]], winner [B,H*W])."""
        B, N, _ = xyz.shape
        dev = xyz.device
        out_xyz = torch.empty(B, H, W, 3, dtype=_F32, device=dev)
        outs = [torch.empty(B, H, W, f.shape[-1], dtype=_F32, device=dev) for f in feats]
        winner = torch.empty(B, H * W, dtype=_I32, device=dev)
        for f in feats:
            self._p(f, _F32, "feature")
        dims = (C.c_int * max(len(feats), 1))(*[int(f.shape[-1]) for f in feats])
        src = _abi.ptr_array(feats)
        dst = _abi.ptr_array(outs)
        self._call("i2p_project_seq", int(B), int(N), int(H), int(W), float(fup), float(fdown),
                   self._p(xyz, _F32, "xyz"), len(feats), C.cast(src, C.c_void_p),
                   C.cast(dims, C.c_void_p), self._p(out_xyz, _F32, "out_xyz"),
                   C.cast(dst, C.c_void_p), self._p(winner, _I32, "cell_winner"),
                   stream=self._stream())
        return out_xyz, outs, winner

    def sa_l1_group(self, sel_img, raw_img, out_h, out_w, stride_h, stride_w, kH, kW, K, distance):
        """Level-1 grouping in one launch (csrc/sa_group.hip): window K-NN around the strided cells of `sel_img`
        [B,H,W,3] + the 10-channel geometric feature rows built from `raw_img` -> [B, out_h*out_w, K, 12]."""
        B, H, W, _ = sel_img.shape
        feat = torch.empty(B, out_h * out_w, K, 12, dtype=_F32, device=sel_img.device)
        self._call("i2p_sa_l1_group", int(B), int(H), int(W), int(out_h), int(out_w), int(stride_h), int(stride_w), int(kH), int(kW),
                   int(K), float(distance), self._p(sel_img, _F32, "sel_xyz"), self._p(raw_img, _F32, "raw_xyz"),
                   self._p(feat, _F32, "feat"), stream=self._stream())
        return feat

    def sa_rows(self, xyz, centre, feat, h_idx, w_idx, K, W, cpad, xyz_col, feat_col):
        """grouped MLP input rows in one launch (csrc/sa_group.hip i2p_sa_rows): xyz [B,HW,3], centre [B,N,3], feat [B,HW,C],
        h_idx/w_idx [B,N*K] i64 -> [B, N*K, cpad]"""
        B, HW, _ = xyz.shape
        N, Cc = centre.shape[1], feat.shape[2]
        out = torch.empty(B, N * K, cpad, dtype=_F32, device=xyz.device)
        self._call("i2p_sa_rows", int(B), int(HW), int(N), int(K), int(W), int(Cc), int(cpad), int(xyz_col), int(feat_col),
                   self._p(xyz, _F32, "xyz"), self._p(centre, _F32, "centre"), self._p(feat, _F32, "feat"), self._p(h_idx, _I64, "h_idx"),
                   self._p(w_idx, _I64, "w_idx"), self._p(out, _F32, "out"), stream=self._stream())
        return out

    def gemm_tn(self, a, b):
        """a [rows, m], b [rows, n] contiguous -> a^T b [m, n], rows cut over the grid (csrc/gemm_tn.hip: the weight
        gradient of a plain linear layer)"""
        rows, m = a.shape
        n = b.shape[1]
        if b.shape[0] != rows:
            raise ValueError("gemm_tn: operands must have the same number of rows")
        nbytes = _lib.helper("i2p_gemm_tn_scratch", int(rows), int(m), int(n))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        out = torch.empty(m, n, dtype=_F32, device=a.device)
        self._call("i2p_gemm_tn", int(rows), int(m), int(n), self._p(a, _F32, "a"), int(m),
                   self._p(b, _F32, "b"), int(n), self._p(scratch, torch.uint8, "scratch"),
                   self._p(out, _F32, "out"), stream=self._stream())
        defer_keep(scratch, dw=out)
        return out

    def knn_rows(self, xyz, pix_xyz, pts, pix, idx, K, cpad):
        """xyz [B,N,3], pix_xyz [B,M,3], pts [B,N,C], pix [B,M,C], idx i64 [B,N*K] -> [B, N*K, cpad] =
        [xyz, pix_xyz[idx], pts * pix[idx], 0...] (csrc/sa_group.hip i2p_knn_rows_fwd)"""
        B, N, C = pts.shape
        M = pix.shape[1]
        out = torch.empty(B, N * K, cpad, dtype=_F32, device=pts.device)
        self._call("i2p_knn_rows_fwd", int(B), int(N), int(M), int(K), int(C), int(cpad), self._p(xyz, _F32, "xyz"),
                   self._p(pix_xyz, _F32, "pix_xyz"), self._p(pts, _F32, "pts"), self._p(pix, _F32, "pix"), self._p(idx, _I64, "idx"),
                   self._p(out, _F32, "out"), stream=self._stream())
        return out

    def knn_rows_backward(self, g, pts, pix, idx, K, need_xyz):
        """-> (d_xyz [B,N,3] or None, d_pts [B,N,C], gq [B,N*K,C])"""
        B, N, C = pts.shape
        M = pix.shape[1]
        cpad = g.shape[2]
        d_xyz = torch.empty(B, N, 3, dtype=_F32, device=g.device) if need_xyz else None
        d_pts = torch.empty(B, N, C, dtype=_F32, device=g.device)
        gq = torch.empty(B, N * K, C, dtype=_F32, device=g.device)
        self._call("i2p_knn_rows_bwd", int(B), int(N), int(M), int(K), int(C), int(cpad), self._p(g, _F32, "g"), self._p(pts, _F32, "pts"),
                   self._p(pix, _F32, "pix"), self._p(idx, _I64, "idx"), self._p(d_xyz, _F32, "d_xyz") if need_xyz else None,
                   self._p(d_pts, _F32, "d_pts"), self._p(gq, _F32, "gq"), stream=self._stream())
        return d_xyz, d_pts, gq

    def gather_rows_grad_ld(self, grad_out, ld, off, h_idx, w_idx, W, grad_feat):
        """gather_rows_grad from columns [off, off+C) of grad_out rows of pitch ld (fixed-point path, device only)"""
        B, HW, Cc = grad_feat.shape
        Q = h_idx.shape[1]
        nbytes = _lib.helper("i2p_gather_rows_grad_fx_scratch", int(B), int(HW), int(Cc))
        scratch = zeros(nbytes, torch.uint8, grad_feat.device)
        self._call("i2p_gather_rows_grad_fx_ld", int(B), int(HW), int(Cc), int(Q), int(W), self._p(grad_out, _F32, "grad_out"), int(ld), int(off),
                   self._p(h_idx, _I64, "h_idx"), self._p(w_idx, _I64, "w_idx"), self._p(scratch, torch.uint8, "scratch"),
                   self._p(grad_feat, _F32, "grad_feat"), stream=self._stream())

    def gather_rows(self, feat, h_idx, w_idx, W, out):
        B, HW, Cc = feat.shape
        Q = h_idx.shape[1]
        self._call("i2p_gather_rows", int(B), int(HW), int(Cc), int(Q), int(W),
                   self._p(feat, _F32, "feat"), self._p(h_idx, _I64, "h_idx"),
                   self._p(w_idx, _I64, "w_idx"), self._p(out, _F32, "out"), stream=self._stream())

    def gather_rows_grad(self, grad_out, h_idx, w_idx, W, grad_feat):
        B, HW, Cc = grad_feat.shape
        Q = h_idx.shape[1]
        if self.device_type == "cuda":
            # fixed-point accumulation on int64 atomics: order-independent => bitwise reproducible (scratch zeroed here)
            nbytes = _lib.helper("i2p_gather_rows_grad_fx_scratch", int(B), int(HW), int(Cc))
            scratch = zeros(nbytes, torch.uint8, grad_feat.device)
            self._call("i2p_gather_rows_grad_fx", int(B), int(HW), int(Cc), int(Q), int(W), self._p(grad_out, _F32, "grad_out"),
                       self._p(h_idx, _I64, "h_idx"), self._p(w_idx, _I64, "w_idx"), self._p(scratch, torch.uint8, "scratch"),
                       self._p(grad_feat, _F32, "grad_feat"), stream=self._stream())
            return
        self._call("i2p_gather_rows_grad", int(B), int(HW), int(Cc), int(Q), int(W),
                   self._p(grad_out, _F32, "grad_out"), self._p(h_idx, _I64, "h_idx"),
                   self._p(w_idx, _I64, "w_idx"), self._p(grad_feat, _F32, "grad_feat"),
                   stream=self._stream())

    def knn(self, xyz, new_xyz, k, idx):
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        self._call("i2p_knn", int(B), int(N), int(S), int(k), self._p(xyz, _F32, "xyz"),
                   self._p(new_xyz, _F32, "new_xyz"), self._p(idx, _I32, "idx"),
                   stream=self._stream())


    def quat_mul(self, a, b, conj_a=False, conj_b=False):
        """a [B,Na,4] (x) b [B,Nb,4], Na/Nb in {1,N} -> [B,N,4] (warp_utils.py:25-55)"""
        B, na, _ = a.shape
        nb = b.shape[1]
        out = torch.empty(B, max(na, nb), 4, dtype=_F32, device=a.device)
        self._call("i2p_quat_mul", int(B), int(na), int(nb), int(bool(conj_a)), int(bool(conj_b)), self._p(a, _F32, "a"),
                   self._p(b, _F32, "b"), self._p(out, _F32, "out"), stream=self._stream())
        return out

    def quat_unit_forward(self, mode, q):
        """mode 0: conj(q)/(|q|^2+1e-10) (warp_utils.py:10-22); mode 1: q/(sqrt(|q|^2+1e-10)+1e-10)
        (PPBackbone_center.py:562).  q [...,4] -> same shape"""
        out = torch.empty_like(q)
        self._call("i2p_quat_unit_fwd", int(mode), int(q.numel() // 4), self._p(q, _F32, "q"), self._p(out, _F32, "out"),
                   stream=self._stream())
        return out

    def quat_unit_backward(self, mode, q, g):
        dq = torch.empty_like(q)
        self._call("i2p_quat_unit_bwd", int(mode), int(q.numel() // 4), self._p(q, _F32, "q"), self._p(g, _F32, "g"),
                   self._p(dq, _F32, "dq"), stream=self._stream())
        return dq

    def row_unitvar_forward(self, x):
        """x [rows,c] -> (y, stat [rows,2]) (PPBackbone_center.py:388-393)"""
        rows, c = x.shape
        y = torch.empty_like(x)
        stat = torch.empty(rows, 2, dtype=_F32, device=x.device)
        self._call("i2p_row_unitvar_fwd", int(rows), int(c), self._p(x, _F32, "x"), self._p(y, _F32, "y"),
                   self._p(stat, _F32, "stat"), stream=self._stream())
        return y, stat

    def row_unitvar_backward(self, gy, y, stat):
        rows, c = y.shape
        gx = torch.empty_like(y)
        self._call("i2p_row_unitvar_bwd", int(rows), int(c), self._p(gy, _F32, "gy"), self._p(y, _F32, "y"),
                   self._p(stat, _F32, "stat"), self._p(gx, _F32, "gx"), stream=self._stream())
        return gx

    # ---- image-encoder block tail: BN(batch stats) + LeakyReLU + MaxPool3 (basicConv.py:13-17) -----------
    def img_bn_pool_forward(self, y, gamma, beta, eps, slope, stride, momentum=0.0, conv_bias=None,
                            running_mean=None, running_var=None):
        """y [B,H,W,C] (NHWC-contiguous conv output) -> (out [B,Ho,Wo,C], arg u8 [B,Ho,Wo,C], mean_invstd [2C]);
        updates the running buffers in place when given."""
        B, H, W, Cc = y.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = torch.empty(B, Ho, Wo, Cc, dtype=_F32, device=y.device)
        arg = torch.empty(B, Ho, Wo, Cc, dtype=torch.uint8, device=y.device)
        mean_invstd = torch.empty(2 * Cc, dtype=_F32, device=y.device)
        opt = lambda t, what: self._p(t, _F32, what) if t is not None else None
        sums = self.bn_stats(y.view(B * H * W, Cc))
        self._call("i2p_img_bn_pool_fwd", int(B), int(H), int(W), int(Cc), int(stride), self._p(y, _F32, "y"),
                   self._p(sums, torch.float64, "sums"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(eps), float(slope), float(momentum), opt(conv_bias, "conv_bias"),
                   opt(running_mean, "running_mean"), opt(running_var, "running_var"), self._p(out, _F32, "out"),
                   self._p(arg, torch.uint8, "arg"), self._p(mean_invstd, _F32, "mean_invstd"), stream=self._stream())
        return out, arg, mean_invstd

    def img_bn_pool_backward(self, gout, arg, y, mean_invstd, gamma, beta, slope, stride):
        """-> (dy [B,H,W,C], dgamma [C], dbeta [C])"""
        B, H, W, Cc = y.shape
        dy = torch.empty_like(y)
        dgamma = torch.empty(Cc, dtype=_F32, device=y.device)
        dbeta = torch.empty(Cc, dtype=_F32, device=y.device)
        dsums = zeros(BN_REPLICAS * 2 * Cc, torch.float64, y.device)
        self._call("i2p_img_bn_pool_bwd", int(B), int(H), int(W), int(Cc), int(stride), self._p(gout, _F32, "gout"),
                   self._p(arg, torch.uint8, "arg"), self._p(y, _F32, "y"), self._p(mean_invstd, _F32, "mean_invstd"),
                   self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(slope),
                   self._p(dsums, torch.float64, "dsums"), self._p(dy, _F32, "dy"), self._p(dgamma, _F32, "dgamma"),
                   self._p(dbeta, _F32, "dbeta"), stream=self._stream())
        return dy, dgamma, dbeta

    @staticmethod
    def _conv_pair(weight):
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        if tuple(weight.shape[2:]) != (3, 3) or (cin, cout) not in ((16, 16), (16, 32)):
            raise RuntimeError(f"img_conv: weight must be [16|32,16,3,3] (got {tuple(weight.shape)})")
        return cin, cout

    def img_conv16(self, x, weight, with_sums=False, input_grad=False):
        """3x3 convolution (padding 1, no bias) of x [B,H,W,cin] NHWC with weight [cout,cin,3,3] (any dense layout; (cin, cout) =
        (16, 16) or (16, 32)) on csrc/image_conv16.hip -> y [B,H,W,cout]; fp32 tensors, or bf16 tensors with a bf16 weight (bf16
        storage mode: bf16 MFMA, fp32 accumulation); with_sums: also the replicated fp64 {sum y, sum y^2} for
        `img_block_forward(sums=...)`; input_grad: x is dL/dy [B,H,W,cout] and the result dL/dx [B,H,W,cin] of that convolution."""
        cin, cout = self._conv_pair(weight)
        B, H, W, Cc = x.shape
        if Cc != (cout if input_grad else cin):
            raise RuntimeError(f"img_conv: x has {Cc} channels, weight {tuple(weight.shape)}")
        dt = x.dtype
        if dt not in (_F32, _BF16) or weight.device.type != self.device_type or weight.dtype != dt:
            raise RuntimeError(f"img_conv: x and weight must both be torch.float32 or torch.bfloat16 on the device (got {dt}, {weight.dtype})")
        bf = int(dt == _BF16)
        ws = (C.c_int * 4)(*[int(v) for v in weight.stride()])
        y = torch.empty(B, H, W, cin if input_grad else cout, dtype=dt, device=x.device)
        if input_grad:
            self._call("i2p_img_conv_bwd_data", int(B), int(H), int(W), cin, cout, bf, self._p(x, dt, "dy"), C.c_void_p(weight.data_ptr()), ws,
                       self._p(y, dt, "dx"), stream=self._stream())
            return y
        sums = zeros(BN_REPLICAS * 2 * cout, torch.float64, x.device) if with_sums else None
        self._call("i2p_img_conv_fwd", int(B), int(H), int(W), cin, cout, bf, self._p(x, dt, "x"), C.c_void_p(weight.data_ptr()), ws,
                   self._p(y, dt, "y"), self._p(sums, torch.float64, "sums") if with_sums else None, stream=self._stream())
        return (y, sums) if with_sums else y

    def img_conv16_wgrad(self, x, dy, weight):
        """dW of the same convolution from x [B,H,W,cin], dy [B,H,W,cout] (fp32, or bf16 with a bf16 weight), in `weight`'s layout and
        dtype (csrc/image_conv16.hip)"""
        cin, cout = self._conv_pair(weight)
        B, H, W, Cc = x.shape
        dt = x.dtype
        if Cc != cin or tuple(dy.shape) != (B, H, W, cout) or dy.dtype != dt or weight.dtype != dt or dt not in (_F32, _BF16):
            raise RuntimeError(f"img_conv16_wgrad: x {tuple(x.shape)} {dt}, dy {tuple(dy.shape)} {dy.dtype} do not match weight "
                               f"{tuple(weight.shape)} {weight.dtype}")
        dW = torch.empty_like(weight)
        if dW.stride() != weight.stride():
            raise RuntimeError("weight must be dense (contiguous or channels_last)")
        ws = (C.c_int * 4)(*[int(v) for v in weight.stride()])
        rows = _lib.helper("i2p_img_conv_wgrad_rows", int(B), int(H), int(W))
        partials = torch.empty(max(rows, 1) * 2304 * (cout // 16), dtype=_F32, device=x.device)
        self._call("i2p_img_conv_wgrad", int(B), int(H), int(W), cin, cout, int(dt == _BF16), self._p(x, dt, "x"), self._p(dy, dt, "dy"), ws,
                   self._p(partials, _F32, "partials"), C.c_void_p(dW.data_ptr()), stream=self._stream())
        defer_keep(partials, dw=dW)
        return dW

    def img_block_forward(self, y, gamma, beta, eps, slope, stride, momentum=0.0, conv_bias=None, running_mean=None,
                          running_var=None, out_bf16=False, sums=None):
        """second generation of `img_bn_pool_forward` (device library only): y [B,H,W,C] fp32 or bf16 -> (out fp32 / bf16, arg u8,
        mean_invstd [2C]); statistics + pooling in two launches, the coefficients formed in the pooling kernel's prologue."""
        B, H, W, Cc = y.shape
        if y.dtype not in (_F32, _BF16):
            raise RuntimeError(f"y must be torch.float32 or torch.bfloat16 (got {y.dtype})")
        ybf = y.dtype == _BF16
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        odt = _BF16 if out_bf16 else _F32
        out = torch.empty(B, Ho, Wo, Cc, dtype=odt, device=y.device)
        arg = torch.empty(B, Ho, Wo, Cc, dtype=torch.uint8, device=y.device)
        mean_invstd = torch.empty(2 * Cc, dtype=_F32, device=y.device)
        entry = "i2p_img_block_fwd" if sums is None else "i2p_img_block_pool"       # `sums` = the producer of y accumulated them
        if sums is None:
            sums = zeros(BN_REPLICAS * 2 * Cc, torch.float64, y.device)
        opt = lambda t, what: self._p(t, _F32, what) if t is not None else None
        self._call(entry, int(B), int(H), int(W), int(Cc), int(stride), int(ybf), int(bool(out_bf16)),
                   self._p(y, y.dtype, "y"), self._p(sums, torch.float64, "sums"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(eps), float(slope), float(momentum), opt(conv_bias, "conv_bias"), opt(running_mean, "running_mean"),
                   opt(running_var, "running_var"), self._p(out, odt, "out"), self._p(arg, torch.uint8, "arg"),
                   self._p(mean_invstd, _F32, "mean_invstd"), stream=self._stream())
        return out, arg, mean_invstd

    def img_conv16_tail_backward(self, g, arg, y, mean_invstd, gamma, beta, slope, weight):
        """backward of a fp32 16 -> 16 block with a stride-1 MaxPool from its incoming gradient g [B,H,W,16] ->
        (dy [B,H,W,16], dx [B,H,W,16], dgamma, dbeta): the block tail's statistics pass, then ONE kernel for the un-pooling, the
        BatchNorm backward and the convolution's input gradient (csrc/image_conv16.hip)"""
        B, H, W, Cc = y.shape
        if Cc != 16 or tuple(g.shape) != (B, H, W, 16) or tuple(weight.shape) != (16, 16, 3, 3) or g.dtype != _F32 or y.dtype != _F32:
            raise RuntimeError("img_conv16_tail_backward: fp32 16 -> 16 blocks with a stride-1 pool only")
        dev = y.device
        dsums = zeros(BN_REPLICAS * 32, torch.float64, dev)
        self._call("i2p_img_block_bwd_stats", int(B), int(H), int(W), 16, 1, 0, 0, self._p(g, _F32, "gout"), self._p(arg, torch.uint8, "arg"),
                   self._p(y, _F32, "y"), self._p(mean_invstd, _F32, "mean_invstd"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(slope), self._p(dsums, torch.float64, "dsums"), stream=self._stream())
        dy, dx = torch.empty_like(y), torch.empty_like(y)
        dgamma, dbeta = torch.empty(16, dtype=_F32, device=dev), torch.empty(16, dtype=_F32, device=dev)
        ws = (C.c_int * 4)(*[int(v) for v in weight.stride()])
        self._call("i2p_img_conv_tail_bwd", int(B), int(H), int(W), self._p(g, _F32, "g"), self._p(arg, torch.uint8, "arg"), self._p(y, _F32, "y"),
                   self._p(mean_invstd, _F32, "mean_invstd"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(slope),
                   self._p(dsums, torch.float64, "dsums"), C.c_void_p(weight.data_ptr()), ws, self._p(dy, _F32, "dy"), self._p(dx, _F32, "dx"),
                   self._p(dgamma, _F32, "dgamma"), self._p(dbeta, _F32, "dbeta"), stream=self._stream())
        return dy, dx, dgamma, dbeta

    def img_block_backward(self, gout, arg, y, mean_invstd, gamma, beta, slope, stride, dsums=None):
        """-> (dy [B,H,W,C] in y's storage type, dgamma [C], dbeta [C]); gout fp32 or bf16; `dsums` = BatchNorm-backward replica sums
        taken elsewhere already (i2p_img_block_bwd_stats): only the dy launch"""
        B, H, W, Cc = y.shape
        dy = torch.empty_like(y)
        dgamma = torch.empty(Cc, dtype=_F32, device=y.device)
        dbeta = torch.empty(Cc, dtype=_F32, device=y.device)
        entry = "i2p_img_block_bwd" if dsums is None else "i2p_img_block_bwd_dx"
        if dsums is None:
            dsums = zeros(BN_REPLICAS * 2 * Cc, torch.float64, y.device)
        self._call(entry, int(B), int(H), int(W), int(Cc), int(stride), int(y.dtype == _BF16), int(gout.dtype == _BF16),
                   self._p(gout, gout.dtype, "gout"), self._p(arg, torch.uint8, "arg"), self._p(y, y.dtype, "y"),
                   self._p(mean_invstd, _F32, "mean_invstd"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(slope),
                   self._p(dsums, torch.float64, "dsums"), self._p(dy, y.dtype, "dy"), self._p(dgamma, _F32, "dgamma"),
                   self._p(dbeta, _F32, "dbeta"), stream=self._stream())
        return dy, dgamma, dbeta

    # ---- first block of the image encoder without its conv output (csrc/image_first.hip) ----------
    def _xview(self, x):
        """x [B,3,H,W] fp32 by strides (NCHW or channels_last storage) -> (pointer, sb, sc, sh, sw)"""
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("x must be a [B,3,H,W] tensor")
        if x.device.type != self.device_type:
            raise RuntimeError(f"x must be a {self.device_type} tensor (got {x.device})")
        if x.dtype != _F32:
            raise RuntimeError(f"x must be torch.float32 (got {x.dtype})")
        return (C.c_void_p(x.data_ptr()),) + tuple(int(v) for v in x.stride())

    def _wview(self, weight):
        """weight [16,3,3,3] fp32 by strides (contiguous or channels_last storage) -> (pointer, host int[4] of element strides)"""
        if not isinstance(weight, torch.Tensor) or tuple(weight.shape) != (16, 3, 3, 3):
            raise RuntimeError(f"weight must be a [16,3,3,3] tensor (got {tuple(getattr(weight, 'shape', ()))})")
        if weight.device.type != self.device_type or weight.dtype != _F32:
            raise RuntimeError(f"weight must be a {self.device_type} torch.float32 tensor (got {weight.device}, {weight.dtype})")
        return C.c_void_p(weight.data_ptr()), (C.c_int * 4)(*[int(v) for v in weight.stride()])

    def img_first_stats(self, x, weight, eps, momentum=0.0, conv_bias=None, running_mean=None, running_var=None):
        """the first block's batch statistics from the Gram matrix of the input windows -> (mean_invstd f32 [32], gram_red f64 [1024]);
        updates the running buffers.  Depends on the images and the conv weights only: may be issued ahead of the block, on any stream."""
        B, _, H, W = x.shape
        if x.stride(3) != 1:                 # rows are read with vector loads: W must be the unit-stride dimension
            x = x.contiguous()
        xp, (wp, ws) = self._xview(x), self._wview(weight)
        mean_invstd = torch.empty(32, dtype=_F32, device=x.device)
        gram = zeros(BN_REPLICAS * 1024, torch.float64, x.device)
        gram_red = torch.empty(1024, dtype=torch.float64, device=x.device)
        opt = lambda t, what: self._p(t, _F32, what) if t is not None else None
        self._call("i2p_img_first_fwd", int(B), int(H), int(W), 2, *xp, wp, ws, None, None, float(eps), 0.0, float(momentum),
                   opt(conv_bias, "conv_bias"), opt(running_mean, "running_mean"), opt(running_var, "running_var"),
                   self._p(gram, torch.float64, "gram"), self._p(gram_red, torch.float64, "gram_red"), 0, None, None,
                   self._p(mean_invstd, _F32, "mean_invstd"), 1, stream=self._stream())
        return mean_invstd, gram_red

    def img_first_forward(self, x, weight, gamma, beta, eps, slope, stride, momentum=0.0, conv_bias=None, running_mean=None,
                          running_var=None, out_bf16=False, stats=None):
        """Conv2d(3,16,3,padding=1, no bias in y) + BN(batch statistics) + LeakyReLU + MaxPool2d(3,stride,1) of x [B,3,H,W] ->
        (out [B,Ho,Wo,16] fp32 / bf16, arg u8, mean_invstd [32], gram_red f64 [1024] for the backward); three launches, the conv
        output is never written (src/modules/basicConv.py:6-20, first block).  `stats` = what `img_first_stats` returned for the same
        x and weight: only the output kernel runs."""
        B, _, H, W = x.shape
        if x.stride(3) != 1:
            x = x.contiguous()
        xp, (wp, ws) = self._xview(x), self._wview(weight)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        odt = _BF16 if out_bf16 else _F32
        out = torch.empty(B, Ho, Wo, 16, dtype=odt, device=x.device)
        arg = torch.empty(B, Ho, Wo, 16, dtype=torch.uint8, device=x.device)
        opt = lambda t, what: self._p(t, _F32, what) if t is not None else None
        if stats is None:
            mean_invstd = torch.empty(32, dtype=_F32, device=x.device)
            gram = zeros(BN_REPLICAS * 1024, torch.float64, x.device)
            gram_red = torch.empty(1024, dtype=torch.float64, device=x.device)
            gp, grp, parts = self._p(gram, torch.float64, "gram"), self._p(gram_red, torch.float64, "gram_red"), 3
        else:
            mean_invstd, gram_red = stats
            gp, grp, parts = None, None, 2
        self._call("i2p_img_first_fwd", int(B), int(H), int(W), int(stride), *xp, wp, ws,
                   self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(eps), float(slope), float(momentum),
                   opt(conv_bias, "conv_bias"), opt(running_mean, "running_mean"), opt(running_var, "running_var"),
                   gp, grp, int(bool(out_bf16)), self._p(out, odt, "out"), self._p(arg, torch.uint8, "arg"),
                   self._p(mean_invstd, _F32, "mean_invstd"), parts, stream=self._stream())
        return out, arg, mean_invstd, gram_red

    def img_first_backward(self, gout, arg, x, weight, gamma, beta, slope, stride, mean_invstd, gram_red):
        """-> (dW [16,3,3,3] in weight's layout, dgamma [16], dbeta [16]); gout [B,Ho,Wo,16] fp32 or bf16"""
        B, _, H, W = x.shape
        if x.stride(3) != 1:
            x = x.contiguous()
        xp, (wp, ws) = self._xview(x), self._wview(weight)
        dev = x.device
        rows = _lib.helper("i2p_img_first_bwd_rows", int(B), int(H), int(W), int(stride))
        partials = torch.empty(max(rows, 1) * 16 * 29, dtype=_F32, device=dev)
        dW = torch.empty_like(weight)
        if dW.stride() != weight.stride():
            raise RuntimeError("weight must be dense (contiguous or channels_last)")
        dgamma = torch.empty(16, dtype=_F32, device=dev)
        dbeta = torch.empty(16, dtype=_F32, device=dev)
        self._call("i2p_img_first_bwd", int(B), int(H), int(W), int(stride), *xp, wp, ws,
                   self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(slope), self._p(mean_invstd, _F32, "mean_invstd"),
                   self._p(gram_red, torch.float64, "gram_red"), int(gout.dtype == _BF16), self._p(gout, gout.dtype, "gout"),
                   self._p(arg, torch.uint8, "arg"), self._p(partials, _F32, "partials"), C.c_void_p(dW.data_ptr()),
                   self._p(dgamma, _F32, "dgamma"), self._p(dbeta, _F32, "dbeta"), stream=self._stream())
        return dW, dgamma, dbeta

    # ---- batch-stat BatchNorm + activation (PPBackbone_center.py:28-46) ---------------------------
    def bn_act_forward(self, y, gamma, beta, eps, slope):
        """y [rows,c] f32 -> (out [rows,c], mean_invstd [2c]) with batch statistics.
        `self.last_bn_sums` = the replicated fp64 {sum y, sum y^2} (callers that also keep running statistics)."""
        rows, c = y.shape
        dev = y.device
        sums = zeros(BN_REPLICAS * 2 * c, torch.float64, dev)
        self.last_bn_sums = sums
        out = torch.empty_like(y)
        mean_invstd = torch.empty(2 * c, dtype=_F32, device=dev)
        st = self._stream()
        self._call("i2p_bn_stats", int(rows), int(c), self._p(y, _F32, "y"), self._p(sums, torch.float64, "sums"),
                   stream=st)
        self._call("i2p_bn_act_fwd", int(rows), int(c), self._p(y, _F32, "y"), self._p(sums, torch.float64, "sums"),
                   self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(eps), float(slope),
                   self._p(out, _F32, "out"), self._p(mean_invstd, _F32, "mean_invstd"), stream=st)
        return out, mean_invstd

    def bn_act_maxk_forward(self, y, coef, slope, K):
        """y [groups*K, c] pre-BN -> (out [groups,c] = max_k act(bn(y)), arg u8 [groups,c])"""
        rows, c = y.shape
        groups = rows // K
        out = torch.empty(groups, c, dtype=_F32, device=y.device)
        arg = torch.empty(groups, c, dtype=torch.uint8, device=y.device)
        self._call("i2p_bn_act_maxk_fwd_bf16" if y.dtype == _BF16 else "i2p_bn_act_maxk_fwd", int(groups), int(K), int(c),
                   self._p(y, y.dtype, "y"), self._p(coef, _F32, "coef"),
                   float(slope), self._p(out, _F32, "out"), self._p(arg, torch.uint8, "arg"), stream=self._stream())
        return out, arg

    def unpool_k(self, g, arg, K, dtype=_F32):
        groups, c = g.shape
        gd = torch.empty(groups * K, c, dtype=dtype, device=g.device)
        self._call("i2p_unpool_k_bf16" if dtype == _BF16 else "i2p_unpool_k", int(groups), int(K), int(c), self._p(g, _F32, "g"),
                   self._p(arg, torch.uint8, "arg"), self._p(gd, dtype, "gd"), stream=self._stream())
        return gd

    def unpool_k_stats(self, g, arg, K, y, mean_invstd, gamma, beta, slope):
        """unpool_k + bn_act_backward_stats of its result in one launch -> (dense dL/da [groups*K, c], dsums)"""
        groups, c = g.shape
        gd = torch.empty(groups * K, c, dtype=_F32, device=g.device)
        dsums = zeros(BN_REPLICAS * 2 * c, torch.float64, g.device)
        self._call("i2p_unpool_k_stats", int(groups), int(K), int(c), self._p(g, _F32, "g"), self._p(arg, torch.uint8, "arg"),
                   self._p(y, _F32, "y"), self._p(mean_invstd, _F32, "mean_invstd"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(slope), self._p(gd, _F32, "gd"), self._p(dsums, torch.float64, "dsums"), stream=self._stream())
        return gd, dsums

    # ---- bf16-storage helpers (csrc/bf16_stream.hip) -----------------------------------------------------------
    def to_bf16(self, x):
        x = x.contiguous()
        if x.numel() % 8:
            return x.to(_BF16)
        y = torch.empty(x.shape, dtype=_BF16, device=x.device)
        self._call("i2p_to_bf16", int(x.numel()), self._p(x, _F32, "x"), self._p(y, _BF16, "y"), stream=self._stream())
        return y

    def outer_sum_bf16(self, enc_n, enc_k):
        """enc_n [B,N,C], enc_k [B,M,C] -> (ye bf16 [B*N*M, C] = enc_n[b,n] + enc_k[b,k], replicated BN sums)"""
        B, N, C = enc_n.shape
        M = enc_k.shape[1]
        ye = torch.empty(B * N * M, C, dtype=_BF16, device=enc_n.device)
        sums = zeros(BN_REPLICAS * 2 * C, torch.float64, enc_n.device)
        self._call("i2p_outer_sum_bf16", int(B), int(N), int(M), int(C), self._p(enc_n, _F32, "enc_n"), self._p(enc_k, _F32, "enc_k"),
                   self._p(ye, _BF16, "ye"), self._p(sums, torch.float64, "sums"), stream=self._stream())
        return ye, sums

    def outer_sum(self, enc_n, enc_k):
        """enc_n [B,N,C], enc_k [B,M,C] -> (ye f32 [B*N*M, C] = enc_n[b,n] + enc_k[b,k], replicated BN sums) in one pass"""
        B, N, C = enc_n.shape
        M = enc_k.shape[1]
        ye = torch.empty(B * N * M, C, dtype=_F32, device=enc_n.device)
        sums = zeros(BN_REPLICAS * 2 * C, torch.float64, enc_n.device)
        self._call("i2p_outer_sum", int(B), int(N), int(M), int(C), self._p(enc_n, _F32, "enc_n"), self._p(enc_k, _F32, "enc_k"),
                   self._p(ye, _F32, "ye"), self._p(sums, torch.float64, "sums"), stream=self._stream())
        return ye, sums

    def bn_act_apply_bf16(self, y, coef, slope):
        """act(bn(y)) of a bf16 pre-BN tensor with finalised coefficients -> fp32 [rows, c] (a chain's output)"""
        rows, c = y.shape
        out = torch.empty(rows, c, dtype=_F32, device=y.device)
        self._call("i2p_bn_act_fwd_bf16", int(rows), int(c), self._p(y, _BF16, "y"), self._p(coef, _F32, "coef"), float(slope),
                   self._p(out, _F32, "out"), stream=self._stream())
        return out

    def bn_act_backward_stats_bf16(self, dout, y, coef, mi, slope):
        rows, c = y.shape
        dsums = zeros(BN_REPLICAS * 2 * c, torch.float64, y.device)
        self._call("i2p_bn_act_bwd_stats_bf16", int(rows), int(c), self._p(dout, _BF16, "dout"), self._p(y, _BF16, "y"),
                   self._p(coef, _F32, "coef"), self._p(mi, _F32, "mi"), float(slope), self._p(dsums, torch.float64, "dsums"),
                   stream=self._stream())
        return dsums

    def bn_act_backward_stats(self, dout, y, mean_invstd, gamma, beta, slope):
        """replicated {sum gz, sum gz*xhat} with gz = dout * act'(bn(y)) — the statistics half of bn_act_backward"""
        rows, c = y.shape
        dsums = zeros(BN_REPLICAS * 2 * c, torch.float64, y.device)
        self._call("i2p_bn_act_bwd_stats", int(rows), int(c), self._p(dout, _F32, "dout"), self._p(y, _F32, "y"),
                   self._p(mean_invstd, _F32, "mean_invstd"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(slope), self._p(dsums, torch.float64, "dsums"), stream=self._stream())
        return dsums

    def bn_act_backward(self, dout, y, mean_invstd, gamma, beta, slope):
        """-> (dy [rows,c], dgamma [c], dbeta [c])"""
        rows, c = y.shape
        dev = y.device
        dsums = zeros(BN_REPLICAS * 2 * c, torch.float64, dev)
        dy = torch.empty_like(y)
        dgamma = torch.empty(c, dtype=_F32, device=dev)
        dbeta = torch.empty(c, dtype=_F32, device=dev)
        st = self._stream()
        args = (self._p(dout, _F32, "dout"), self._p(y, _F32, "y"), self._p(mean_invstd, _F32, "mean_invstd"),
                self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(slope))
        self._call("i2p_bn_act_bwd_stats", int(rows), int(c), *args, self._p(dsums, torch.float64, "dsums"), stream=st)
        self._call("i2p_bn_act_bwd", int(rows), int(c), *args, self._p(dsums, torch.float64, "dsums"),
                   self._p(dy, _F32, "dy"), self._p(dgamma, _F32, "dgamma"), self._p(dbeta, _F32, "dbeta"), stream=st)
        return dy, dgamma, dbeta

    # ---- layer + BN finalisation in one launch (device library; other backends: the two calls) ----------------------
    def _fin_buffers(self, cout, dev):
        coef = torch.empty(3, cout, dtype=_F32, device=dev)
        mi = torch.empty(2 * cout, dtype=_F32, device=dev)
        counter = zeros(1, torch.int32, dev)
        return coef, mi, counter

    def lin_forward_fin(self, x, in_coef, slope_in, w, gamma, beta, eps, out_dtype=_F32):
        """lin_forward + bn_finalize of its output statistics -> (y, sums, coef [3,cout], mean_invstd [2cout])"""
        rows, cin = x.shape
        cout = w.shape[0]
        if self.device_type != "cuda" or out_dtype != _F32:
            y, sums = self.lin_forward(x, in_coef, slope_in, w, out_dtype=out_dtype)
            coef, mi = self.bn_finalize(rows, sums, gamma, beta, eps)
            return y, sums, coef, mi
        dev = x.device
        y = torch.empty(rows, cout, dtype=_F32, device=dev)
        sums = zeros(BN_REPLICAS * 2 * cout, torch.float64, dev)
        coef, mi, counter = self._fin_buffers(cout, dev)
        self._call("i2p_lin_fwd_fin", int(rows), int(cin), int(cout), self._p(x, _F32, "x"),
                   self._p(in_coef, _F32, "in_coef") if in_coef is not None else None, float(slope_in), self._p(w, _F32, "w"),
                   self._p(y, _F32, "y"), self._p(sums, torch.float64, "sums"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(eps), self._p(coef, _F32, "coef"), self._p(mi, _F32, "mi"), self._p(counter, torch.int32, "counter"),
                   stream=self._stream())
        return y, sums, coef, mi

    def lin_forward_2src_fin(self, xa, coef_a, slope_a, xb, coef_b, slope_b, w, gamma, beta, eps):
        rows, ca = xa.shape
        cb, cout = xb.shape[1], w.shape[0]
        if self.device_type != "cuda" or xa.dtype != _F32:
            y, sums = self.lin_forward_2src(xa, coef_a, slope_a, xb, coef_b, slope_b, w)
            coef, mi = self.bn_finalize(rows, sums, gamma, beta, eps)
            return y, sums, coef, mi
        dev = xa.device
        y = torch.empty(rows, cout, dtype=_F32, device=dev)
        sums = zeros(BN_REPLICAS * 2 * cout, torch.float64, dev)
        coef, mi, counter = self._fin_buffers(cout, dev)
        self._call("i2p_lin_fwd_2src_fin", int(rows), int(ca), int(cb), int(cout), self._p(xa, _F32, "xa"), self._p(coef_a, _F32, "coef_a"),
                   float(slope_a), self._p(xb, _F32, "xb"), self._p(coef_b, _F32, "coef_b"), float(slope_b), self._p(w, _F32, "w"),
                   self._p(y, _F32, "y"), self._p(sums, torch.float64, "sums"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"),
                   float(eps), self._p(coef, _F32, "coef"), self._p(mi, _F32, "mi"), self._p(counter, torch.int32, "counter"),
                   stream=self._stream())
        return y, sums, coef, mi

    def pair_lin_forward_fin(self, f, g, bias_n, bias_k, w, gamma, beta, eps, out_dtype=_F32):
        B, N, C = f.shape
        M, Co = g.shape[1], w.shape[0]
        if self.device_type != "cuda" or out_dtype != _F32:
            y, sums = self.pair_lin_forward(f, g, bias_n, bias_k, w, out_dtype=out_dtype)
            coef, mi = self.bn_finalize(B * N * M, sums, gamma, beta, eps)
            return y, sums, coef, mi
        dev = f.device
        y = torch.empty(B * N * M, Co, dtype=_F32, device=dev)
        sums = zeros(BN_REPLICAS * 2 * Co, torch.float64, dev)
        coef, mi, counter = self._fin_buffers(Co, dev)
        self._call("i2p_pair_lin_fwd_fin", int(B), int(N), int(M), int(C), int(Co), self._p(f, _F32, "f"), self._p(g, _F32, "g"),
                   self._p(bias_n, _F32, "bias_n"), self._p(bias_k, _F32, "bias_k"), self._p(w, _F32, "w"), self._p(y, _F32, "y"),
                   self._p(sums, torch.float64, "sums"), self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(eps),
                   self._p(coef, _F32, "coef"), self._p(mi, _F32, "mi"), self._p(counter, torch.int32, "counter"), stream=self._stream())
        return y, sums, coef, mi

    # ---- fused linear layers (csrc/mlp.hip) -----------------------------------------------------
    def lin_forward(self, x, in_coef, slope_in, w, want_stats=True, out_dtype=_F32):
        """x [rows,cin]; in_coef [3,cin] or None; w [cout,cin] -> (y [rows,cout], sums or None).
        out_dtype bf16: bf16-stored output (x fp32 or bf16) on the bf16 MFMA kernels."""
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        if out_dtype == _BF16:
            y = torch.empty(rows, cout, dtype=_BF16, device=dev)
            sums = zeros(BN_REPLICAS * 2 * cout, torch.float64, dev)
            self._call("i2p_lin_fwd_bf16", int(rows), int(cin), int(cout), self._p(x, x.dtype, "x"), int(x.dtype == _BF16),
                       self._p(in_coef, _F32, "in_coef") if in_coef is not None else None, float(slope_in),
                       self._p(w, _F32, "w"), self._p(y, _BF16, "y"), self._p(sums, torch.float64, "sums"), stream=self._stream())
            return y, sums
        y = torch.empty(rows, cout, dtype=_F32, device=dev)
        sums = zeros(BN_REPLICAS * 2 * cout, torch.float64, dev) if want_stats else None
        self._call("i2p_lin_fwd", int(rows), int(cin), int(cout), self._p(x, _F32, "x"),
                   self._p(in_coef, _F32, "in_coef") if in_coef is not None else None, float(slope_in),
                   self._p(w, _F32, "w"), self._p(y, _F32, "y"),
                   self._p(sums, torch.float64, "sums") if sums is not None else None, stream=self._stream())
        return y, sums

    def lin_backward(self, gz, y, out_coef, out_mi, out_dsums, x, in_coef, in_mi, slope_in, w, need_gx=True,
                     slope_out=1.0):
        """-> (gz_in [rows,cin] or None, in_dsums or None, dw [cout,cin]); see i2p_lin_bwd.
        slope_out != 1: `gz` is dL/da of this layer's activation (applied on load).
        `self.last_bn_grads` = (dgamma, dbeta) [cout] of the BN behind this layer when out_coef is given
        (reduced from out_dsums by the launcher; views into the scratch tensor), else None."""
        rows, cout = gz.shape
        cin = x.shape[1]
        dev = gz.device
        if gz.dtype == _BF16:
            gz_in = torch.empty(rows, cin, dtype=x.dtype, device=dev) if need_gx else None
            in_dsums = (zeros(BN_REPLICAS * 2 * cin, torch.float64, dev) if (need_gx and in_coef is not None) else None)
            grid = _lib.helper("i2p_lin_bwd_bf16_grid", int(rows))
            part = torch.empty(grid * cout * cin + 8 * cout, dtype=_F32, device=dev)
            dw = torch.empty(cout, cin, dtype=_F32, device=dev)
            P = lambda t, dt=_F32: (self._p(t, dt, "t") if t is not None else None)
            self._call("i2p_lin_bwd_bf16", int(rows), int(cin), int(cout), P(gz, _BF16), P(y, _BF16), P(out_coef), P(out_mi),
                       P(out_dsums, torch.float64), P(x, x.dtype), int(x.dtype == _BF16), P(in_coef), P(in_mi), float(slope_in),
                       P(w), P(gz_in, x.dtype), int(x.dtype == _BF16), P(in_dsums, torch.float64), P(part), P(dw), float(slope_out),
                       stream=self._stream())
            n = part.numel()
            self.last_bn_grads = (part[n - cout:], part[n - 2 * cout:n - cout]) if out_coef is not None else None
            defer_keep(part, dw=dw)
            return gz_in, in_dsums, dw
        gz_in = torch.empty(rows, cin, dtype=_F32, device=dev) if need_gx else None
        in_dsums = (zeros(BN_REPLICAS * 2 * cin, torch.float64, dev)
                    if (need_gx and in_coef is not None) else None)
        grid = 256 if self.device_type == "cuda" else 1
        part = torch.empty(min(grid, (rows + 63) // 64) * cout * cin + 8 * cout, dtype=_F32, device=dev)
        dw = torch.empty(cout, cin, dtype=_F32, device=dev)
        P = lambda t, dt=_F32, n="t": (self._p(t, dt, n) if t is not None else None)
        n = part.numel()
        self.last_bn_grads = (part[n - cout:], part[n - 2 * cout:n - cout]) if out_coef is not None else None
        self._call("i2p_lin_bwd", int(rows), int(cin), int(cout), P(gz, _F32, "gz"), P(y, _F32, "y"),
                   P(out_coef, _F32, "out_coef"), P(out_mi, _F32, "out_mi"), P(out_dsums, torch.float64, "out_dsums"),
                   P(x, _F32, "x"), P(in_coef, _F32, "in_coef"), P(in_mi, _F32, "in_mi"), float(slope_in),
                   P(w, _F32, "w"), P(gz_in, _F32, "gz_in"), P(in_dsums, torch.float64, "in_dsums"),
                   P(part, _F32, "dw_partial"), P(dw, _F32, "dw"), float(slope_out), stream=self._stream())
        defer_keep(part, dw=dw)
        return gz_in, in_dsums, dw

    def after_wgrad(self, fn):
        """run `fn()` (launches that consume the weight gradient of the LAST lin_backward) where that gradient is produced — in place
        (a side stream for the weight gradients was measured in round 3, 577 vs 582 samples/s, and removed in round 5)"""
        return fn()

    def take_bn_grads(self):
        """(dgamma, dbeta) of the last lin_backward / lin_backward_2src, handing the reference over: a pair still
        referenced from here when autograd's AccumulateGrad sees it is CLONED instead of adopted (one copy launch
        per tensor)."""
        g, self.last_bn_grads = self.last_bn_grads, None
        return g

    def pair_lin_forward(self, f, g, bias_n, bias_k, w, out_dtype=_F32):
        """f [B,N,C], g [B,M,C], bias_n [B,N,Co], bias_k [B,M,Co], w [Co,C] -> y [B*N*M, Co], sums"""
        B, N, C = f.shape
        M = g.shape[1]
        Co = w.shape[0]
        if out_dtype == _BF16:
            y = torch.empty(B * N * M, Co, dtype=_BF16, device=f.device)
            sums = zeros(BN_REPLICAS * 2 * Co, torch.float64, f.device)
            self._call("i2p_pair_lin_fwd_bf16", int(B), int(N), int(M), int(C), int(Co), self._p(f, _F32, "f"), self._p(g, _F32, "g"),
                       self._p(bias_n, _F32, "bias_n"), self._p(bias_k, _F32, "bias_k"), self._p(w, _F32, "w"), self._p(y, _BF16, "y"),
                       self._p(sums, torch.float64, "sums"), stream=self._stream())
            return y, sums
        y = torch.empty(B * N * M, Co, dtype=_F32, device=f.device)
        sums = zeros(BN_REPLICAS * 2 * Co, torch.float64, f.device)
        self._call("i2p_pair_lin_fwd", int(B), int(N), int(M), int(C), int(Co), self._p(f, _F32, "f"),
                   self._p(g, _F32, "g"), self._p(bias_n, _F32, "bias_n"), self._p(bias_k, _F32, "bias_k"),
                   self._p(w, _F32, "w"), self._p(y, _F32, "y"), self._p(sums, torch.float64, "sums"),
                   stream=self._stream())
        return y, sums

    def pair_lin_backward(self, gy, f, g, w, y=None, out_coef=None, out_mi=None, out_dsums=None):
        """gy = dL/dy [B*N*M, Co] (or dL/dz with the BN behind given by y/out_coef/out_mi/out_dsums: BN backward
        on load) -> (d_f, d_g, d_bias_n, d_bias_k, dw)"""
        B, N, C = f.shape
        M = g.shape[1]
        Co = w.shape[0]
        dev = f.device
        # the four accumulated outputs from ONE zeroed buffer (one fill launch instead of four; every piece a multiple of 128 floats)
        sizes = (B * N * C, B * M * C, B * N * Co, B * M * Co)
        flat = zeros(sum(sizes), _F32, dev)
        d_f, d_g, d_bn, d_bk = [t.view(shape) for t, shape in zip(flat.split(sizes), ((B, N, C), (B, M, C), (B, N, Co), (B, M, Co)))]
        if gy.dtype == _BF16:
            grid = _lib.helper("i2p_pair_lin_bwd_bf16_grid", int(B), int(N), int(M))
            part = torch.empty(grid * Co * C + 8 * Co, dtype=_F32, device=dev)
            dw = torch.empty(Co, C, dtype=_F32, device=dev)
            o = lambda t, dt=_F32: self._p(t, dt, "bn") if t is not None else None
            self._call("i2p_pair_lin_bwd_bf16", int(B), int(N), int(M), int(C), int(Co), self._p(gy, _BF16, "gy"), o(y, _BF16),
                       o(out_coef), o(out_mi), o(out_dsums, torch.float64), self._p(f, _F32, "f"), self._p(g, _F32, "g"),
                       self._p(w, _F32, "w"), self._p(d_f, _F32, "d_f"), self._p(d_g, _F32, "d_g"), self._p(d_bn, _F32, "d_bn"),
                       self._p(d_bk, _F32, "d_bk"), self._p(part, _F32, "part"), self._p(dw, _F32, "dw"), stream=self._stream())
            defer_keep(part, dw=dw)
            return d_f, d_g, d_bn, d_bk, dw
        # per-block weight-gradient partials + the slabs of the deterministic pair sums (i2p_pair_lin_bwd_scratch)
        nscr = _lib.helper("i2p_pair_lin_bwd_scratch", int(B), int(N), int(M), int(C), int(Co)) if self.device_type == "cuda" else Co * C
        part = torch.empty(nscr, dtype=_F32, device=dev)
        dw = torch.empty(Co, C, dtype=_F32, device=dev)
        opt = lambda t, dt=_F32: self._p(t, dt, "bn") if t is not None else None
        self._call("i2p_pair_lin_bwd", int(B), int(N), int(M), int(C), int(Co), self._p(gy, _F32, "gy"), opt(y),
                   opt(out_coef), opt(out_mi), opt(out_dsums, torch.float64), self._p(f, _F32, "f"), self._p(g, _F32, "g"), self._p(w, _F32, "w"),
                   self._p(d_f, _F32, "d_f"), self._p(d_g, _F32, "d_g"), self._p(d_bn, _F32, "d_bn"),
                   self._p(d_bk, _F32, "d_bk"), self._p(part, _F32, "part"), self._p(dw, _F32, "dw"),
                   stream=self._stream())
        defer_keep(part, dw=dw)
        return d_f, d_g, d_bn, d_bk, dw

    # ---- cost-volume tail --------------------------------------------------------------------------
    def bn_stats(self, x):
        rows, c = x.shape
        sums = zeros(BN_REPLICAS * 2 * c, torch.float64, x.device)
        self._call("i2p_bn_stats", int(rows), int(c), self._p(x, _F32, "x"), self._p(sums, torch.float64, "sums"),
                   stream=self._stream())
        return sums

    def lin_forward_2src(self, xa, coef_a, slope_a, xb, coef_b, slope_b, w):
        rows, ca = xa.shape
        cb = xb.shape[1]
        cout = w.shape[0]
        y = torch.empty(rows, cout, dtype=xa.dtype, device=xa.device)
        sums = zeros(BN_REPLICAS * 2 * cout, torch.float64, xa.device)
        if xa.dtype == _BF16:
            self._call("i2p_lin_fwd_2src_bf16", int(rows), int(ca), int(cb), int(cout), self._p(xa, _BF16, "xa"),
                       self._p(coef_a, _F32, "coef_a"), float(slope_a), self._p(xb, _BF16, "xb"), self._p(coef_b, _F32, "coef_b"),
                       float(slope_b), self._p(w, _F32, "w"), self._p(y, _BF16, "y"), self._p(sums, torch.float64, "sums"),
                       stream=self._stream())
            return y, sums
        self._call("i2p_lin_fwd_2src", int(rows), int(ca), int(cb), int(cout), self._p(xa, _F32, "xa"),
                   self._p(coef_a, _F32, "coef_a"), float(slope_a), self._p(xb, _F32, "xb"),
                   self._p(coef_b, _F32, "coef_b"), float(slope_b), self._p(w, _F32, "w"), self._p(y, _F32, "y"),
                   self._p(sums, torch.float64, "sums"), stream=self._stream())
        return y, sums

    def lin_backward_2src(self, gz, y, out_coef, out_mi, out_dsums, xa, coef_a, mi_a, slope_a, xb, coef_b, mi_b,
                          slope_b, e_add_b, w):
        """-> (gz_a, dsums_a, gz_b, dsums_b, dw)"""
        rows, cout = gz.shape
        ca, cb = xa.shape[1], xb.shape[1]
        dev = gz.device
        gz_a = torch.empty(rows, ca, dtype=gz.dtype, device=dev); gz_b = torch.empty(rows, cb, dtype=gz.dtype, device=dev)
        ds_a = zeros(BN_REPLICAS * 2 * ca, torch.float64, dev)
        ds_b = zeros(BN_REPLICAS * 2 * cb, torch.float64, dev)
        if gz.dtype == _BF16:
            grid = _lib.helper("i2p_lin_bwd_bf16_grid", int(rows))
            part = torch.empty(grid * cout * (ca + cb) + 8 * cout, dtype=_F32, device=dev)
            dw = torch.empty(cout, ca + cb, dtype=_F32, device=dev)
            P = lambda t, dt=_F32: (self._p(t, dt, "t") if t is not None else None)
            self._call("i2p_lin_bwd_2src_bf16", int(rows), int(ca), int(cb), int(cout), P(gz, _BF16), P(y, _BF16), P(out_coef),
                       P(out_mi), P(out_dsums, torch.float64), P(xa, _BF16), P(coef_a), P(mi_a), float(slope_a), P(xb, _BF16),
                       P(coef_b), P(mi_b), float(slope_b), P(e_add_b, _BF16), P(w), P(gz_a, _BF16), P(ds_a, torch.float64),
                       P(gz_b, _BF16), P(ds_b, torch.float64), P(part), P(dw), stream=self._stream())
            n = part.numel()
            self.last_bn_grads = (part[n - cout:], part[n - 2 * cout:n - cout]) if out_coef is not None else None
            defer_keep(part, dw=dw)
            return gz_a, ds_a, gz_b, ds_b, dw
        grid = 256 if self.device_type == "cuda" else 1
        part = torch.empty(min(grid, (rows + 63) // 64) * cout * (ca + cb) + 8 * cout, dtype=_F32, device=dev)
        dw = torch.empty(cout, ca + cb, dtype=_F32, device=dev)
        P = lambda t, dt=_F32: (self._p(t, dt, "t") if t is not None else None)
        self._call("i2p_lin_bwd_2src", int(rows), int(ca), int(cb), int(cout), P(gz), P(y), P(out_coef), P(out_mi),
                   P(out_dsums, torch.float64), P(xa), P(coef_a), P(mi_a), float(slope_a), P(xb), P(coef_b), P(mi_b),
                   float(slope_b), P(e_add_b), P(w), P(gz_a), P(ds_a, torch.float64), P(gz_b), P(ds_b, torch.float64),
                   P(part), P(dw), stream=self._stream())
        n = part.numel()
        self.last_bn_grads = (part[n - cout:], part[n - 2 * cout:n - cout]) if out_coef is not None else None
        defer_keep(part, dw=dw)
        return gz_a, ds_a, gz_b, ds_b, dw

    def cv_softmax_wsum_forward(self, B, N, M, y5, coef5, slope5, y3, coef3, slope3):
        C = y5.shape[1]
        out = torch.empty(B, N, C, dtype=_F32, device=y5.device)
        msave = torch.empty(B * N, 2, C, dtype=_F32, device=y5.device)
        dt = y5.dtype
        self._call("i2p_cv_softmax_wsum_fwd_bf16" if dt == _BF16 else "i2p_cv_softmax_wsum_fwd", int(B), int(N), int(M), int(C), self._p(y5, dt, "y5"),
                   self._p(coef5, _F32, "coef5"), float(slope5), self._p(y3, dt, "y3"), self._p(coef3, _F32, "coef3"),
                   float(slope3), self._p(out, _F32, "out"), self._p(msave, _F32, "msave"), stream=self._stream())
        return out, msave

    def cv_softmax_wsum_backward(self, B, N, M, g_out, out, msave, y5, coef5, mi5, slope5, y3, coef3, slope3):
        C = y5.shape[1]
        gz5 = torch.empty_like(y5); ga3 = torch.empty_like(y3)
        ds5 = zeros(BN_REPLICAS * 2 * C, torch.float64, y5.device)
        dt = y5.dtype
        self._call("i2p_cv_softmax_wsum_bwd_bf16" if dt == _BF16 else "i2p_cv_softmax_wsum_bwd", int(B), int(N), int(M), int(C),
                   self._p(g_out, _F32, "g_out"),
                   self._p(out, _F32, "out"), self._p(msave, _F32, "msave"), self._p(y5, dt, "y5"),
                   self._p(coef5, _F32, "coef5"), self._p(mi5, _F32, "mi5"), float(slope5), self._p(y3, dt, "y3"),
                   self._p(coef3, _F32, "coef3"), float(slope3), self._p(gz5, dt, "gz5"),
                   self._p(ds5, torch.float64, "ds5"), self._p(ga3, dt, "ga3"), stream=self._stream())
        return gz5, ds5, ga3

    def pair_bias_bn_backward(self, B, N, M, gz, enc_n, enc_k, dsums, coef, mi):
        """-> (d_enc_n [B,N,C], d_enc_k [B,M,C]); see i2p_pair_bias_bn_bwd"""
        C = gz.shape[1]
        dev = gz.device
        d_n = torch.empty(B, N, C, dtype=_F32, device=dev); d_k = torch.empty(B, M, C, dtype=_F32, device=dev)
        if self.device_type == "cuda" and gz.dtype == _F32:        # deterministic two-level sums (no atomics)
            scr = torch.empty(_lib.helper("i2p_pair_bias_bn_bwd_scratch", int(B), int(N), int(M), int(C)), dtype=_F32, device=dev)
            self._call("i2p_pair_bias_bn_bwd_det", int(B), int(N), int(M), int(C), self._p(gz, _F32, "gz"), self._p(enc_n, _F32, "enc_n"),
                       self._p(enc_k, _F32, "enc_k"), self._p(dsums, torch.float64, "dsums"), self._p(coef, _F32, "coef"),
                       self._p(mi, _F32, "mi"), self._p(scr, _F32, "scratch"), self._p(d_n, _F32, "d_enc_n"),
                       self._p(d_k, _F32, "d_enc_k"), stream=self._stream())
            return d_n, d_k
        sum_k = zeros((B, N, C), _F32, dev)
        sum_n = zeros((B, M, C), _F32, dev)
        self._call("i2p_pair_bias_bn_bwd_bf16" if gz.dtype == _BF16 else "i2p_pair_bias_bn_bwd", int(B), int(N), int(M), int(C),
                   self._p(gz, gz.dtype, "gz"),
                   self._p(enc_n, _F32, "enc_n"), self._p(enc_k, _F32, "enc_k"), self._p(dsums, torch.float64, "dsums"),
                   self._p(coef, _F32, "coef"), self._p(mi, _F32, "mi"), self._p(sum_k, _F32, "sum_k"),
                   self._p(sum_n, _F32, "sum_n"), self._p(d_n, _F32, "d_enc_n"), self._p(d_k, _F32, "d_enc_k"),
                   stream=self._stream())
        return d_n, d_k

    def pose_loss(self, out3, out4, q_gt, t_gt, w_x, w_q, l1_trans):
        """-> (loss3 [3], d_out3 [B,7], d_out4 [B,7], d_w [2]); compute_loss.py:102-133"""
        B = out3.shape[0]
        dev = out3.device
        loss3 = torch.empty(3, dtype=_F32, device=dev)
        # the three gradients live in ONE buffer (16-byte aligned pieces): the backward scales them by dL/dloss with a single launch
        n7 = (B * 7 + 3) // 4 * 4
        flat = torch.empty(2 * n7 + 4, dtype=_F32, device=dev)
        d3, d4, d_w = flat[:B * 7].view(B, 7), flat[n7:n7 + B * 7].view(B, 7), flat[2 * n7:2 * n7 + 2]
        self.last_pose_loss_flat = (flat, n7)
        self._call("i2p_pose_loss", int(B), int(bool(l1_trans)), self._p(out3, _F32, "out3"), self._p(out4, _F32, "out4"),
                   self._p(q_gt, _F32, "q_gt"), self._p(t_gt, _F32, "t_gt"), self._p(w_x, _F32, "w_x"), self._p(w_q, _F32, "w_q"),
                   self._p(loss3, _F32, "loss3"), self._p(d3, _F32, "d_out3"), self._p(d4, _F32, "d_out4"), self._p(d_w, _F32, "d_w"),
                   stream=self._stream())
        return loss3, d3, d4, d_w

    # ---- loader: the device-side build of a KITTI batch (csrc/loader_build.hip; device library only) ------------------
    def kitti_points_build(self, table, B, sample_point, noise):
        """table: device uint8 [B*128] rows of i2p_kitti_points_build -> lidar, raw [B,SP,3], feats [B,SP,1] (padding rows zero)"""
        dev = table.device
        lidar = torch.empty(B, sample_point, 3, dtype=_F32, device=dev); raw = torch.empty_like(lidar)
        feats = torch.empty(B, sample_point, 1, dtype=_F32, device=dev)
        self._call("i2p_kitti_points_build", int(B), int(sample_point), self._p(table, torch.uint8, "table"),
                   self._p(noise, _F32, "noise") if noise is not None else None, self._p(lidar, _F32, "lidar"), self._p(raw, _F32, "raw"),
                   self._p(feats, _F32, "feats"), stream=self._stream())
        return lidar, raw, feats

    def kitti_image_build(self, table, B, out_h, out_w):
        """table: device uint8 [B*64] rows of i2p_kitti_image_build -> rgb [B,3,out_h,out_w] float in 0..255"""
        rgb = torch.empty(B, 3, out_h, out_w, dtype=_F32, device=table.device)
        self._call("i2p_kitti_image_build", int(B), int(out_h), int(out_w), self._p(table, torch.uint8, "table"), self._p(rgb, _F32, "rgb"),
                   stream=self._stream())
        return rgb

    # ---- small glue kernels (csrc/glue.hip; device library only) -------------------------------------------------
    def intrinsic_inverse(self, K, sx, sy):
        """K [B,3,3] -> inverse of K with fx, cx scaled by sx and fy, cy by sy (change_intrinsic + inverse_3x3 of model.py in one launch)"""
        out = torch.empty_like(K)
        self._call("i2p_intrinsic_inverse", int(K.shape[0]), self._p(K, _F32, "K"), float(sx), float(sy), self._p(out, _F32, "out"), stream=self._stream())
        return out

    def row_valid(self, x):
        """x [..., c] -> 0/1 float [..., 1]: any(x != 0) over the last axis (check_valid)"""
        c = x.shape[-1]
        x2 = x.reshape(-1, c)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out = torch.empty(x2.shape[0], dtype=_F32, device=x.device)
        self._call("i2p_row_valid", int(x2.shape[0]), int(c), self._p(x2, _F32, "x"), self._p(out, _F32, "out"), stream=self._stream())
        return out.view(*x.shape[:-1], 1)

    def mask_fill_rows(self, x2, valid, fill):
        """x2 [rows, c], valid [rows] -> valid > 0 ? x : fill"""
        out = torch.empty_like(x2)
        self._call("i2p_mask_fill", int(x2.shape[0]), int(x2.shape[1]), self._p(x2, _F32, "x"), self._p(valid, _F32, "valid"), float(fill),
                   self._p(out, _F32, "out"), stream=self._stream())
        return out

    # ---- whole small MLP chain in one launch (csrc/mlp_chain.hip) --------------------------------
    def chain_fits(self, rows, widths, pool_k):
        """does i2p_chain_fwd take this chain on the current device?  widths = [row length of x, cout_1, ..., cout_nl].
        (Chain launches need their whole grid resident: they must not overlap another chain launch on the same GPU — the step runs
        them on one stream; the side-stream weight-gradient option only moves layer kernels, never these.)"""
        if self.name != "hip" or self.device_type != "cuda" or os.environ.get("I2P_NO_CHAIN") == "1" or _CHAINS_OFF[0]:
            return False
        key = (torch.cuda.current_device(), int(rows), tuple(int(c) for c in widths), int(pool_k))
        chain_error_words()                    # registers this device's error sinks on first use (outside graph capture)
        hit = _CHAIN_OK.get(key)
        if hit is None:
            arr = (C.c_int * len(widths))(*key[2])
            hit = bool(_lib.helper("i2p_chain_fwd_ok", key[1], len(widths) - 1, C.cast(arr, C.c_void_p), key[3]))
            _CHAIN_OK[key] = hit
        return hit

    def chain_forward(self, x, weights, gammas, betas, slopes, eps, pool_k, want_w0_pad):
        """x [rows, c0]; weights[l] [c_{l+1}, cin_l] with cin_0 <= c0 -> (ys, coefs, mis, out, arg or None, w0_pad or None):
        every pre-BN output, its coef [3,c] / mean_invstd [2c], and out = act(bn(y_last)) or its max over groups of pool_k rows."""
        rows, c0 = x.shape
        nl = len(weights)
        dev = x.device
        widths = [int(c0)] + [int(w.shape[0]) for w in weights]
        ys = [torch.empty(rows, c, dtype=_F32, device=dev) for c in widths[1:]]
        coefs = [torch.empty(3, c, dtype=_F32, device=dev) for c in widths[1:]]
        mis = [torch.empty(2 * c, dtype=_F32, device=dev) for c in widths[1:]]
        c_last = widths[-1]
        if pool_k:
            out = torch.empty(rows // pool_k, c_last, dtype=_F32, device=dev)
            arg = torch.empty(rows // pool_k, c_last, dtype=torch.uint8, device=dev)
        else:
            out, arg = torch.empty(rows, c_last, dtype=_F32, device=dev), None
        w0_pad = torch.empty(widths[1], c0, dtype=_F32, device=dev) if want_w0_pad else None
        sums = zeros(int(_lib.helper("i2p_chain_sums_len", nl, max(widths[1:]))), torch.float64, dev)
        nsync = int(_lib.helper("i2p_chain_sync_words"))
        sync = zeros(nsync, torch.int32, dev)
        i_arr = lambda v: (C.c_int * len(v))(*[int(a) for a in v])
        wd, ld = i_arr(widths), i_arr([w.shape[1] for w in weights])
        sl = (C.c_float * nl)(*[float(a) for a in slopes])
        pa = [_abi.ptr_array(ts) for ts in (weights, gammas, betas, ys, coefs, mis)]
        for ts in (weights, gammas, betas):
            for t in ts:
                self._p(t, _F32, "chain parameter")
        cast = lambda a: C.cast(a, C.c_void_p)
        self._call("i2p_chain_fwd", int(rows), nl, cast(wd), cast(ld), self._p(x, _F32, "x"), cast(pa[0]), cast(pa[1]), cast(pa[2]),
                   cast(sl), float(eps), cast(pa[3]), cast(pa[4]), cast(pa[5]), self._p(sums, torch.float64, "sums"), int(pool_k),
                   self._p(out, _F32, "out"), self._p(arg, torch.uint8, "arg") if arg is not None else None,
                   self._p(w0_pad, _F32, "w0_pad") if w0_pad is not None else None, self._p(sync, torch.int32, "sync"),
                   stream=self._stream())
        self.last_chain_sync = sync            # word [-32]: 1 if a grid barrier timed out (tests read it)
        return ys, coefs, mis, out, arg, w0_pad

    def chain_bwd_fits(self, rows, widths, pool_k):
        # the library takes the chains it measured faster than the layer-by-layer backward (8192 .. 16384 rows: i2p_chain_bwd_ok);
        # I2P_CHAIN_BWD=0 switches the one-launch backward off
        if (self.name != "hip" or self.device_type != "cuda" or os.environ.get("I2P_NO_CHAIN") == "1"
                or os.environ.get("I2P_CHAIN_BWD") == "0"):
            return False
        key = ("bwd", torch.cuda.current_device(), int(rows), tuple(int(c) for c in widths), int(pool_k))
        hit = _CHAIN_OK.get(key)
        if hit is None:
            arr = (C.c_int * len(widths))(*key[3])
            hit = bool(_lib.helper("i2p_chain_bwd_ok", key[2], len(widths) - 1, C.cast(arr, C.c_void_p), key[4]))
            _CHAIN_OK[key] = hit
        return hit

    def chain_backward(self, x, weights, ys, coefs, mis, slopes, g, arg, pool_k, need_gx):
        """backward of chain_forward in two launches (chain + slab reduction): g = dL/dout ([rows, c] or pooled [rows / pool_k, c]
        with arg) -> (dL/dx [rows, c0] or None, [dW_l] (shapes of weights), [dgamma_l], [dbeta_l])"""
        rows, c0 = x.shape
        nl = len(weights)
        dev = x.device
        widths = [int(c0)] + [int(w.shape[0]) for w in weights]
        i_arr = lambda v: (C.c_int * len(v))(*[int(a) for a in v])
        wd, ld = i_arr(widths), i_arr([w.shape[1] for w in weights])
        cast = lambda a: C.cast(a, C.c_void_p)
        tw = int(_lib.helper("i2p_chain_bwd_slab", nl, cast(wd), cast(ld)))
        grid = (rows + 63) // 64
        dw_part = torch.empty(grid * tw, dtype=_F32, device=dev)
        dw = torch.empty(tw, dtype=_F32, device=dev)
        gx = torch.empty(rows, c0, dtype=_F32, device=dev) if need_gx else None
        dgs = [torch.empty(c, dtype=_F32, device=dev) for c in widths[1:]]
        dbs = [torch.empty(c, dtype=_F32, device=dev) for c in widths[1:]]
        sums = zeros(int(_lib.helper("i2p_chain_sums_len", nl, max(widths[1:]))), torch.float64, dev)
        sync = zeros(int(_lib.helper("i2p_chain_sync_words")), torch.int32, dev)
        sl = (C.c_float * nl)(*[float(a) for a in slopes])
        for ts in (weights, ys, coefs, mis):
            for t in ts:
                self._p(t, _F32, "chain tensor")
        pa = [_abi.ptr_array(ts) for ts in (weights, ys, coefs, mis, dgs, dbs)]
        self._call("i2p_chain_bwd", int(rows), nl, cast(wd), cast(ld), self._p(x, _F32, "x"), cast(pa[0]), cast(pa[1]), cast(pa[2]), cast(pa[3]),
                   cast(sl), self._p(g, _F32, "g"), self._p(arg, torch.uint8, "arg") if arg is not None else None, int(pool_k),
                   self._p(gx, _F32, "gx") if gx is not None else None, self._p(dw_part, _F32, "dw_part"), self._p(dw, _F32, "dw"),
                   cast(pa[4]), cast(pa[5]), self._p(sums, torch.float64, "sums"), self._p(sync, torch.int32, "sync"), stream=self._stream())
        self.last_chain_sync = sync
        dws, off = [], 0
        for w in weights:
            n = w.shape[0] * w.shape[1]
            dws.append(dw[off:off + n].view(w.shape[0], w.shape[1]))
            off += n
        defer_keep(dw_part, dw=dws)
        return gx, dws, dgs, dbs

    def pad_cols(self, w, cpad):
        out = torch.empty(w.shape[0], cpad, dtype=_F32, device=w.device)
        self._call("i2p_pad_cols", int(w.shape[0]), int(w.shape[1]), int(cpad), self._p(w, _F32, "w"), self._p(out, _F32, "out"),
                   stream=self._stream())
        return out

    def strided_pick2(self, a, b, oh, ow, sh, sw):
        """a, b [B,H,W,3] (b may be None) -> the [B,oh,ow,3] tensors of their cells (h*sh, w*sw)"""
        B, H, W, _ = a.shape
        oa = torch.empty(B, oh, ow, 3, dtype=_F32, device=a.device)
        ob = torch.empty(B, oh, ow, 3, dtype=_F32, device=a.device) if b is not None else None
        self._call("i2p_strided_pick2", int(B), int(H), int(W), int(oh), int(ow), int(sh), int(sw), self._p(a, _F32, "a"),
                   self._p(b, _F32, "b") if b is not None else None, self._p(oa, _F32, "oa"), self._p(ob, _F32, "ob") if b is not None else None,
                   stream=self._stream())
        return oa, ob

    def pc_rows_forward(self, xyz, pts, feat, h_idx, w_idx, K, W):
        """-> (geo [B,HW*K,12], part [B,HW*K,C+c], nbf [B,HW*K,c]); see i2p_pc_rows_fwd"""
        B, HW, _ = xyz.shape
        Cp, cf = pts.shape[2], feat.shape[2]
        dev = xyz.device
        geo = torch.empty(B, HW * K, 12, dtype=_F32, device=dev)
        part = torch.empty(B, HW * K, Cp + cf, dtype=_F32, device=dev)
        nbf = torch.empty(B, HW * K, cf, dtype=_F32, device=dev)
        self._call("i2p_pc_rows_fwd", int(B), int(HW), int(K), int(W), int(Cp), int(cf), self._p(xyz, _F32, "xyz"), self._p(pts, _F32, "pts"),
                   self._p(feat, _F32, "feat"), self._p(h_idx, _I64, "h_idx"), self._p(w_idx, _I64, "w_idx"), self._p(geo, _F32, "geo"),
                   self._p(part, _F32, "part"), self._p(nbf, _F32, "nbf"), stream=self._stream())
        return geo, part, nbf

    def pc_rows_backward(self, xyz, h_idx, w_idx, K, W, Cp, cf, g_geo, g_part, g_nbf):
        """-> (d_pts [B,HW,C], comb [B,HW,c+4] = [d_feat | d_xyz | 0] after the row scatter)"""
        B, HW, _ = xyz.shape
        dev = xyz.device
        d_pts = torch.empty(B, HW, Cp, dtype=_F32, device=dev)
        comb = torch.empty(B, HW, cf + 4, dtype=_F32, device=dev)
        rows = torch.empty(B, HW * K, cf + 4, dtype=_F32, device=dev)
        opt = lambda t: self._p(t, _F32, "g") if t is not None else None
        self._call("i2p_pc_rows_bwd", int(B), int(HW), int(K), int(W), int(Cp), int(cf), self._p(xyz, _F32, "xyz"), self._p(h_idx, _I64, "h_idx"),
                   self._p(w_idx, _I64, "w_idx"), opt(g_geo), self._p(g_part, _F32, "g_part"), opt(g_nbf), self._p(d_pts, _F32, "d_pts"),
                   self._p(comb, _F32, "comb"), self._p(rows, _F32, "rows"), stream=self._stream())
        self.gather_rows_grad(rows, h_idx, w_idx, W, comb)             # deterministic fixed-point scatter of the per-neighbour parts
        return d_pts, comb

    def max_response_forward(self, pts, pix, valid):
        """pts [B,N,C], pix [B,M,C], valid [B,N] -> (respond [B,M,C], fmaxmin [B,2,C], imaxmin i32 [B,2,C], anyv i32 [B])"""
        B, N, Cc = pts.shape
        M = pix.shape[1]
        dev = pts.device
        respond = torch.empty(B, M, Cc, dtype=_F32, device=dev)
        fm = torch.empty(B, 2, Cc, dtype=_F32, device=dev); im = torch.empty(B, 2, Cc, dtype=_I32, device=dev)
        anyv = torch.empty(B, dtype=_I32, device=dev)
        self._call("i2p_max_response_fwd", int(B), int(N), int(M), int(Cc), self._p(pts, _F32, "pts"), self._p(pix, _F32, "pix"),
                   self._p(valid, _F32, "valid"), self._p(respond, _F32, "respond"), self._p(fm, _F32, "fmaxmin"), self._p(im, _I32, "imaxmin"),
                   self._p(anyv, _I32, "anyv"), stream=self._stream())
        return respond, fm, im, anyv

    def max_response_backward(self, g, pix, fm, im, anyv, N):
        B, M, Cc = pix.shape
        d_pts = torch.empty(B, N, Cc, dtype=_F32, device=g.device); d_pix = torch.empty_like(pix)
        self._call("i2p_max_response_bwd", int(B), int(N), int(M), int(Cc), self._p(g, _F32, "g"), self._p(pix, _F32, "pix"),
                   self._p(fm, _F32, "fmaxmin"), self._p(im, _I32, "imaxmin"), self._p(anyv, _I32, "anyv"), self._p(d_pts, _F32, "d_pts"),
                   self._p(d_pix, _F32, "d_pix"), stream=self._stream())
        return d_pts, d_pix

    def bn_finalize(self, rows, sums, gamma, beta, eps):
        """-> (coef [3,c] = mean, invstd*gamma, beta ; mean_invstd [2c])"""
        c = gamma.shape[0]
        coef = torch.empty(3, c, dtype=_F32, device=gamma.device)
        mean_invstd = torch.empty(2 * c, dtype=_F32, device=gamma.device)
        self._call("i2p_bn_finalize", int(rows), int(c), self._p(sums, torch.float64, "sums"),
                   self._p(gamma, _F32, "gamma"), self._p(beta, _F32, "beta"), float(eps), self._p(coef, _F32, "coef"),
                   self._p(mean_invstd, _F32, "mean_invstd"), stream=self._stream())
        return coef, mean_invstd


_hip = None
_active = None


def hip_backend():
    """The product backend: libi2p_ops.so on the current HIP device.  Raises if the library
    is missing (no CPU fallback)."""
    global _hip
    if _hip is None:
        _lib.load()
        _hip = CBackend(_lib.call, "cuda", "hip")
    return _hip


def get_backend():
    return _active if _active is not None else hip_backend()


def set_backend(backend):
    """Test / cpu-baseline hook only (see module docstring).  `None` restores the HIP backend."""
    global _active
    prev = _active
    _active = backend
    return prev
