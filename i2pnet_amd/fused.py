"""Fused execution of stacks of `Conv2d` blocks (1x1 conv -> batch-stat BN -> activation) on the
MFMA layer kernels of csrc/mlp.hip.

A stack of L blocks runs as L forward kernels (each applies the previous block's BN+activation on
load and accumulates its own output statistics) and L backward kernels; only the pre-BN tensors
are ever materialised.  The reference runs every block as permute / conv / BN / activation /
permute in eager PyTorch (src/projectPN/PPBackbone_center.py:34-46).
"""
import os

import contextlib

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import ops

_EPS = 1e-5
_LDS = 160 * 1024


def _r32(v):
    return (v + 31) // 32 * 32


def layer_fits(cin, cout):
    """shape limits of i2p_lin_fwd / i2p_lin_bwd (LDS-resident weights)."""
    if cin % 4 or cout % 4 or cout > 128:
        return False
    cin_p, cout_p = _r32(cin), _r32(cout)
    bwd = (cout_p * (cin_p + 1) + 64 * (cout_p + 1) + 64 * (cin_p + 1) + 6 * cout_p + 4 * cin_p) * 4
    fwd = (32 + 128) * (cin + 2) * 4
    return cin_p <= 160 and bwd <= _LDS and fwd <= _LDS


USE_BIG_LAYERS = os.environ.get("I2P_NO_BIG", "0") != "1"


def big_layer_fits(cin, cout):
    """wide layers on few rows (cin > 160 or cout > 128, up to 320 channels): the K-tiled kernels of csrc/mlp_big.hip behind the
    same i2p_lin_fwd / i2p_lin_bwd entries (device library only)"""
    be = ops.get_backend()
    return (USE_BIG_LAYERS and be.device_type == "cuda" and be.name == "hip" and cin % 4 == 0 and cout % 4 == 0
            and cin <= 320 and cout <= 320 and (cin > 128 or cout > 128))


class _LinearTN(Function):
    """x [rows, cin] @ w[cout, cin]^T with the weight gradient on i2p_gemm_tn (rows cut over the grid) instead of the BLAS's
    one-workgroup-per-tile tall-skinny product; forward and input gradient stay on rocBLAS (regular shapes)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        # (w is a column slice of a parameter here — autograd concatenates the slices' gradients during backward, i.e. READS this one at
        #  once: its slab reduction is not deferred)
        with ops.defer_paused():
            dw = ops.get_backend().gemm_tn(g, x) if ctx.needs_input_grad[1] else None
        return gx, dw


LINEAR_TN_MIN_ROWS = 1024


def linear(x, w):
    """F.linear(x, w) (no bias) for the layers outside the fused kernels: same forward; on the HIP backend and enough rows
    the weight gradient runs on `ops.gemm_tn`"""
    be = ops.get_backend()
    rows = x.numel() // max(x.shape[-1], 1)
    if (be.device_type == "cuda" and be.name == "hip" and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32
            and rows >= LINEAR_TN_MIN_ROWS and w.requires_grad and torch.is_grad_enabled()):
        y = _LinearTN.apply(x.reshape(rows, x.shape[-1]).contiguous(), w)
        return y.view(*x.shape[:-1], w.shape[0])
    return F.linear(x, w)


_BF = torch.bfloat16


def _p2(c):
    return c in (16, 32, 64, 128)


def chain_bf16_ok(x, first_bn, weights):
    """bf16 storage for this chain?  (ops.set_precision("bf16"), enough rows, shapes the bf16 kernels take:
    every output width in {16,32,64,128}; raw fp32 input of any width % 4 == 0 up to 160 — a power of two when its
    gradient is needed; no leading BN)"""
    if first_bn or not weights or not ops.bf16_rows_ok(x.shape[0], x.device) or x.dtype != torch.float32:
        return False
    cin = x.shape[1]
    if cin % 4 or cin > 160 or (x.requires_grad and not _p2(cin)):
        return False
    return all(_p2(W.shape[0]) for W in weights)


def _rep_sum(dsums, c, dtype=None):
    """[2, c] sums over the replicas; dtype=torch.float32: the cast happens inside the one reduction launch (the fp64 replica values
    are each rounded to fp32 first: 6e-8 relative, far below the 1e-3 gradient contract; sum + two casts were three launches)"""
    return dsums.view(ops.BN_REPLICAS, 2, c).sum(0, dtype=dtype)


def _update_running(running, idx, mi, rows):
    """BatchNorm2d bookkeeping for chains whose BNs keep running buffers: momentum update from this batch's mean /
    unbiased variance (mean_invstd from bn_finalize; the conv bias, left out of the GEMM, re-enters the mean)."""
    if not running or running[idx] is None:
        return
    bn, bias = running[idx]
    if not bn.track_running_stats or bn.momentum is None:
        return
    with torch.no_grad():
        c = mi.shape[0] // 2
        mean, invstd = mi[:c], mi[c:]
        var = (1.0 / (invstd * invstd) - _EPS).clamp_min(0.0)
        mb = mean if bias is None else mean + bias.detach()
        bn.running_mean.mul_(1 - bn.momentum).add_(mb * bn.momentum)
        bn.running_var.mul_(1 - bn.momentum).add_(var * (rows / max(rows - 1, 1)) * bn.momentum)
        bn.num_batches_tracked += 1


def _pad_cols(W, cpad):
    """[cout, cin] -> [cout, cpad] with zero columns: one launch on the device library (F.pad: a fill and a copy)"""
    be = ops.get_backend()
    if be.name == "hip" and W.is_cuda and W.dtype == torch.float32 and W.is_contiguous():
        return be.pad_cols(W, cpad)
    return F.pad(W, (0, cpad - W.shape[1]))


class _MlpChain(Function):
    """x [rows,c0] (raw input, or a pre-BN tensor when `first_bn`) -> act(BN(...)) of the last block.

    params = (g0, b0)? + (W1, g1, b1, W2, g2, b2, ...); slopes[i] = activation slope of BN i
    (index 0 = the optional leading BN)."""

    @staticmethod
    def forward(ctx, x, first_bn, slopes, pool_k, running, *params):
        """pool_k > 0: the output is the max over groups of pool_k consecutive rows (set-abstraction tail),
        [rows/pool_k, c]; the activated [rows, c] tensor is not materialised.
        `running`: None, or one entry per BN of the chain (index 0 = the leading BN): (BatchNorm module, conv bias)
        whose running buffers are updated from this batch's statistics (models with BatchNorm2d running stats)."""
        be = ops.get_backend()
        rows = x.shape[0]
        p = list(params)
        coefs, mis, ys = [], [], [x]
        in_coef, slope_in = None, 1.0
        k = 0
        if first_bn:
            g0, b0 = p[0], p[1]; k = 2
            sums = ops.zeros(ops.BN_REPLICAS * 2 * x.shape[1], torch.float64, x.device)
            be._call("i2p_bn_stats", int(rows), int(x.shape[1]), be._p(x, torch.float32, "x"),
                     be._p(sums, torch.float64, "sums"), stream=be._stream())
            in_coef, mi = be.bn_finalize(rows, sums, g0.detach(), b0.detach(), _EPS)
            _update_running(running, 0, mi, rows)
            coefs.append(in_coef); mis.append(mi); slope_in = slopes[0]
        else:
            coefs.append(None); mis.append(None)
        nl = (len(p) - k) // 3
        sums = None
        # input rows zero-padded beyond the first layer's real cin (multiple of 4 / power of two for the kernels): the weight is
        # padded HERE, outside autograd — its gradient is cut back in backward() where the wgrad runs (no ConstantPadNd node)
        ctx.w0_cin = None
        if nl and not first_bn and x.shape[1] > p[k].shape[1]:
            ctx.w0_cin = p[k].shape[1]
        bf = chain_bf16_ok(x, first_bn, [p[k + 3 * i] for i in range(nl)])
        ctx.bf16 = bf
        # few rows, 64-multiple widths: the whole chain (layers, BN finalisations, activation / max-over-K tail, weight pad) in ONE
        # launch on a resident grid (csrc/mlp_chain.hip); the node saves exactly what the layer-by-layer path saves
        if (nl and not first_bn and not bf and running is None and x.dtype == torch.float32 and x.is_cuda
                and (not pool_k or (256 % (params[-3].shape[0] // 4) == 0 and pool_k <= 255))
                and be.chain_fits(rows, [x.shape[1]] + [params[k + 3 * i].shape[0] for i in range(nl)], pool_k)):
            d = lambda t: t.detach()
            Ws = [d(params[k + 3 * i]) for i in range(nl)]
            # the backward of the chain in one launch too (csrc/mlp_chain.hip: chain_bwd_kernel) when its two LDS strips fit; otherwise the
            # layer-by-layer backward, which wants W_0 with its zero columns
            ctx.chain_bwd = be.chain_bwd_fits(rows, [x.shape[1]] + [w.shape[0] for w in Ws], pool_k)
            ys_, coefs_, mis_, out, arg, w0p = be.chain_forward(x, Ws, [d(p[k + 3 * i + 1]) for i in range(nl)],
                                                                [d(p[k + 3 * i + 2]) for i in range(nl)], slopes[1:], _EPS, pool_k,
                                                                ctx.w0_cin is not None and not ctx.chain_bwd)
            if w0p is not None:
                p[k] = w0p
            ys += ys_; coefs += coefs_; mis += mis_
            ctx.pool_k = pool_k
            ctx.first_bn, ctx.slopes, ctx.nl, ctx.k = first_bn, slopes, nl, k
            ctx.save_for_backward(*ys, *[c for c in coefs if c is not None], *[m for m in mis if m is not None], *p,
                                  *([arg] if pool_k else []))
            ctx.n_ys, ctx.n_coef = len(ys), len([c for c in coefs if c is not None])
            ctx.x_needs_grad = x.requires_grad
            return out
        if ctx.w0_cin is not None:
            p[k] = _pad_cols(p[k].detach(), x.shape[1])
        for i in range(nl):
            W, g, b = p[k + 3 * i], p[k + 3 * i + 1], p[k + 3 * i + 2]
            y, sums, in_coef, mi = be.lin_forward_fin(ys[-1], in_coef, slope_in, W.detach(), g.detach(), b.detach(), _EPS,
                                                      out_dtype=_BF if bf else torch.float32)
            _update_running(running, i + 1, mi, rows)
            coefs.append(in_coef); mis.append(mi); ys.append(y)
            slope_in = slopes[i + 1]
        ctx.pool_k = 0
        if pool_k and nl and ys[-1].shape[1] % 4 == 0 and 256 % (ys[-1].shape[1] // 4) == 0 and pool_k <= 255:
            out, arg = be.bn_act_maxk_forward(ys[-1], coefs[-1], slopes[-1], pool_k)
            ctx.pool_k = pool_k
            ctx.first_bn, ctx.slopes, ctx.nl, ctx.k = first_bn, slopes, nl, k
            ctx.save_for_backward(*ys, *[c for c in coefs if c is not None], *[m for m in mis if m is not None], *p, arg)
            ctx.n_ys, ctx.n_coef = len(ys), len([c for c in coefs if c is not None])
            ctx.x_needs_grad = x.requires_grad
            return out
        if bf:                                          # the stack's output (fp32) from the bf16 pre-BN tensor
            out = be.bn_act_apply_bf16(ys[-1], coefs[-1], slopes[-1])
            ctx.first_bn, ctx.slopes, ctx.nl, ctx.k = first_bn, slopes, nl, k
            ctx.save_for_backward(*ys, *[c for c in coefs if c is not None], *[m for m in mis if m is not None], *p)
            ctx.n_ys, ctx.n_coef = len(ys), len([c for c in coefs if c is not None])
            ctx.x_needs_grad = x.requires_grad
            if pool_k:
                raise RuntimeError("_MlpChain: pool_k on a shape the fused tail does not take (mlp_stack pools those outside the node)")
            return out
        # the stack's output: BN + activation of the last pre-BN tensor, materialised once
        out = torch.empty_like(ys[-1])
        mi_last = torch.empty_like(mis[-1])
        last_g, last_b = (p[-2], p[-1]) if nl else (p[0], p[1])
        if nl == 0:      # only the leading BN: recompute its sums for the apply kernel
            sums = ops.zeros(ops.BN_REPLICAS * 2 * x.shape[1], torch.float64, x.device)
            be._call("i2p_bn_stats", int(rows), int(x.shape[1]), be._p(x, torch.float32, "x"),
                     be._p(sums, torch.float64, "sums"), stream=be._stream())
        c_last = ys[-1].shape[1]
        be._call("i2p_bn_act_fwd", int(rows), int(c_last), be._p(ys[-1], torch.float32, "y"),
                 be._p(sums, torch.float64, "sums"), be._p(last_g.detach(), torch.float32, "g"),
                 be._p(last_b.detach(), torch.float32, "b"), _EPS, float(slopes[-1]), be._p(out, torch.float32, "out"),
                 be._p(mi_last, torch.float32, "mi"), stream=be._stream())
        ctx.first_bn, ctx.slopes, ctx.nl, ctx.k = first_bn, slopes, nl, k
        ctx.save_for_backward(*ys, *[c for c in coefs if c is not None], *[m for m in mis if m is not None], *p)
        ctx.n_ys, ctx.n_coef = len(ys), len([c for c in coefs if c is not None])
        ctx.x_needs_grad = x.requires_grad
        if pool_k:
            raise RuntimeError("_MlpChain: pool_k on a shape the fused tail does not take (mlp_stack pools those outside the node)")
        return out

    @staticmethod
    def backward(ctx, g_out):
        be = ops.get_backend()
        saved = list(ctx.saved_tensors)
        bf = getattr(ctx, "bf16", False)
        fused_ds = None
        if getattr(ctx, "chain_bwd", False):
            arg = saved.pop() if ctx.pool_k else None
            n_ys, n_cf, nl = ctx.n_ys, ctx.n_coef, ctx.nl
            ys, cf, ms, p = saved[:n_ys], saved[n_ys:n_ys + n_cf], saved[n_ys + n_cf:n_ys + 2 * n_cf], saved[n_ys + 2 * n_cf:]
            gx, dws, dgs, dbs = be.chain_backward(ys[0], [p[3 * i].detach() for i in range(nl)], ys[1:], cf, ms, ctx.slopes[1:],
                                                  g_out.contiguous(), arg, ctx.pool_k, ctx.x_needs_grad)
            grads = [None] * len(p)
            for i in range(nl):
                grads[3 * i], grads[3 * i + 1], grads[3 * i + 2] = dws[i], dgs[i], dbs[i]
            return gx, None, None, None, None, *grads
        if ctx.pool_k:
            arg = saved.pop()
            if (not bf and be.name == "hip" and ctx.nl and 256 % (g_out.shape[1] // 4) == 0):
                # dense dL/da AND the last block's BN-backward statistics in one pass (they only live on the arg-max rows)
                n_ys = ctx.n_ys
                last_y, last_mi = saved[n_ys - 1], saved[n_ys + 2 * ctx.n_coef - 1]
                g_out, fused_ds = be.unpool_k_stats(g_out.contiguous(), arg, ctx.pool_k, last_y, last_mi, saved[-2].detach(), saved[-1].detach(),
                                                    ctx.slopes[-1])
            else:
                g_out = be.unpool_k(g_out.contiguous(), arg, ctx.pool_k, dtype=_BF if bf else torch.float32)   # dense dL/da in one pass
        elif bf:
            g_out = be.to_bf16(g_out)
        ys = saved[:ctx.n_ys]
        cf = saved[ctx.n_ys:ctx.n_ys + ctx.n_coef]
        ms = saved[ctx.n_ys + ctx.n_coef:ctx.n_ys + 2 * ctx.n_coef]
        p = saved[ctx.n_ys + 2 * ctx.n_coef:]
        first_bn, slopes, nl, k = ctx.first_bn, ctx.slopes, ctx.nl, ctx.k
        coefs = ([] if first_bn else [None]) + cf           # index i = BN behind layer i (0 = leading BN / none)
        mis = ([] if first_bn else [None]) + ms
        grads = [None] * len(p)
        g_out = g_out.contiguous()
        last_g, last_b = (p[-2], p[-1]) if nl else (p[0], p[1])
        if not nl:          # only the leading BN: dL/dx and its gamma/beta gradients
            gz, dg, db = be.bn_act_backward(g_out, ys[-1], mis[-1], last_g.detach(), last_b.detach(), slopes[-1])
            grads[0], grads[1] = dg, db
            return (gz if ctx.x_needs_grad else None), None, None, None, None, *grads
        # last block: only the statistics pass over (dL/da, y_L); the activation derivative and the BN backward are
        # applied by the layer kernels as they load dL/da (slope_out), so dL/dy_L is never written
        if fused_ds is not None:
            out_ds = fused_ds
        elif bf:
            out_ds = be.bn_act_backward_stats_bf16(g_out, ys[-1], coefs[-1], mis[-1], slopes[-1])
        else:
            out_ds = be.bn_act_backward_stats(g_out, ys[-1], mis[-1], last_g.detach(), last_b.detach(), slopes[-1])
        gz, y_out, out_coef, out_mi, slope_out = g_out, ys[-1], coefs[-1], mis[-1], slopes[-1]
        for i in range(nl, 0, -1):
            W = p[k + 3 * (i - 1)]
            has_in = coefs[i - 1] is not None
            need_gx = has_in or ctx.x_needs_grad
            call = lambda: be.lin_backward(gz, y_out, out_coef, out_mi, out_ds, ys[i - 1], coefs[i - 1], mis[i - 1],
                                           slopes[i - 1] if has_in else 1.0, W.detach(), need_gx=need_gx, slope_out=slope_out)
            if i == 1 and ctx.w0_cin is not None:            # zero-padded input rows: only the real columns of dW are the gradient
                gz_in, in_ds, dw = ops.defer_compact(call, W.shape[0], W.shape[1], ctx.w0_cin)
            else:
                gz_in, in_ds, dw = call()
            grads[k + 3 * (i - 1)] = dw
            # gamma/beta gradients of the BN behind layer i: reduced from out_ds by the launcher (scratch tail)
            grads[k + 3 * (i - 1) + 1], grads[k + 3 * (i - 1) + 2] = be.take_bn_grads()
            gz, y_out, out_coef, out_mi, out_ds, slope_out = gz_in, ys[i - 1], coefs[i - 1], mis[i - 1], in_ds, 1.0
        if first_bn:            # leading BN: its sums were accumulated by the last dgrad call
            s = _rep_sum(out_ds, ys[0].shape[1])
            grads[0], grads[1] = s[1].float(), s[0].float()
        gx = None
        if ctx.x_needs_grad:
            if first_bn:       # finish the leading BN: dL/dx = scale*(gz - mean(gz) - xhat*mean(gz*xhat))
                gx, _, _ = _bn_bwd_from_gz(be, gz, ys[0], mis[0], p[0], p[1], out_ds)
            else:
                gx = gz
        return gx, None, None, None, None, *grads


def _bn_bwd_from_gz(be, gz, y, mi, gamma, beta, dsums):
    rows, c = y.shape
    dy = torch.empty_like(y)
    dg = torch.empty(c, dtype=torch.float32, device=y.device)
    db = torch.empty(c, dtype=torch.float32, device=y.device)
    be._call("i2p_bn_act_bwd", int(rows), int(c), be._p(gz, torch.float32, "gz"), be._p(y, torch.float32, "y"),
             be._p(mi, torch.float32, "mi"), be._p(gamma.detach(), torch.float32, "g"),
             be._p(beta.detach(), torch.float32, "b"), 1.0, be._p(dsums, torch.float64, "dsums"),
             be._p(dy, torch.float32, "dy"), be._p(dg, torch.float32, "dg"), be._p(db, torch.float32, "db"),
             stream=be._stream())
    return dy, dg, db


class _PairLinear(Function):
    """y[b,n,k,:] = (f[b,n,:]*g[b,k,:]) . W^T + bias_n[b,n,:] + bias_k[b,k,:]  ->  [B*N*M, Co] (pre-BN)."""

    @staticmethod
    def forward(ctx, f, g, bias_n, bias_k, W):
        f, g, bias_n, bias_k, W = [t.contiguous() for t in (f, g, bias_n, bias_k, W)]
        y, _ = ops.get_backend().pair_lin_forward(f.detach(), g.detach(), bias_n.detach(), bias_k.detach(), W.detach())
        ctx.save_for_backward(f, g, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        f, g, W = ctx.saved_tensors
        with ops.defer_paused():          # (W = a column slice of the first layer's weight: its gradient is concatenated during backward)
            d_f, d_g, d_bn, d_bk, dw = ops.get_backend().pair_lin_backward(gy.contiguous(), f, g, W)
        return d_f, d_g, d_bn, d_bk, dw


def pair_linear(f, g, bias_n, bias_k, W):
    return _PairLinear.apply(f, g, bias_n, bias_k, W)


def pair_fits(cin, cout):
    return cin % 4 == 0 and cin <= 128 and layer_fits(cin, cout)


class _CvPiTail(Function):
    """The all-pixel pi-stage of the cost volume as one autograd node on pre-BN tensors:

        y1[b,n,k,:] = (f[b,n,:]*g[b,k,:]) . W1^T + bias_n[b,n,:] + bias_k[b,k,:]      (factored first layer)
        y1 -(bn1,act)-> W2 -> y2 -(bn2,act)-> W3 -> y3
        ye[b,n,k,:] = enc_n[b,n,:] + enc_k[b,k,:]                                     (position encoding, pre-BN)
        [ye -(bne,act) | y3 -(bn3,act)] -> W4 -> y4 -(bn4,act)-> W5 -> y5
        out[b,n,:] = sum_k softmax_k(act(bn5(y5))) * act(bn3(y3))

    (PPBackbone_center.py:383-433).  Nothing but y1..y5, ye and their statistics is materialised: no
    point x pixel product, no activation tensors, no concatenation, no softmax tensor; in the backward
    neither dL/dy1 nor dL/dye is written (BN backward on load in the pair kernel / in closed form on the
    encoding factors)."""

    @staticmethod
    def forward(ctx, f, g, bias_n, bias_k, W1, enc_n, enc_k, slopes, running, g1, b1, W2, g2, b2, W3, g3, b3, ge, be, W4,
                g4, b4, W5, g5, b5):
        be_ = ops.get_backend()
        f, g, bias_n, bias_k, W1, enc_n, enc_k = [t.detach().contiguous() for t in (f, g, bias_n, bias_k, W1, enc_n, enc_k)]
        B, N, M = f.shape[0], f.shape[1], g.shape[1]
        rows = B * N * M
        s1, s2, s3, se, s4, s5 = slopes
        d = lambda t: t.detach()
        widths = (W1.shape[0], W2.shape[0], W3.shape[0], enc_n.shape[-1], W4.shape[0], W5.shape[0])
        bf = (ops.bf16_rows_ok(rows, f.device) and all(_p2(c) for c in widths) and f.shape[-1] in (32, 64, 128)
              and W1.shape[0] in (32, 64, 128))
        dt = _BF if bf else torch.float32
        y1, st1, c1, m1 = be_.pair_lin_forward_fin(f, g, bias_n, bias_k, W1, d(g1), d(b1), _EPS, out_dtype=dt)
        y2, st2, c2, m2 = be_.lin_forward_fin(y1, c1, s1, d(W2), d(g2), d(b2), _EPS, out_dtype=dt)
        y3, st3, c3, m3 = be_.lin_forward_fin(y2, c2, s2, d(W3), d(g3), d(b3), _EPS, out_dtype=dt)
        if bf:
            ye, ste = be_.outer_sum_bf16(enc_n, enc_k)
        elif be_.device_type == "cuda" and be_.name == "hip" and enc_n.shape[-1] % 8 == 0:
            ye, ste = be_.outer_sum(enc_n, enc_k)              # the broadcast sum and its statistics in one pass
        else:
            ye = (enc_n.unsqueeze(2) + enc_k.unsqueeze(1)).view(rows, -1)
            ste = be_.bn_stats(ye)
        ce, me = be_.bn_finalize(rows, ste, d(ge), d(be), _EPS)
        y4, st4, c4, m4 = be_.lin_forward_2src_fin(ye, ce, se, y3, c3, s3, d(W4), d(g4), d(b4), _EPS)
        y5, st5, c5, m5 = be_.lin_forward_fin(y4, c4, s4, d(W5), d(g5), d(b5), _EPS, out_dtype=dt)
        for i_, m_ in enumerate((m1, m2, m3, me, m4, m5)):       # BatchNorm2d running buffers (small-range model)
            _update_running(running, i_, m_, rows)
        out, msave = be_.cv_softmax_wsum_forward(B, N, M, y5, c5, s5, y3, c3, s3)
        ctx.save_for_backward(y1, ye, y2, y3, y4, y5, c1, m1, c2, m2, c3, m3, ce, me, c4, m4, c5, m5, out, msave,
                              W2, W3, W4, W5, f, g, W1, enc_n, enc_k)
        ctx.dims, ctx.slopes = (B, N, M), slopes
        return out

    @staticmethod
    def backward(ctx, g_out):
        be_ = ops.get_backend()
        (y1, ye, y2, y3, y4, y5, c1, m1, c2, m2, c3, m3, ce, me, c4, m4, c5, m5, out, msave,
         W2, W3, W4, W5, f, g, W1, enc_n, enc_k) = ctx.saved_tensors
        B, N, M = ctx.dims
        s1, s2, s3, se, s4, s5 = ctx.slopes
        d = lambda t: t.detach()
        f32 = lambda t: t.float()
        gz5, ds5, ga3 = be_.cv_softmax_wsum_backward(B, N, M, g_out.contiguous(), out, msave, y5, c5, m5, s5, y3, c3, s3)
        # (gamma/beta gradients of a BN come back reduced from the call that consumes its sums: `last_bn_grads`)
        gz4, ds4, dW5 = be_.lin_backward(gz5, y5, c5, m5, ds5, y4, c4, m4, s4, d(W5)); dg5, db5 = be_.take_bn_grads()
        gze, dse, gz3, ds3, dW4 = be_.lin_backward_2src(gz4, y4, c4, m4, ds4, ye, ce, me, se, y3, c3, m3, s3, ga3, d(W4))
        dg4, db4 = be_.take_bn_grads()
        re = _rep_sum(dse, ye.shape[1], torch.float32)
        gz2, ds2, dW3 = be_.lin_backward(gz3, y3, c3, m3, ds3, y2, c2, m2, s2, d(W3)); dg3, db3 = be_.take_bn_grads()
        gz1, ds1, dW2 = be_.lin_backward(gz2, y2, c2, m2, ds2, y1, c1, m1, s1, d(W2)); dg2, db2 = be_.take_bn_grads()
        r1 = _rep_sum(ds1, y1.shape[1], torch.float32)
        # first layer: BN backward of bn1 formed on load inside the pair kernel
        with ops.defer_paused():          # (W1 = a column slice of mlp1_convs[0].weight: its gradient is concatenated during backward)
            d_f, d_g, d_bn, d_bk, dW1 = be_.pair_lin_backward(gz1, f, g, W1, y=y1, out_coef=c1, out_mi=m1, out_dsums=ds1)
        # position encoding: k-/n-sums of dL/dye in closed form from one pass over gz_e
        d_en, d_ek = be_.pair_bias_bn_backward(B, N, M, gze, enc_n, enc_k, dse, ce, me)
        return (d_f, d_g, d_bn, d_bk, dW1, d_en, d_ek, None, None,
                r1[1], r1[0], dW2, dg2, db2, dW3, dg3, db3,
                re[1], re[0], dW4, dg4, db4, dW5, dg5, db5)


class _CvKnnTail(Function):
    """The pi-stage of the kNN cost volume (PPBackbone_center.py:367-433 with nsample_q > 0) as one autograd node:
    the same tail as `_CvPiTail`, with materialised first-layer inputs x1 [rows, cin1] (geometry + correlation,
    zero-padded to a multiple of 4) and xe [rows, cine] (geometry) instead of the factored pair form.  No
    activated tensor, no concatenation and no softmax tensor is written; rows = B*N*K, softmax over K."""

    @staticmethod
    def forward(ctx, x1, xe, W1, We, dims, slopes, running, g1, b1, W2, g2, b2, W3, g3, b3, ge, be, W4, g4, b4, W5, g5, b5):
        be_ = ops.get_backend()
        B, N, K = dims
        rows = B * N * K
        s1, s2, s3, se, s4, s5 = slopes
        d = lambda t: t.detach()
        x1, xe, W1, We = [t.detach().contiguous() for t in (x1, xe, W1, We)]
        ctx.cins = (W1.shape[1], We.shape[1])            # real input widths: the rows carry zero padding, the weights get it here
        if x1.shape[1] > W1.shape[1]:
            W1 = _pad_cols(W1, x1.shape[1])
        if xe.shape[1] > We.shape[1]:
            We = _pad_cols(We, xe.shape[1])
        y1, st1, c1, m1 = be_.lin_forward_fin(x1, None, 1.0, W1, d(g1), d(b1), _EPS)
        y2, st2, c2, m2 = be_.lin_forward_fin(y1, c1, s1, d(W2), d(g2), d(b2), _EPS)
        y3, st3, c3, m3 = be_.lin_forward_fin(y2, c2, s2, d(W3), d(g3), d(b3), _EPS)
        ye, ste, ce, me = be_.lin_forward_fin(xe, None, 1.0, We, d(ge), d(be), _EPS)
        y4, st4, c4, m4 = be_.lin_forward_2src_fin(ye, ce, se, y3, c3, s3, d(W4), d(g4), d(b4), _EPS)
        y5, st5, c5, m5 = be_.lin_forward_fin(y4, c4, s4, d(W5), d(g5), d(b5), _EPS)
        for i_, m_ in enumerate((m1, m2, m3, me, m4, m5)):
            _update_running(running, i_, m_, rows)
        out, msave = be_.cv_softmax_wsum_forward(B, N, K, y5, c5, s5, y3, c3, s3)
        ctx.save_for_backward(y1, ye, y2, y3, y4, y5, c1, m1, c2, m2, c3, m3, ce, me, c4, m4, c5, m5, out, msave,
                              W2, W3, W4, W5, x1, xe, W1, We)
        ctx.dims, ctx.slopes = dims, slopes
        ctx.need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return out

    @staticmethod
    def backward(ctx, g_out):
        be_ = ops.get_backend()
        (y1, ye, y2, y3, y4, y5, c1, m1, c2, m2, c3, m3, ce, me, c4, m4, c5, m5, out, msave,
         W2, W3, W4, W5, x1, xe, W1, We) = ctx.saved_tensors
        B, N, K = ctx.dims
        s1, s2, s3, se, s4, s5 = ctx.slopes
        d = lambda t: t.detach()
        gz5, ds5, ga3 = be_.cv_softmax_wsum_backward(B, N, K, g_out.contiguous(), out, msave, y5, c5, m5, s5, y3, c3, s3)
        gz4, ds4, dW5 = be_.lin_backward(gz5, y5, c5, m5, ds5, y4, c4, m4, s4, d(W5)); dg5, db5 = be_.take_bn_grads()
        gze, dse, gz3, ds3, dW4 = be_.lin_backward_2src(gz4, y4, c4, m4, ds4, ye, ce, me, se, y3, c3, m3, s3, ga3, d(W4))
        dg4, db4 = be_.take_bn_grads()
        gz2, ds2, dW3 = be_.lin_backward(gz3, y3, c3, m3, ds3, y2, c2, m2, s2, d(W3)); dg3, db3 = be_.take_bn_grads()
        gz1, ds1, dW2 = be_.lin_backward(gz2, y2, c2, m2, ds2, y1, c1, m1, s1, d(W2)); dg2, db2 = be_.take_bn_grads()
        call1 = lambda: be_.lin_backward(gz1, y1, c1, m1, ds1, x1, None, None, 1.0, W1, need_gx=ctx.need[0])
        dx1, _, dW1 = ops.defer_compact(call1, W1.shape[0], W1.shape[1], ctx.cins[0]) if W1.shape[1] > ctx.cins[0] else call1()
        dg1, db1 = be_.take_bn_grads()
        calle = lambda: be_.lin_backward(gze, ye, ce, me, dse, xe, None, None, 1.0, We, need_gx=ctx.need[1])
        dxe, _, dWe = ops.defer_compact(calle, We.shape[0], We.shape[1], ctx.cins[1]) if We.shape[1] > ctx.cins[1] else calle()
        dge, dbe = be_.take_bn_grads()
        return (dx1, dxe, dW1, dWe, None, None, None, dg1, db1, dW2, dg2, db2, dW3, dg3, db3, dge, dbe, dW4, dg4, db4, dW5, dg5, db5)


def cv_knn_tail(x1, xe, dims, first, mlp1_rest, enc, mlp2):
    """x1 [B,N,K,cin1] / xe [B,N,K,cine] (channel counts already padded to a multiple of 4) -> pi_feat [B,N,c]"""
    c2, c3 = mlp1_rest
    c4, c5 = mlp2
    slopes = tuple(_slope(m) for m in (first, c2, c3, enc, c4, c5))
    bn = lambda m: (m.bn_linear.weight, m.bn_linear.bias)
    running = _running_list((first, c2, c3, enc, c4, c5))
    rows = dims[0] * dims[1] * dims[2]
    W1, We = first.weight2d(), enc.weight2d()
    return _CvKnnTail.apply(x1.reshape(rows, -1), xe.reshape(rows, -1), W1, We, dims, slopes, running, *bn(first), c2.weight2d(), *bn(c2),
                            c3.weight2d(), *bn(c3), *bn(enc), c4.weight2d(), *bn(c4), c5.weight2d(), *bn(c5))


def cv_pi_tail(f, g, bias_n, bias_k, W1, enc_n, enc_k, first, mlp1_rest, enc, mlp2):
    """f [B,N,C] / g [B,M,C] normalised point / pixel features, bias_n/bias_k the per-point / per-pixel parts of
    the first layer, W1 its bilinear weight block, enc_n/enc_k the factors of the position encoding
    -> pi_feat [B,N,c]"""
    c2, c3 = mlp1_rest
    c4, c5 = mlp2
    slopes = tuple(_slope(m) for m in (first, c2, c3, enc, c4, c5))
    bn = lambda m: (m.bn_linear.weight, m.bn_linear.bias)
    return _CvPiTail.apply(f, g, bias_n, bias_k, W1, enc_n, enc_k, slopes, _running_list((first, c2, c3, enc, c4, c5)), *bn(first), c2.weight2d(), *bn(c2),
                           c3.weight2d(), *bn(c3), *bn(enc), c4.weight2d(), *bn(c4), c5.weight2d(), *bn(c5))


_IDENT = {}


def _identity_coef(c, device):
    key = (c, str(device))
    t = _IDENT.get(key)
    if t is None:
        coef = torch.stack([torch.zeros(c), torch.ones(c), torch.zeros(c)]).to(device).contiguous()
        mi = torch.cat([torch.zeros(c), torch.ones(c)]).to(device).contiguous()
        t = _IDENT[key] = (coef, mi)
    return t


class _SoftmaxPool(Function):
    """out[b,:] = sum_n softmax_n(mask[b,n,:]) * value[b,n,:]  (PoseHead, PPBackbone_center.py:551-552) on the
    cost-volume softmax-weighted-sum kernels with identity BN coefficients: one launch each way instead of
    softmax / mul / sum and their five autograd nodes."""

    @staticmethod
    def forward(ctx, mask, value):
        B, N, C = mask.shape
        be_ = ops.get_backend()
        coef, mi = _identity_coef(C, mask.device)
        m2, v2 = mask.detach().reshape(B * N, C).contiguous(), value.detach().reshape(B * N, C).contiguous()
        out, msave = be_.cv_softmax_wsum_forward(B, 1, N, m2, coef, 1.0, v2, coef, 1.0)
        ctx.save_for_backward(m2, v2, out, msave)
        ctx.dims = (B, N, C)
        return out.view(B, 1, C)

    @staticmethod
    def backward(ctx, g):
        m2, v2, out, msave = ctx.saved_tensors
        B, N, C = ctx.dims
        be_ = ops.get_backend()
        coef, mi = _identity_coef(C, m2.device)
        gmask, _, gval = be_.cv_softmax_wsum_backward(B, 1, N, g.reshape(B, 1, C).contiguous(), out, msave, m2, coef, mi, 1.0,
                                                      v2, coef, 1.0)
        return gmask.view(B, N, C), gval.view(B, N, C)


def softmax_pool(mask, value):
    """mask, value [B,N,C] -> [B,1,C]; C must divide 256 (else use the torch formulation)"""
    return _SoftmaxPool.apply(mask, value)


class _SoftmaxWsumK(Function):
    """out[b,n,:] = sum_k softmax_k(logit[b,n,k,:]) * value[b,n,k,:] over the K neighbours of a point (pc-stage of the cost
    volume, PPBackbone_center.py:481-487) on the same kernels (groups = B*N, M = K): one launch each way instead of
    softmax / mul / sum and their autograd nodes."""

    @staticmethod
    def forward(ctx, logit, value):
        B, N, K, C = logit.shape
        be_ = ops.get_backend()
        coef, mi = _identity_coef(C, logit.device)
        m2, v2 = logit.detach().reshape(B * N * K, C).contiguous(), value.detach().reshape(B * N * K, C).contiguous()
        out, msave = be_.cv_softmax_wsum_forward(B, N, K, m2, coef, 1.0, v2, coef, 1.0)
        ctx.save_for_backward(m2, v2, out, msave)
        ctx.dims = (B, N, K, C)
        return out.view(B, N, C)

    @staticmethod
    def backward(ctx, g):
        m2, v2, out, msave = ctx.saved_tensors
        B, N, K, C = ctx.dims
        be_ = ops.get_backend()
        coef, mi = _identity_coef(C, m2.device)
        glog, _, gval = be_.cv_softmax_wsum_backward(B, N, K, g.reshape(B, N, C).contiguous(), out, msave, m2, coef, mi, 1.0,
                                                     v2, coef, 1.0)
        return glog.view(B, N, K, C), gval.view(B, N, K, C)


def softmax_wsum_k(logit, value):
    """logit, value [B,N,K,C] -> [B,N,C]; C must divide 256 (else use the torch formulation)"""
    return _SoftmaxWsumK.apply(logit, value)


def cv_tail_fits(first, mlp1_rest, enc, mlp2):
    if len(mlp1_rest) != 2 or len(mlp2) != 2:
        return False
    c2, c3 = mlp1_rest
    c4, c5 = mlp2
    p2 = lambda c: c in (16, 32, 64, 128)
    chans = [first.out_channels, c2.out_channels, c3.out_channels, enc.out_channels, c4.out_channels, c5.out_channels]
    ok = all(p2(c) for c in chans) and c4.in_channels == enc.out_channels + c3.out_channels and c5.out_channels == c3.out_channels
    return ok and 256 % c5.out_channels == 0 and all(m.bn and _batch_stat(m) for m in (first, c2, c3, enc, c4, c5))


def _running_list(convs):
    r = [(m.bn_linear, m.conv.bias) if m.bn_linear.track_running_stats else None for m in convs]
    return r if any(x is not None for x in r) else None


def _batch_stat(conv):
    """the conv's BN normalises with THIS batch's statistics: always for the projection model's BNs
    (track_running_stats False), in training mode for BatchNorm2d with running buffers"""
    bn = conv.bn_linear
    return (not bn.track_running_stats) or (bn.training and bn.momentum is not None)


def _slope(conv):
    return conv.negative_slope if conv.activation_fn else 1.0


def mlp_stack(x, convs, first_bn=None, pool_k=0):
    """Apply `convs` (list of modules.Conv2d with batch-stat BN) to channel-last `x [..., C]`.
    `first_bn`: a Conv2d whose BN+activation still has to be applied to `x` (x is its pre-BN output).
    Consecutive blocks that fit the fused kernels run as one chain; others run block by block.
    `pool_k`: x is `[..., K, C]` with K = pool_k and the result is the max over that axis `[..., C']`
    (fused into the last chain when it ends the stack)."""
    if first_bn is not None and not _batch_stat(first_bn):      # running-statistics BN in eval mode: plain affine
        x, first_bn = first_bn.finish(x), None
    lead = x.shape[:-1]
    cur = x.reshape(-1, x.shape[-1])
    pending_bn = first_bn
    i, n = 0, len(convs)
    while i < n or pending_bn is not None:
        run = []
        cin = cur.shape[1]
        j = i
        while j < n:
            c = convs[j]
            cin_eff = (cin + 3) // 4 * 4 if not run and pending_bn is None else cin
            if not (c.bn and _batch_stat(c) and (layer_fits(cin_eff, c.out_channels) or big_layer_fits(cin_eff, c.out_channels))):
                break
            run.append(c); cin = c.out_channels; j += 1
        if run or pending_bn is not None:
            params, slopes = [], []
            keeps = lambda m: (m.bn_linear, m.conv.bias) if m.bn_linear.track_running_stats else None
            running = [keeps(pending_bn) if pending_bn is not None else None] + [keeps(c) for c in run]
            if not any(r is not None for r in running):
                running = None
            if pending_bn is not None:
                params += [pending_bn.bn_linear.weight, pending_bn.bn_linear.bias]; slopes.append(_slope(pending_bn))
            else:
                slopes.append(1.0)
            xin = cur
            for t, c in enumerate(run):
                W = c.weight2d()
                if t == 0 and pending_bn is None:
                    if xin.shape[1] % 4:                 # pad raw input channels to a multiple of 4
                        xin = F.pad(xin, (0, 4 - xin.shape[1] % 4))
                    # (callers may deliver the zero channels already: modules.cat_padded; the weight's zero columns are added
                    # inside the chain, outside autograd)
                params += [W, c.bn_linear.weight, c.bn_linear.bias]; slopes.append(_slope(c))
            # the fused BN + activation + max-over-K tail takes widths whose float4 count divides 256 and K <= 255; other shapes
            # pool outside the node (plain torch.max below, autograd keeps its own arg-max)
            c_last = run[-1].out_channels if run else 0
            pool_ok = bool(run) and c_last % 4 == 0 and 256 % (c_last // 4) == 0 and pool_k <= 255
            pool_here = pool_k if (j >= n and pool_ok) else 0
            cur = _MlpChain.apply(xin.contiguous(), pending_bn is not None, tuple(slopes), pool_here, running, *params)
            if pool_here:
                return cur.reshape(*lead[:-1], cur.shape[-1])
            pending_bn = None
            i = j
        if i < n and not run:           # block that does not fit: library GEMM + fused BN/activation kernels
            cur = convs[i](cur[:, :convs[i].in_channels] if cur.shape[1] > convs[i].in_channels else cur)
            i += 1
    cur = cur.reshape(*lead, cur.shape[-1])
    return torch.max(cur, dim=-2)[0] if pool_k else cur
