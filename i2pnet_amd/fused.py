"""Fused execution of stacks of `Conv2d` blocks (1x1 conv -> batch-stat BN -> activation) on the
MFMA layer kernels of csrc/mlp.hip.

A stack of L blocks runs as L forward kernels (each applies the previous block's BN+activation on
load and accumulates its own output statistics) and L backward kernels; only the pre-BN tensors
are ever materialised.  The reference runs every block as permute / conv / BN / activation /
permute in eager PyTorch (src/projectPN/PPBackbone_center.py:34-46).
"""
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
    bwd = (cout_p * (cin_p + 1) + 64 * (cout_p + 1) + 64 * (cin_p + 1) + 5 * cout_p + 4 * cin_p) * 4
    fwd = (32 + 128) * (cin + 2) * 4
    return cin_p <= 160 and bwd <= _LDS and fwd <= _LDS


def _rep_sum(dsums, c):
    return dsums.view(ops.BN_REPLICAS, 2, c).sum(0)


class _MlpChain(Function):
    """x [rows,c0] (raw input, or a pre-BN tensor when `first_bn`) -> act(BN(...)) of the last block.

    params = (g0, b0)? + (W1, g1, b1, W2, g2, b2, ...); slopes[i] = activation slope of BN i
    (index 0 = the optional leading BN)."""

    @staticmethod
    def forward(ctx, x, first_bn, slopes, *params):
        be = ops.get_backend()
        rows = x.shape[0]
        p = list(params)
        coefs, mis, ys = [], [], [x]
        in_coef, slope_in = None, 1.0
        k = 0
        if first_bn:
            g0, b0 = p[0], p[1]; k = 2
            sums = torch.zeros(ops.BN_REPLICAS * 2 * x.shape[1], dtype=torch.float64, device=x.device)
            be._call("i2p_bn_stats", int(rows), int(x.shape[1]), be._p(x, torch.float32, "x"),
                     be._p(sums, torch.float64, "sums"), stream=be._stream())
            in_coef, mi = be.bn_finalize(rows, sums, g0.detach(), b0.detach(), _EPS)
            coefs.append(in_coef); mis.append(mi); slope_in = slopes[0]
        else:
            coefs.append(None); mis.append(None)
        nl = (len(p) - k) // 3
        sums = None
        for i in range(nl):
            W, g, b = p[k + 3 * i], p[k + 3 * i + 1], p[k + 3 * i + 2]
            y, sums = be.lin_forward(ys[-1], in_coef, slope_in, W.detach())
            in_coef, mi = be.bn_finalize(rows, sums, g.detach(), b.detach(), _EPS)
            coefs.append(in_coef); mis.append(mi); ys.append(y)
            slope_in = slopes[i + 1]
        # the stack's output: BN + activation of the last pre-BN tensor, materialised once
        out = torch.empty_like(ys[-1])
        mi_last = torch.empty_like(mis[-1])
        last_g, last_b = (p[-2], p[-1]) if nl else (p[0], p[1])
        if nl == 0:      # only the leading BN: recompute its sums for the apply kernel
            sums = torch.zeros(ops.BN_REPLICAS * 2 * x.shape[1], dtype=torch.float64, device=x.device)
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
        return out

    @staticmethod
    def backward(ctx, g_out):
        be = ops.get_backend()
        saved = list(ctx.saved_tensors)
        ys = saved[:ctx.n_ys]
        cf = saved[ctx.n_ys:ctx.n_ys + ctx.n_coef]
        ms = saved[ctx.n_ys + ctx.n_coef:ctx.n_ys + 2 * ctx.n_coef]
        p = saved[ctx.n_ys + 2 * ctx.n_coef:]
        first_bn, slopes, nl, k = ctx.first_bn, ctx.slopes, ctx.nl, ctx.k
        coefs = ([] if first_bn else [None]) + cf           # index i = BN behind layer i (0 = leading BN / none)
        mis = ([] if first_bn else [None]) + ms
        grads = [None] * len(p)
        g_out = g_out.contiguous()
        # BN + activation of the last block: dL/dy_L and its gamma/beta gradients
        last_g, last_b = (p[-2], p[-1]) if nl else (p[0], p[1])
        gz, dg, db = be.bn_act_backward(g_out, ys[-1], mis[-1], last_g.detach(), last_b.detach(), slopes[-1])
        if nl:
            grads[-2], grads[-1] = dg, db
        else:
            grads[0], grads[1] = dg, db
            return (gz if ctx.x_needs_grad else None), None, None, *grads
        y_out = out_coef = out_mi = out_ds = None           # top layer: gz already is dL/dy
        for i in range(nl, 0, -1):
            W = p[k + 3 * (i - 1)]
            has_in = coefs[i - 1] is not None
            need_gx = has_in or ctx.x_needs_grad
            gz_in, in_ds, dw = be.lin_backward(gz, y_out, out_coef, out_mi, out_ds, ys[i - 1], coefs[i - 1],
                                               mis[i - 1], slopes[i - 1] if has_in else 1.0, W.detach(),
                                               need_gx=need_gx)
            grads[k + 3 * (i - 1)] = dw
            if has_in:
                s = _rep_sum(in_ds, ys[i - 1].shape[1])
                if i - 1 >= 1:
                    grads[k + 3 * (i - 2) + 1], grads[k + 3 * (i - 2) + 2] = s[1].float(), s[0].float()
                else:
                    grads[0], grads[1] = s[1].float(), s[0].float()
            gz, y_out, out_coef, out_mi, out_ds = gz_in, ys[i - 1], coefs[i - 1], mis[i - 1], in_ds
        gx = None
        if ctx.x_needs_grad:
            if first_bn:       # finish the leading BN: dL/dx = scale*(gz - mean(gz) - xhat*mean(gz*xhat))
                gx, _, _ = _bn_bwd_from_gz(be, gz, ys[0], mis[0], p[0], p[1], out_ds)
            else:
                gx = gz
        return gx, None, None, *grads


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
        d_f, d_g, d_bn, d_bk, dw = ops.get_backend().pair_lin_backward(gy.contiguous(), f, g, W)
        return d_f, d_g, d_bn, d_bk, dw


def pair_linear(f, g, bias_n, bias_k, W):
    return _PairLinear.apply(f, g, bias_n, bias_k, W)


def pair_fits(cin, cout):
    return cin % 4 == 0 and cin <= 128 and layer_fits(cin, cout)


def _slope(conv):
    return conv.negative_slope if conv.activation_fn else 1.0


def mlp_stack(x, convs, first_bn=None):
    """Apply `convs` (list of modules.Conv2d with batch-stat BN) to channel-last `x [..., C]`.
    `first_bn`: a Conv2d whose BN+activation still has to be applied to `x` (x is its pre-BN output).
    Consecutive blocks that fit the fused kernels run as one chain; others run block by block."""
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
            if not (c.bn and not c.bn_linear.track_running_stats and layer_fits(cin_eff, c.out_channels)):
                break
            run.append(c); cin = c.out_channels; j += 1
        if run or pending_bn is not None:
            params, slopes = [], []
            if pending_bn is not None:
                params += [pending_bn.bn_linear.weight, pending_bn.bn_linear.bias]; slopes.append(_slope(pending_bn))
            else:
                slopes.append(1.0)
            xin = cur
            for t, c in enumerate(run):
                W = c.weight2d()
                if t == 0 and pending_bn is None and xin.shape[1] % 4:       # pad raw input channels to a multiple of 4
                    pad = 4 - xin.shape[1] % 4
                    xin = F.pad(xin, (0, pad)); W = F.pad(W, (0, pad))
                params += [W, c.bn_linear.weight, c.bn_linear.bias]; slopes.append(_slope(c))
            cur = _MlpChain.apply(xin.contiguous(), pending_bn is not None, tuple(slopes), *params)
            pending_bn = None
            i = j
        if i < n and not run:           # block that does not fit: library GEMM + fused BN/activation kernels
            cur = convs[i](cur)
            i += 1
    return cur.reshape(*lead, cur.shape[-1])
