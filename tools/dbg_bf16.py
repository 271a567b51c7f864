import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import test_bf16_gpu as T
hip = T._hip()
for (cin, cout, coef) in [(32, 64, True), (32, 64, False), (32, 32, True), (64, 64, True), (128, 128, True)]:
    rows = 3 * 4096 + 77
    x = T._rnd(rows, cin, seed=1).to(T.BF)
    w = T._rnd(cout, cin, seed=2, scale=cin ** -0.5)
    cf = T._coef(cin, 3)[0] if coef else None
    y, sums = hip.lin_forward(x, cf, 0.1, w, out_dtype=T.BF)
    a = T._bn_act(x, cf, 0.1)[0] if coef else x.float()
    want64 = T._bfr(a).double() @ T._bfr(w).double().t()
    want = T._bfr(want64.float())
    d = (y.float() - want).abs()
    ulp = torch.pow(2.0, torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 7)
    k = d / ulp
    bad = k > 1.01
    print(cin, cout, coef, "n>1ulp", int(bad.sum()), "max ulps", float(k.max()), "max abs", float(d.max()))
    if bad.any():
        idx = bad.nonzero()
        print(" rows", idx[:10, 0].tolist(), "cols", idx[:10, 1].tolist())
        print(" rows mod 32 hist", torch.bincount(idx[:, 0] % 32, minlength=32).tolist())
        print(" cols hist", torch.bincount(idx[:, 1], minlength=cout).tolist())
        i, j = idx[0].tolist()
        print(" got", float(y[i, j]), "want", float(want[i, j]), "want64", float(want64[i, j]))
        # recompute that element with fp32 a (unrounded)
        print(" unrounded-a", float((a[i].double() * T._bfr(w)[j].double()).sum()))
