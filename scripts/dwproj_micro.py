#!/usr/bin/env python3
"""Fused q/k/v projection kernels (dwproj.hip) vs the per-projection kernels (conv.hip) at the three CvT-21 stage shapes, batch 32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cxrmate_amd import ops

BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.4e9 * 0.03))            # ~30 ms of GPU idle-spin: the host enqueues everything behind it, events see GPU time only
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for Bn, C, H, tok0 in [(64, 64, 96, 0), (64, 192, 48, 0), (64, 384, 24, 1)]:
    W = H
    x = (torch.randn(Bn, tok0 + H * W, C, device="cuda") + 0.2).to(BF)
    strides = (1, 2, 2)
    par = [dict(w=torch.randn(C, 9, device="cuda") * 0.3, g=1 + 0.1 * torch.randn(C, device="cuda"), b=0.1 * torch.randn(C, device="cuda"),
                rm=torch.zeros(C, device="cuda"), rv=torch.ones(C, device="cuda")) for _ in range(3)]
    raws = [p["w"].t().contiguous() for p in par]
    sp = [dict(stride=s, taps=r, w=p["w"], gamma=p["g"], beta=p["b"], run_mean=p["rm"], run_var=p["rv"]) for s, r, p in zip(strides, raws, par)]
    st = ops.dwproj_bn_train_stats(x, H, W, tok0, 1e-5, 0.1, sp)
    ys = ops.dwproj_apply(x, H, W, tok0, st)
    dys = [torch.randn_like(y) for y in ys]
    dg = [torch.zeros(C, device="cuda") for _ in par]; db = [torch.zeros(C, device="cuda") for _ in par]; dw = [torch.zeros(C, 9, device="cuda") for _ in par]
    bp = [dict(stride=s, taps=r, y=d, gamma=p["g"], mean=t["mean"], rstd=t["rstd"], dgamma=a, dbeta=b) for s, r, d, p, t, a, b in zip(strides, raws, dys, par, st, dg, db)]
    coefs = ops.dwproj_bn_train_bwd_stats(x, H, W, tok0, bp)
    cp = [dict(stride=s, taps=r, y=d, coef=cf, dw=w_) for s, r, d, cf, w_ in zip(strides, raws, dys, coefs, dw)]
    xp = [dict(stride=s, taps=r, y=d) for s, r, d in zip(strides, raws, dys)]
    mb = x.numel() * 2 / 1e6
    t = dict(stats=timeit(lambda: ops.dwproj_bn_train_stats(x, H, W, tok0, 1e-5, 0.1, sp)), apply=timeit(lambda: ops.dwproj_apply(x, H, W, tok0, st)),
             bwd_stats=timeit(lambda: ops.dwproj_bn_train_bwd_stats(x, H, W, tok0, bp)),
             bwd_stats_y=timeit(lambda: ops.dwproj_bn_train_bwd_stats(x, H, W, tok0, [dict(b_, beta=p_["b"], yf=y_) for b_, p_, y_ in zip(bp, par, ys)])), dc_taps=timeit(lambda: ops.dwproj_dc_taps_(x, H, W, tok0, cp)),
             dx=timeit(lambda: ops.dwproj_dx(xp, Bn, C, H, W, tok0)))
    # per-projection path of conv.hip (what the encoder used before)
    op = [dict(wt=r, w=p["w"], g=p["g"], b=p["b"], run_mean=p["rm"], run_var=p["rv"]) for r, p in zip(raws, par)]

    def old_fwd():
        ops.dwconv_bn_train_fwd_stats(x, H, W, 1, tok0, 1e-5, 0.1, op[:1]); ops.dwconv_bn_train_fwd_stats(x, H, W, 2, tok0, 1e-5, 0.1, op[1:])
        ops.dwconv_bn(x, H, W, 1, tok0, (st[0]["taps"], st[0]["shift"])); ops.dwconv_bn(x, H, W, 2, tok0, (st[1]["taps"], st[1]["shift"]), (st[2]["taps"], st[2]["shift"]))

    def old_bwd():
        ob = [dict(wt=r, dy=d, g=p["g"], mean=t_["mean"], rstd=t_["rstd"], dg=a, db=b) for r, d, p, t_, a, b in zip(raws, dys, par, st, dg, db)]
        c0 = ops.dwconv_bn_train_bwd_stats(x, H, W, 1, tok0, ob[:1]); c12 = ops.dwconv_bn_train_bwd_stats(x, H, W, 2, tok0, ob[1:])
        for i, cf in enumerate(list(c0) + list(c12)):
            ops.dwconv_bn_train_dc_(x, raws[i], cf, dys[i], H, W, strides[i], tok0)
            G2, _ = ops.dwconv_bn_bwd_w(x, dys[i], H, W, strides[i], tok0)
            ops.tap_grad_accum(G2, dw[i])
        ops.dwconv_bn_bwd_dx([(d, r, s) for d, r, s in zip(dys, raws, strides)], Bn, C, H, W, tok0)
    t_old_f, t_old_b = timeit(old_fwd), timeit(old_bwd)
    new_f, new_b = t["stats"] + t["apply"], t["bwd_stats_y"] + t["dc_taps"] + t["dx"]
    print(f"Bn={Bn} C={C} H=W={H}  x={mb:.1f} MB | " + "  ".join(f"{k} {v:6.1f}us" for k, v in t.items()) +
          f" | fwd new {new_f:6.1f} old {t_old_f:6.1f}  bwd new {new_b:6.1f} old {t_old_b:6.1f} us")
