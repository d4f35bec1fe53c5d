#!/usr/bin/env python3
"""Kill criterion of the fp32-equivalent three-term split tile (csrc/igemm_x3.h, descriptor precision 3): the same convolution in one
process as exact fp32 (the 128-row tile, precision 0), as the two-term bf16x3 mode (precision 2) and as the six-product split (precision
3) -- interleaved rounds, median and min per arm -- and each arm's error against float64 on a random sample of output pixels.
Keep only if precision 3 is >= 1.4x faster than precision 0 AND its max error is <= 2x the fp32 tile's.

    python scripts/bench_x3.py [rounds]            (MI355X)
"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import ctypes as C
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv, emit_wgrad
from nirgan_hip.options import OPT

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = 10
dev = "cuda:0"
ctx = Ctx(dev, "fp32")
g = torch.Generator().manual_seed(0)


def split3(wp):
    """three bf16 planes of a packed fp32 weight buffer (nirgan_split3)"""
    n = wp.numel()
    plane = (n + 7) // 8 * 8
    tw = torch.zeros(3 * plane, dtype=torch.bfloat16, device=dev)
    L.call("nirgan_split3", wp.data_ptr(), tw.data_ptr(), n, plane, None)
    torch.cuda.synchronize()
    ctx.keep.append(tw)
    return tw, plane


def conv_problem(B, H, cin, cout, k, s):
    p = 1
    OH = G.conv_out(H, k, s, p)
    x = Halo(ctx, B, H, H, cin, p)
    x.interior().copy_(torch.randn(B, H, H, cin, generator=g).to(dev))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.02).to(dev)
    spec = G.conv_fwd_pack(cout, cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    ctx.keep.append(wp)
    tw, plane = split3(wp)
    arms = {}
    for name, prec, algo in (("fp32 (precision 0)", 0, 0), ("bf16x3 two terms (precision 2)", 2, 0), ("three terms, six products (precision 3)", 3, 0),
                             ("   ... on 256 x 64 tiles", 3, L.CONV_X3_BN64)):
        y = Halo(ctx, B, OH, OH, cout, 0)
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, None, y, N=cout, OH=OH, OW=OH, in_stride=s, allow_split=False)
        d.precision = prec
        if prec == 3:
            d.w_x3, d.w_x3_plane, d.algo = tw.data_ptr(), plane, algo
        arms[name] = (d, y)
    flops = 2.0 * B * OH * OH * cout * k * k * cin
    return arms, flops, (x, w, k, s, p, OH)


def check(arms, ref_args, nsample=8192):
    """every arm against float64 on `nsample` random output pixels (all channels)"""
    x, w, k, s, p, OH = ref_args
    B, cin, cout = x.B, x.C, w.shape[0]
    idx = torch.randint(0, B * OH * OH, (nsample,), generator=g).to(dev)
    b, r = idx // (OH * OH), idx % (OH * OH)
    oh, ow = r // OH, r % OH
    xp = x.t.double()                                        # [B][H+2][W+2][cin], zero halo
    patches = torch.stack([xp[b, oh * s + kh, ow * s + kw, :] for kh in range(k) for kw in range(k)], 1)       # [n][k*k][cin]
    wk = w.double().permute(0, 2, 3, 1).reshape(cout, k * k, cin)
    ref = torch.einsum("ntc,otc->no", patches, wk)
    scale = ref.abs().max().item()
    out = {}
    for name, (d, y) in arms.items():
        y.t.zero_()
        L.call("nirgan_conv_igemm", C.byref(d), None)
        torch.cuda.synchronize()
        got = y.t[b, oh, ow, :].double()
        e = (got - ref).abs()
        out[name] = (e.max().item() / scale, e.pow(2).mean().sqrt().item() / scale)
    return out


def once(d):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        L.call("nirgan_conv_igemm", C.byref(d), None)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


CASES = [("conv M=65536 N=256 K=9x128 s2 (generator, 128 -> 256)", (16, 128, 128, 256, 3, 2)),
         ("conv M=262144 N=128 K=9x64 s2 (generator, 64 -> 128)", (16, 256, 64, 128, 3, 2)),
         ("conv M=32768 N=256 K=16x128 s2 (PatchGAN, 2B = 32)", (32, 64, 128, 256, 4, 2)),
         ("conv M=131072 N=128 K=16x64 s2 (PatchGAN, 2B = 32)", (32, 128, 64, 128, 4, 2)),
         ("conv M=65536 N=256 K=9x256 s1 (residual trunk, direct)", (16, 64, 256, 256, 3, 1))]


def wgrad_problem(B, H, cin, cout, k, s):
    """dW of Conv2d(cin, cout, k, stride s, padding 1): P = dY, Q = the halo'd input"""
    OH = G.conv_out(H, k, s, 1)
    x = Halo(ctx, B, H, H, cin, 1)
    x.interior().copy_(torch.randn(B, H, H, cin, generator=g).to(dev))
    dy = Halo(ctx, B, OH, OH, cout, 0)
    dy.t.copy_(torch.randn(B, OH, OH, cout, generator=g).to(dev))
    spec = G.conv_fwd_pack(cout, cin, k)
    arms = {}
    for name, prec in (("fp32 (precision 0)", 0), ("bf16x3 two terms (precision 2)", 2), ("three terms, six products (precision 3)", 3)):
        OPT.split3 = prec == 3
        gw = ctx.zeros(cout, cin, k, k)
        ctx.keep.append(gw)
        plan = Plan(ctx)
        d = emit_wgrad(plan, ctx, dy, x, G.conv_fwd_taps(k, cin), spec, gw, N=cout, OH=OH, OW=OH, p_oh=0, p_ow=0, q_stride=s, q_oh=0, q_ow=0)
        if prec == 2:
            d.precision = 2
        assert d.precision == prec
        arms[name] = (d, gw, plan)
    OPT.split3 = True
    flops = 2.0 * B * OH * OH * cout * k * k * cin
    return arms, flops, (x, dy, k, s, OH)


def wgrad_check(arms, ref_args):
    x, dy, k, s, OH = ref_args
    xd, dyd = x.t.double(), dy.t.double().reshape(-1, dy.C)
    ref = torch.stack([torch.stack([dyd.T @ xd[:, kh:kh + (OH - 1) * s + 1:s, kw:kw + (OH - 1) * s + 1:s, :].reshape(-1, x.C) for kw in range(k)], -1)
                       for kh in range(k)], -2)          # [cout][cin][kh][kw]
    scale = ref.abs().max().item()
    out = {}
    for name, (d, gw, plan) in arms.items():
        gw.zero_()
        plan.run()
        torch.cuda.synchronize()
        e = (gw.double() - ref).abs()
        out[name] = (e.max().item() / scale, e.pow(2).mean().sqrt().item() / scale)
    return out


def once_w(d):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        L.call("nirgan_wgrad_igemm", C.byref(d), None)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


WCASES = [("wgrad M=262144 N=128 K=576 (generator, 64 -> 128 s2)", (16, 256, 64, 128, 3, 2)),
          ("wgrad M=65536 N=256 K=1152 (generator, 128 -> 256 s2)", (16, 128, 128, 256, 3, 2)),
          ("wgrad M=32768 N=256 K=2048 (PatchGAN, 128 -> 256 4x4 s2)", (32, 64, 128, 256, 4, 2)),
          ("wgrad M=131072 N=128 K=1024 (PatchGAN, 64 -> 128 4x4 s2)", (32, 128, 64, 128, 4, 2))]
only = os.environ.get("X3_CASES")
for title, args in WCASES:
    if only and not any(t in title for t in only.split(",")):
        continue
    arms, flops, ref_args = wgrad_problem(*args)
    names = {n: (L.backend().nirgan_wgrad_kernel_name(C.byref(d)) or b"?").decode() + f" split={d.nsplit}" for n, (d, _, _) in arms.items()}
    errs = wgrad_check(arms, ref_args)
    for d, _, _ in arms.values():
        once_w(d)
    times = {n: [] for n in arms}
    for r in range(rounds):
        for n, (d, _, _) in arms.items():
            times[n].append(once_w(d))
    print(title, flush=True)
    base = statistics.median(times["fp32 (precision 0)"])
    for n in arms:
        med, mn = statistics.median(times[n]), min(times[n])
        print(f"   {n:42s} {names[n]:34s} median {med:7.1f} us  min {mn:7.1f}   {flops / (med * 1e-6) / 1e12:6.1f} TF/s  x{base / med:4.2f}"
              f"   err vs fp64: max {errs[n][0]:.2e} rms {errs[n][1]:.2e}  (x{errs[n][0] / errs['fp32 (precision 0)'][0]:.2f} / x{errs[n][1] / errs['fp32 (precision 0)'][1]:.2f} of fp32)", flush=True)
    ctx.keep.clear()
    torch.cuda.empty_cache()
for title, args in CASES:
    if only and not any(t in title for t in only.split(",")):
        continue
    arms, flops, ref_args = conv_problem(*args)
    names = {n: (L.backend().nirgan_conv_kernel_name(C.byref(d)) or b"?").decode() for n, (d, _) in arms.items()}
    errs = check(arms, ref_args)
    for d, _ in arms.values():
        once(d)
    times = {n: [] for n in arms}
    for r in range(rounds):
        for n, (d, _) in arms.items():
            times[n].append(once(d))
    print(title, flush=True)
    base = statistics.median(times["fp32 (precision 0)"])
    for n in arms:
        med, mn = statistics.median(times[n]), min(times[n])
        print(f"   {n:42s} {names[n]:26s} median {med:7.1f} us  min {mn:7.1f}   {flops / (med * 1e-6) / 1e12:6.1f} TF/s  x{base / med:4.2f}"
              f"   err vs fp64: max {errs[n][0]:.2e} rms {errs[n][1]:.2e}  (x{errs[n][0] / errs['fp32 (precision 0)'][0]:.2f} / x{errs[n][1] / errs['fp32 (precision 0)'][1]:.2f} of fp32)", flush=True)
    ctx.keep.clear()
    torch.cuda.empty_cache()
