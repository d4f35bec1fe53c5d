#!/usr/bin/env python3
"""Micro-benchmark of single launches (GPU box): the residual-block 3x3 256->256 convolution forward, its fused
data-gradient + weight-gradient pair, at the benchmark batch.  Prints event-timed TFLOP/s; use under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv, emit_wgrad

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
Cc = 256
ctx = Ctx("cuda:0")
x = Halo(ctx, B, H, H, Cc, 1)
x.t.normal_()
w = torch.randn(Cc, Cc, 3, 3, device="cuda:0") * 0.02
spec = G.conv_fwd_pack(Cc, Cc, 3)
wp = ctx.zeros(spec.N, spec.K)
L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
y = Halo(ctx, B, H, H, Cc, 0)
fwd = Plan(ctx)
emit_conv(fwd, ctx, x, G.conv_fwd_taps(3, Cc), wp, None, y, N=Cc, OH=H, OW=H)
dy = Halo(ctx, B, H, H, Cc, 2)
dy.interior().normal_()
gx = Halo(ctx, B, H, H, Cc, 1)
gw = ctx.zeros(Cc, Cc, 3, 3)
dspec = G.conv_dgrad_pack(Cc, Cc, 3, [(a, b) for a in range(3) for b in range(3)])
wd = ctx.zeros(dspec.N, dspec.K)
L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), dspec.row_stride, ctx.i32(dspec.index_map).data_ptr(), wd.data_ptr(), dspec.N, dspec.K, None)
pair = Plan(ctx)
cd = emit_conv(None, ctx, dy, G.conv_dgrad_s1_taps(3, Cc), wd, None, gx, N=Cc, OH=gx.hp, OW=gx.wp)
emit_wgrad(pair, ctx, dy, x, G.conv_fwd_taps(3, Cc), spec, gw, N=Cc, OH=H, OW=H, p_oh=2, p_ow=2, pair_with=cd)
dg = Plan(ctx)
emit_conv(dg, ctx, dy, G.conv_dgrad_s1_taps(3, Cc), wd, None, gx, N=Cc, OH=gx.hp, OW=gx.wp)
wg = Plan(ctx)
emit_wgrad(wg, ctx, dy, x, G.conv_fwd_taps(3, Cc), spec, gw, N=Cc, OH=H, OW=H, p_oh=2, p_ow=2)


def timeit(plan, flops, name):
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        plan.run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{name:28s} {ms * 1e3:9.1f} us  {flops / (ms * 1e-3) / 1e12:7.1f} TF/s")


M = B * H * H
f_fwd = 2.0 * M * Cc * 9 * Cc
f_dg = 2.0 * B * (H + 2) ** 2 * Cc * 9 * Cc
which = sys.argv[4] if len(sys.argv) > 4 else "all"
if which in ("all", "fwd"):
    timeit(fwd, f_fwd, "conv fwd 3x3 256")
if which in ("all", "dgrad"):
    timeit(dg, f_dg, "dgrad (full corr.)")
if which in ("all", "wgrad"):
    timeit(wg, f_fwd, "wgrad + reduce")
if which in ("all", "pair"):
    timeit(pair, f_fwd + f_dg, "pair + reduce")
