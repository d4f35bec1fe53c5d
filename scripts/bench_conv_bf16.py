#!/usr/bin/env python3
"""Micro-benchmark of the bf16 operand mode's residual-block convolution (both operands stored as bf16): 3x3 256 -> 256 at the benchmark
batch, forward and data gradient (full correlation over the padded extent), event-timed TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
Cc = 256
ctx = Ctx("cuda:0", "bf16")
g = torch.Generator().manual_seed(0)


def packed(w, spec):
    wp = ctx.zeros(spec.N, spec.K)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    torch.cuda.synchronize()
    return wp.to(torch.bfloat16).contiguous()


def twin(h):
    h.t16.copy_(h.t.to(torch.bfloat16))


x = Halo(ctx, B, H, H, Cc, 1, twin=True)
x.t.copy_(torch.randn(x.t.shape, generator=g).to("cuda:0"))
twin(x)
w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.02).to("cuda:0")
wf = packed(w, G.conv_fwd_pack(Cc, Cc, 3))
y = Halo(ctx, B, H, H, Cc, 0)
fwd = Plan(ctx)
emit_conv(fwd, ctx, x, G.conv_fwd_taps(3, Cc), wf, None, y, N=Cc, OH=H, OW=H, allow_split=False)
dy = Halo(ctx, B, H, H, Cc, 2, twin=True)
dy.interior().copy_(torch.randn(B, H, H, Cc, generator=g).to("cuda:0"))
twin(dy)
gx = Halo(ctx, B, H, H, Cc, 1)
wd = packed(w, G.conv_dgrad_pack(Cc, Cc, 3, [(a, b) for a in range(3) for b in range(3)]))
dg = Plan(ctx)
emit_conv(dg, ctx, dy, G.conv_dgrad_s1_taps(3, Cc), wd, None, gx, N=Cc, OH=gx.hp, OW=gx.wp, allow_split=False)


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
    print(f"{name:28s} {ms * 1e3:9.1f} us  {flops / (ms * 1e-3) / 1e12:7.1f} TF/s = {flops / (ms * 1e-3) / 2.5e15:.3f} of 2.5 PFLOP/s", flush=True)


M = B * H * H
timeit(fwd, 2.0 * M * Cc * 9 * Cc, "conv fwd 3x3 256 (bf16)")
timeit(dg, 2.0 * B * (H + 2) ** 2 * Cc * 9 * Cc, "dgrad (full corr., bf16)")
