#!/usr/bin/env python3
"""The instance-norm launches of a residual-trunk layer (16 x 64 x 64 x 256) and of the first / last generator layer (16 x 256 x 256 x 64) alone,
back to back: nirgan_instnorm_fwd (statistics from producer partials are NOT used here: finalize of the kernel's own statistics pass is
excluded by timing the apply-only form where possible) and nirgan_instnorm_bwd, in the fp32 layout and in the bf16 operand mode's storage
(y and the gradient stored as bf16, twin-only outputs); us per launch and the algorithmic bytes over that time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import ctypes as C
import torch
from nirgan_hip import lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_in_bwd, emit_in_fwd

dev = "cuda:0"
reps = 30


def timeit(plans):
    """plans: SETS copies of the same launches on disjoint buffers, run round-robin -- with SETS > 1 the buffers of one copy have left the
    256 MB last-level cache by the time it runs again, as they have inside a training step"""
    for _ in range(2):
        for plan in plans:
            plan.run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for plan in plans:
            plan.run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * len(plans)) * 1e3


SETS = int(os.environ.get("SETS", "1"))


for (B, H, W, Cc, pad) in ((16, 64, 64, 256, 1), (16, 256, 256, 64, 3), (16, 128, 128, 128, 1)):
    for mode in ("fp32", "bf16"):
        b16 = mode == "bf16"
        pfs, pbs, keep = [], [], []
        for _set in range(SETS):
            ctx = Ctx(dev, mode)
            y = Halo(ctx, B, H, W, Cc, 0, bf16=b16)
            y.t.copy_(torch.randn(y.t.shape, device=dev).to(y.t.dtype))
            out = Halo(ctx, B, H, W, Cc, pad, twin=True)
            stats = (ctx.zeros(B, Cc), ctx.zeros(B, Cc))
            stats[1].fill_(1.0)
            ws = ctx.zeros(int(L.backend().nirgan_instnorm_ws_elems(B, H, W, Cc)) + B * 2 * Cc)
            pf = Plan(ctx)
            d = emit_in_fwd(pf, ctx, y, out, norm=True, act=L.ACT_RELU, border=L.BORDER_REFLECT, stats=stats, ws=ws)
            if b16:
                for n, a in pf.ops:
                    if n == "nirgan_instnorm_fwd":
                        a[0]._obj.out = None           # twin only, as drop_dead_fp32_stores leaves a trunk layer
            g = Halo(ctx, B, H, W, Cc, pad, bf16=b16)
            g.t.copy_(torch.randn(g.t.shape, device=dev).to(g.t.dtype))
            dy = Halo(ctx, B, H, W, Cc, 2, twin=True)
            pb = Plan(ctx)
            emit_in_bwd(pb, ctx, g=g, g_fold=True, act=L.ACT_RELU, y=y, stats=stats, norm=True, dy=dy, ws=ws, shape=(B, H, W, Cc))
            if b16:
                for n, a in pb.ops:
                    if n == "nirgan_instnorm_bwd":
                        a[0]._obj.dy = None
            pfs.append(pf); pbs.append(pb); keep.append((ctx, y, out, stats, ws, g, dy))
        e_in = 2 if b16 else 4
        px, pxh = B * H * W * Cc, B * (H + 2 * pad) * (W + 2 * pad) * Cc
        fwd_bytes = px * e_in * 2 + pxh * (2 if b16 else 6)                         # statistics pass + apply pass read y; out (+ twin) written
        bwd_bytes = 2 * (pxh * e_in + px * e_in) + B * (H + 4) * (W + 4) * Cc * 0 + px * (2 if b16 else 6)
        tf, tb = timeit(pfs), timeit(pbs)
        print(f"{B}x{H}x{W}x{Cc} {mode}: forward (stats + finalize + apply) {tf:6.1f} us = {fwd_bytes / tf / 1e6:5.2f} TB/s algorithmic;  "
              f"backward (pass 1 + finalize + pass 2) {tb:6.1f} us = {bwd_bytes / tb / 1e6:5.2f} TB/s", flush=True)
