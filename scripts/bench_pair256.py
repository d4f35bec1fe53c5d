#!/usr/bin/env python3
"""The bf16 residual-block backward launch (data gradient over the padded extent + weight gradient, nirgan_conv_wgrad_pair) on the persistent
256-wide tiles against the 128-row tiles, interleaved rounds in one process; the weight gradient alone; reduce_rows of each split count."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv, emit_wgrad
from nirgan_hip.options import OPT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reps, Cc, k = 20, 256, 3
ctx = Ctx("cuda:0", "bf16")
g = torch.Generator().manual_seed(0)
x = Halo(ctx, B, H, H, Cc, 1, twin=True)
x.t.copy_(torch.randn(x.t.shape, generator=g).to("cuda:0")); x.t16.copy_(x.t.to(torch.bfloat16))
dy = Halo(ctx, B, H, H, Cc, 2, twin=True)
dy.interior().copy_(torch.randn(B, H, H, Cc, generator=g).to("cuda:0")); dy.t16.copy_(dy.t.to(torch.bfloat16))
w = (torch.randn(Cc, Cc, k, k, generator=g) * 0.02).to("cuda:0")
spec = G.conv_dgrad_pack(Cc, Cc, k, [(a, b) for a in range(k) for b in range(k)])
wd = torch.zeros(spec.N, spec.K, dtype=torch.bfloat16, device="cuda:0")
L.call("nirgan_pack_rows_bf16", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wd.data_ptr(), spec.N, spec.K, None)
torch.cuda.synchronize()


def build(tile256, pair, g16=True):
    OPT.tile256 = tile256
    try:
        gw = ctx.zeros(Cc, Cc, k, k)
        plan = Plan(ctx)
        cd = None
        if pair:
            gx = Halo(ctx, B, H, H, Cc, 1, bf16=g16)
            cd = emit_conv(None, ctx, dy, G.conv_dgrad_s1_taps(k, Cc), wd, None, gx, N=Cc, OH=gx.hp, OW=gx.wp)
        d = emit_wgrad(plan, ctx, dy, x, G.conv_fwd_taps(k, Cc), G.conv_fwd_pack(Cc, Cc, k), gw, N=Cc, OH=H, OW=H, p_oh=2, p_ow=2, pair_with=cd)
    finally:
        OPT.reset()
    main, red = Plan(ctx), Plan(ctx)
    main.ops, red.ops = plan.ops[:1], plan.ops[1:]
    return main, red, d


def once(plan):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        plan.run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


fl_w = 2.0 * B * H * H * Cc * 9 * Cc
fl_d = 2.0 * B * (H + 2) ** 2 * Cc * 9 * Cc
arms = {}
for name, t256, pair in (("pair 256", True, True), ("pair 128", False, True), ("wgrad 256", True, False), ("wgrad 128", False, False)):
    m, r, d = build(t256, pair)
    arms[name] = (m, fl_w + (fl_d if pair else 0), f"split {d.nsplit} x {d.rows_per_split}")
    arms[name + " reduce"] = (r, 0.0, "")
for p, _, _ in arms.values():
    once(p)
times = {n: [] for n in arms}
for r in range(rounds):
    for n, (p, _, _) in arms.items():
        times[n].append(once(p))
for n, (p, fl, note) in arms.items():
    med, mn = statistics.median(times[n]), min(times[n])
    tf = f"{fl / (med * 1e-6) / 1e12:7.1f} TF/s = {fl / (med * 1e-6) / 2.5e15:.3f} of 2.5 PFLOP/s" if fl else ""
    print(f"{n:18s} median {med:7.1f} us  min {mn:7.1f} us   {tf}  {note}", flush=True)
