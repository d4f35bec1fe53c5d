#!/usr/bin/env python3
"""Part timings of the bf16 256 x 256 x 64 tile (csrc/igemm_tile256.h) on the residual-block shape: the same launch with 1x1 (4 K-tiles)
and 3x3 (36 K-tiles) taps separates the per-K-tile cost from prologue + epilogue; fp32 against bf16 output separates the store;
NIRGAN_CONV_TILE128 is the A/B partner in the same process (interleaved rounds, median and min)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import ctypes as C
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, emit_conv

PREC = sys.argv[4] if len(sys.argv) > 4 else "bf16"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reps = 20
Cc = 256
ctx = Ctx("cuda:0", PREC)
g = torch.Generator().manual_seed(0)


def packed16(w, spec):
    if PREC == "fp32":
        buf = torch.zeros(spec.N, spec.K, device="cuda:0")
        L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), buf.data_ptr(), spec.N, spec.K, None)
        torch.cuda.synchronize()
        return buf
    buf = torch.zeros(spec.N, spec.K, dtype=torch.bfloat16, device="cuda:0")
    L.call("nirgan_pack_rows_bf16", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), buf.data_ptr(), spec.N, spec.K, None)
    torch.cuda.synchronize()
    return buf


def problem(k, out16, algo, cin=Cc, cout=Cc):
    p = (k - 1) // 2
    x = Halo(ctx, B, H, H, cin, p, twin=(PREC == "bf16"))
    x.t.copy_(torch.randn(x.t.shape, generator=g).to("cuda:0"))
    if PREC == "bf16":
        x.t16.copy_(x.t.to(torch.bfloat16))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.02).to("cuda:0")
    wp = packed16(w, G.conv_fwd_pack(cout, cin, k))
    ctx.keep.append(wp)
    y = Halo(ctx, B, H, H, cout, 0, bf16=out16 and PREC == "bf16")
    d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, None, y, N=cout, OH=H, OW=H, allow_split=False)
    d.algo = L.CONV_TILE256 if (algo == 0 and PREC == "fp32") else algo
    return d, 2.0 * B * H * H * cout * k * k * cin


def once(d):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        L.call("nirgan_conv_igemm", C.byref(d), None)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


arms = {}
for name, k, out16, algo in (("256 3x3 fp32-out", 3, False, 0), ("256 3x3 bf16-out", 3, True, 0), ("256 1x1 fp32-out", 1, False, 0), ("256 1x1 bf16-out", 1, True, 0),
                             ("128 3x3 fp32-out", 3, False, L.CONV_TILE128), ("128 3x3 bf16-out", 3, True, L.CONV_TILE128)):
    arms[name] = problem(k, out16, algo)
for d, _ in arms.values():
    once(d)
times = {n: [] for n in arms}
for r in range(rounds):
    for n, (d, _) in arms.items():
        times[n].append(once(d))
for n, (d, fl) in arms.items():
    med, mn = statistics.median(times[n]), min(times[n])
    print(f"{n:20s} median {med:7.1f} us  min {mn:7.1f} us   {fl / (med * 1e-6) / 1e12:7.1f} TF/s = {fl / (med * 1e-6) / 2.5e15:.3f} of 2.5 PFLOP/s", flush=True)
t3, t1 = statistics.median(times["256 3x3 bf16-out"]), statistics.median(times["256 1x1 bf16-out"])
PEAK = 2.5e15 if PREC == "bf16" else 157.3e12
for n, (d, fl) in arms.items():
    print(f"   {n:20s} {fl / (statistics.median(times[n]) * 1e-6) / PEAK:.3f} of the {PREC} peak")
print(f"per 64 k of a 256 x 256 tile: {(t3 - t1) / 32 * 1e3:.0f} ns  -> loop-only rate {2.0 * 256 * 256 * 64 * 256 / ((t3 - t1) / 32 * 1e-6) / 1e12:.0f} TF/s; "
      f"fixed part (prologue + epilogue + launch): {t1 - 4 * (t3 - t1) / 32:.1f} us")
