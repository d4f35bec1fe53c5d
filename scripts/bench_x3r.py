#!/usr/bin/env python3
"""The four-wave register-fed split tile (csrc/igemm_x3r.h) against the eight-wave tile it replaces (csrc/igemm_x3.h, descriptor algo
NIRGAN_CONV_X3_R4 / NIRGAN_W6_X3_R4) in ONE process: outputs compared BITWISE (same operand order, same accumulation order per output
element), interleaved timing rounds, median and min per arm.  Kill criterion of VERDICT r5 next #1: keep if >= 1.12 x on the stand-alone
plane GEMM (64 x [1936 x 256] x [256]) and on `conv M=65536 N=256 K=9x128`.

    python scripts/bench_x3r.py [rounds]            (MI355X)
"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, emit_conv

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = 20
dev = "cuda:0"
ctx = Ctx(dev, "fp32")
g = torch.Generator().manual_seed(0)
st = None


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def report(title, arms, flops, outs):
    names = list(arms)
    for n in names:
        arms[n]()
    torch.cuda.synchronize()
    same = all(torch.equal(outs[names[0]](), outs[n]()) for n in names[1:])
    times = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            times[n].append(timeit(arms[n]))
    base = statistics.median(times[names[0]])
    print(title, "   outputs bitwise equal" if same else "   OUTPUTS DIFFER", flush=True)
    for n in names:
        med, mn = statistics.median(times[n]), min(times[n])
        print(f"   {n:44s} median {med:7.1f} us  min {mn:7.1f}   {flops / (med * 1e-6) / 1e12:6.1f} TF/s fp32-equivalent  x{base / med:4.2f}", flush=True)
    return same


def plane_gemm(B, H, W, Cc, K, r=6):
    t = {6: (H + 5) // 6 * ((W + 5) // 6), 4: None}[r]
    T = B * t
    np_ = 64
    V = torch.randn(np_ * T * Cc, generator=g).to(dev)
    U = (torch.randn(np_ * K * Cc, generator=g) * 0.05).to(dev)
    zero = torch.zeros(64, device=dev)
    plane = np_ * K * Cc
    U3 = torch.zeros(3 * plane, dtype=torch.bfloat16, device=dev)
    L.call("nirgan_split3", U.data_ptr(), U3.data_ptr(), plane, plane, None)
    arms, outs = {}, {}
    for name, algo in (("eight waves, A through LDS (conv_x3_kernel)", 0), ("four waves, A from registers (conv_x3r_kernel)", L.W6_X3_R4)):
        M = torch.full((np_ * T * K,), float("nan"), device=dev)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = r, B, H, W, Cc, K
        d.U3, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U3.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
        d.algo = algo
        print("     ", name, "->", L.backend().nirgan_wino6_gemm_kernel_name(C.byref(d)).decode())
        arms[name] = (lambda d=d: L.call("nirgan_wino6_gemm", C.byref(d), st))
        outs[name] = (lambda M=M: M)
        ctx.keep += [M, d]
    ctx.keep += [V, U, U3, zero]
    return report(f"plane GEMM 64 x [{T} x {Cc}] x [{K}]", arms, 2.0 * np_ * T * Cc * K, outs)


def conv(B, H, cin, cout, k, s, with_bias):
    p = 1
    OH = G.conv_out(H, k, s, p)
    x = Halo(ctx, B, H, H, cin, p)
    x.interior().copy_(torch.randn(B, H, H, cin, generator=g).to(dev))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.02).to(dev)
    bias = torch.randn(cout, generator=g).to(dev) if with_bias else None
    spec = G.conv_fwd_pack(cout, cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    n = wp.numel()
    plane = (n + 7) // 8 * 8
    tw = torch.zeros(3 * plane, dtype=torch.bfloat16, device=dev)
    L.call("nirgan_split3", wp.data_ptr(), tw.data_ptr(), n, plane, None)
    ctx.keep += [wp, tw]
    arms, outs = {}, {}
    for name, algo in (("eight waves, A through LDS (conv_x3_kernel)", 0), ("four waves, A from registers (conv_x3r_kernel)", L.CONV_X3_R4)):
        y = Halo(ctx, B, OH, OH, cout, 0)
        y.t.fill_(float("nan"))
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, bias, y, N=cout, OH=OH, OW=OH, in_stride=s, allow_split=False)
        d.precision, d.w_x3, d.w_x3_plane, d.algo = 3, tw.data_ptr(), plane, algo
        print("     ", name, "->", L.backend().nirgan_conv_kernel_name(C.byref(d)).decode())
        arms[name] = (lambda d=d: L.call("nirgan_conv_igemm", C.byref(d), st))
        outs[name] = (lambda y=y: y.t)
        ctx.keep += [d]
    return report(f"conv B={B} {H}x{H} {cin}->{cout} k{k} s{s}{' +bias' if with_bias else ''}", arms, 2.0 * B * OH * OH * cout * k * k * cin, outs)


ok = True
only = os.environ.get("X3R_CASES", "")
if not only or "small" in only:
    ok &= plane_gemm(2, 16, 16, 128, 128)                 # T = 18: one partly filled tile per plane
    ok &= conv(2, 32, 64, 128, 3, 2, True)                # one partly filled M tile, bias
    ok &= conv(3, 31, 128, 256, 3, 1, False)              # ragged last tile, two column tiles
    ok &= conv(1, 16, 32, 128, 3, 1, True)                # one K-tile per tap
if not only or "big" in only:
    ok &= plane_gemm(16, 64, 64, 256, 256)                # the benchmark's trunk layer
    ok &= conv(16, 128, 128, 256, 3, 2, False)            # conv M=65536 N=256 K=9x128 s2
    ok &= conv(16, 256, 64, 128, 3, 2, False)             # conv M=262144 N=128 K=9x64 s2
    ok &= conv(16, 64, 256, 256, 3, 1, False)             # the trunk layer as a direct convolution
print("ALL BITWISE EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
