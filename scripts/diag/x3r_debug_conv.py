#!/usr/bin/env python3
"""conv_x3r_kernel against the eight-wave tile on convolutions of odd extents (bitwise)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, emit_conv
dev = "cuda:0"
ctx = Ctx(dev, "fp32")
g = torch.Generator().manual_seed(0)
for (B, H, cin, cout, k, s, bias) in [(1, 266, 128, 256, 3, 2, False), (1, 133, 256, 256, 3, 1, False), (1, 532, 64, 128, 3, 2, True), (2, 133, 256, 128, 3, 1, False), (1, 266, 128, 128, 3, 1, True)]:
    OH = G.conv_out(H, k, s, 1)
    x = Halo(ctx, B, H, H, cin, 1)
    x.interior().copy_(torch.randn(B, H, H, cin, generator=g).to(dev))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.02).to(dev)
    bv = torch.randn(cout, generator=g).to(dev) if bias else None
    spec = G.conv_fwd_pack(cout, cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    n = wp.numel(); plane = (n + 7) // 8 * 8
    tw = torch.zeros(3 * plane, dtype=torch.bfloat16, device=dev)
    L.call("nirgan_split3", wp.data_ptr(), tw.data_ptr(), n, plane, None)
    outs = []
    for algo in (0, L.CONV_X3_R4):
        y = Halo(ctx, B, OH, OH, cout, 0); y.t.fill_(float("nan"))
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, bv, y, N=cout, OH=OH, OW=OH, in_stride=s, allow_split=False)
        d.precision, d.w_x3, d.w_x3_plane, d.algo = 3, tw.data_ptr(), plane, algo
        name = L.backend().nirgan_conv_kernel_name(C.byref(d)).decode()
        L.call("nirgan_conv_igemm", C.byref(d), None); torch.cuda.synchronize()
        outs.append((name, y.t.clone()))
    a, b = outs[0][1], outs[1][1]
    bad = (a != b) | torch.isnan(b)
    print(f"conv B={B} {H}x{H} {cin}->{cout} k{k} s{s} OW={OH}: {outs[0][0]} vs {outs[1][0]}: mismatching {int(bad.sum())} of {bad.numel()}", flush=True)
    if bad.any():
        idx = bad.nonzero()
        print("   rows (b, oh) first:", idx[:3].tolist(), " ow histogram (first 20 cols):", torch.bincount(idx[:, 2], minlength=OH)[:20].tolist(), " ch//16:", torch.bincount(idx[:, 3] // 16).tolist())
