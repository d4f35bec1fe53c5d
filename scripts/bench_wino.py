#!/usr/bin/env python3
"""Winograd F(2x2,3x3) forward of the residual-block layer (bs 16, 64x64, 256 -> 256) vs the direct implicit-GEMM tile."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import lib as L

dev = "cuda:0"
B, H, W, Cc, K = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 64, 64, 256, 256)))
x = torch.randn(B, H + 2, W + 2, Cc, device=dev)
w = torch.randn(K, Cc, 3, 3, device=dev) * 0.05
T = B * (H // 2) * (W // 2)
U, V, y, zero = torch.zeros(16 * K * Cc, device=dev), torch.zeros(16 * T * Cc, device=dev), torch.zeros(B, H, W, K, device=dev), torch.zeros(64, device=dev)
d = L.WinoDesc()
d.x, d.x_hp, d.x_wp, d.B, d.H, d.W, d.C, d.K = x.data_ptr(), H + 2, W + 2, B, H, W, Cc, K
d.U, d.V, d.V_elems, d.y, d.zero_page = U.data_ptr(), V.data_ptr(), V.numel(), y.data_ptr(), zero.data_ptr()
st = torch.cuda.current_stream().cuda_stream
L.call("nirgan_wino_weights", w.data_ptr(), K, Cc, 0, U.data_ptr(), st)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


ms = timeit(lambda: L.call("nirgan_wino_conv3x3", C.byref(d), st))
msw = timeit(lambda: L.call("nirgan_wino_weights", w.data_ptr(), K, Cc, 0, U.data_ptr(), st))
fl = 2.0 * B * H * W * K * 9 * Cc
print(f"winograd conv3x3 (input transform + GEMM/output transform): {ms * 1e3:.1f} us = {fl / ms / 1e9:.1f} TFLOP/s direct-equivalent; weight transform {msw * 1e3:.1f} us")
