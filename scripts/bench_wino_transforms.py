#!/usr/bin/env python3
"""Input / output-gradient Winograd transforms alone (HBM-bound: read x, write 16 planes): time and bytes/s against the batch size
(the plane stride T*C*4 is a power of two at B = 16)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import lib as L

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for B in (int(v) for v in (sys.argv[1:] or (15, 16, 17, 32))):
    H = W = 64
    Cc = K = 256
    x = torch.randn(B, H + 2, W + 2, Cc, device=dev)
    T = B * (H // 2) * (W // 2)
    V = torch.zeros(16 * T * Cc + 4096, device=dev)
    d = L.WinoDesc()
    d.x, d.x_hp, d.x_wp, d.B, d.H, d.W, d.C, d.K = x.data_ptr(), H + 2, W + 2, B, H, W, Cc, K
    d.V, d.V_elems = V.data_ptr(), V.numel()
    ms = timeit(lambda: L.call("nirgan_wino_input", C.byref(d), st))
    by = x.numel() * 4 + 16 * T * Cc * 4
    dy = torch.randn(B, H, W, K, device=dev)
    y = L.WinoDyDesc()
    y.dy, y.dy_hp, y.dy_wp, y.dy_pad, y.B, y.H, y.W, y.K = dy.data_ptr(), H, W, 0, B, H, W, K
    y.Yt, y.Yt_elems = V.data_ptr(), V.numel()
    ms2 = timeit(lambda: L.call("nirgan_wino_dy", C.byref(y), st))
    by2 = dy.numel() * 4 + 16 * T * K * 4
    a, b2 = torch.empty(by // 8, device=dev), torch.empty(by // 8, device=dev)
    ms3 = timeit(lambda: b2.copy_(a))
    print(f"B={B:3d}  wino_input {ms * 1e3:6.1f} us ({by / ms / 1e9:6.2f} TB/s)   wino_dy {ms2 * 1e3:6.1f} us ({by2 / ms2 / 1e9:6.2f} TB/s)   "
          f"torch copy of the same bytes {ms3 * 1e3:6.1f} us ({by / ms3 / 1e9:6.2f} TB/s)")
