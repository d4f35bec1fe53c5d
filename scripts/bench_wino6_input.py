#!/usr/bin/env python3
"""Microbenchmark (GPU box): the F(6x6,3x3) input transform of a residual-block layer at bs 16 alone (x halo'd [16][66][66][256] -> V)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import lib as L
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
B, H, W, Cc = 16, 64, 64, 256
T = B * 11 * 11
x = torch.randn(B, H + 2, W + 2, Cc, device=dev)
V = torch.zeros(64 * (T * Cc + 4096), device=dev)
d = L.Wino6Desc(); d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, 256
d.x, d.x_hp, d.x_wp, d.V, d.V_elems = x.data_ptr(), H + 2, W + 2, V.data_ptr(), V.numel()
d.algo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def timeit(reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): L.call("nirgan_wino6_input", C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
timeit(50)
res = sorted(timeit(50) for _ in range(5))
by = x.numel() * 4 + 64 * T * Cc * 4
print(f"input transform algo {d.algo}: {res[2]*1e3:6.1f} us  {by/res[2]/1e9:5.2f} TB/s of algorithmic bytes ({by/1e6:.0f} MB)  pad {os.environ.get('NIRGAN_X_PAD','0')}")
