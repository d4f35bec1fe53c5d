#!/usr/bin/env python3
"""Microbenchmark (GPU box): the F(6x6,3x3) plane-GEMM launch of a residual-block layer at bs 16 (64 planes x [T = 1936 x 256] x [256]),
back to back; prints us per launch and executed TFLOP/s.  Used for A/B of kernel variants (descriptor field algo, experiments)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import lib as L
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
zero = torch.zeros(64, device=dev)
B, H, W, Cc, K = 16, 64, 64, 256, 256
T = B * 11 * 11
V = torch.randn(64 * T * Cc, device=dev); U = torch.randn(64 * K * Cc, device=dev) * 0.05; M = torch.zeros(64 * T * K, device=dev)
d = L.Wino6Desc(); d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, K
d.U, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
d.algo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def timeit(reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): L.call("nirgan_wino6_gemm", C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
timeit(100)
res = [timeit(50) for _ in range(5)]
ms = sorted(res)[2]
print(f"{L.backend().nirgan_wino6_gemm_kernel_name(C.byref(d)).decode()}: {ms*1e3:7.1f} us  {2.0*64*T*Cc*K/ms/1e9:6.1f} TF/s  (runs {[round(r*1e3,1) for r in res]})")
