#!/usr/bin/env python3
"""Microbenchmark (GPU box): the F(4x4,3x3) plane-GEMM launch alone, as a function of the contraction length C and the tile count T,
to separate the per-tile fixed cost (prologue, epilogue, launch tail) from the K loop."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import lib as L
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
zero = torch.zeros(64, device=dev)
def run(B, H, W, Cc, K, reps=20):
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    V = torch.randn(36 * T * Cc, device=dev); U = torch.randn(36 * K * Cc, device=dev) * 0.05; M = torch.zeros(36 * T * K, device=dev)
    d = L.Wino6Desc(); d.B, d.H, d.W, d.C, d.K = B, H, W, Cc, K
    d.U, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
    for _ in range(3): L.call("nirgan_wino6_gemm", C.byref(d), st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): L.call("nirgan_wino6_gemm", C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    fl = 2.0 * 36 * T * Cc * K
    blocks = 36 * -(-T // 128) * -(-K // 128)
    print(f"B={B} {H}x{W} T={T} C={Cc} K={K}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s  blocks={blocks} ({blocks/512:.2f} rounds)  bytes/launch {(36*T*(Cc+K)*4)/1e6:.0f} MB -> {(36*T*(Cc+K)*4)/ms/1e9:.2f} TB/s")
run(16, 64, 64, 256, 256, reps=60)        # clock ramp: the first case of a process measures 15 % low, discard it
print("--- measured cases")
for args in [(16, 64, 64, 256, 256), (16, 64, 64, 128, 256), (16, 64, 64, 512, 256), (16, 64, 64, 1024, 256), (16, 66, 66, 256, 256), (16, 64, 64, 256, 128),
             (14, 64, 64, 256, 256), (8, 64, 64, 256, 256), (32, 64, 64, 256, 256), (2, 64, 64, 256, 256), (16, 64, 64, 256, 256), (16, 66, 66, 256, 256),
             (16, 64, 64, 256, 256)]:
    run(*args)
