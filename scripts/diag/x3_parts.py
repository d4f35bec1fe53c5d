#!/usr/bin/env python3
"""Part timings of the trunk's plane GEMM on the split tile (64 planes x [T = 1936 x 256] x [256]): the shipped launch, the same without
the tiles' global stores, and without the epilogue -- from the diagnostic library of scripts/diag/x3_parts.sh.  Results are wrong with a
switch set; only the times mean something."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import ctypes as C
import torch
from nirgan_hip import lib as L
L.set_backend(L._CLib(os.path.join(ROOT, "scripts", "diag", "libnirgan_x3diag.so")))
DEV = "cuda:0"
for (v, B, H, W, Cc, K) in ((6, 16, 64, 64, 256, 256), (4, 32, 31, 31, 256, 512)):
    r, mo = (3, 6) if v == 6 else (v, 4)
    NP = (mo + r - 1) ** 2
    be = L.backend()
    T = int(be.nirgan_wino6_tiles_r(B, H, W, v))
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(K, Cc, r, r, generator=g) * 0.05).to(DEV)
    U = torch.zeros(NP * K * Cc, device=DEV)
    U3 = torch.zeros(3 * NP * K * Cc, dtype=torch.bfloat16, device=DEV)
    L.call("nirgan_wino6_weights_x3", w.data_ptr(), K, Cc, v, 0, U.data_ptr(), U3.data_ptr(), None)
    V = torch.randn(NP * T * Cc, generator=g).to(DEV)
    M = torch.zeros(NP * T * K, device=DEV)
    zero = torch.zeros(64, device=DEV)
    descs = {}
    for name, algo in (("shipped", 0), ("no global stores", 0x100), ("no epilogue", 0x200), ("no epilogue, no conversion", 0x200 | 0x400),
                       ("no epilogue, no conversion, no LDS stores", 0x200 | 0x800), ("no epilogue, no fetch", 0x200 | 0x1000),
                       ("no epilogue, no fetch, no conversion / stores", 0x200 | 0x1000 | 0x800)):
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = v, B, H, W, Cc, K
        d.U, d.V, d.V_elems, d.M, d.M_elems, d.zero_page, d.U3, d.algo = U.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr(), U3.data_ptr(), algo
        descs[name] = d

    def once(d, reps=20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            L.call("nirgan_wino6_gemm", C.byref(d), None)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3
    for d in descs.values():
        once(d, 3)
    times = {n: [] for n in descs}
    for _ in range(5):
        for n, d in descs.items():
            times[n].append(once(d))
    fl = 2.0 * NP * T * Cc * K
    print(f"plane GEMM variant {v}: {NP} x [T={T} x C={Cc}] x [K={K}]")
    for n in descs:
        med = statistics.median(times[n])
        print(f"   {n:48s} {med:7.1f} us   {fl / med / 1e6:6.1f} TF/s fp32-equivalent", flush=True)
