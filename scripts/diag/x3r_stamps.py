#!/usr/bin/env python3
"""Cycles per segment of the four-wave register-fed split tile (diagnostic build: scripts/diag/x3r_stamps.sh) on the trunk's plane GEMM
and on the trunk layer as a direct convolution."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import numpy as np
import torch
from nirgan_hip import lib as L
LIB = os.environ.get("X3R_LIB", os.path.join(ROOT, "scripts", "diag", "libnirgan_x3rstamp.so"))
print("library:", LIB)
L.set_backend(L._CLib(os.path.abspath(LIB)))
be = L.backend()
fn = be._dll.nirgan_x3r_stamps
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
NAMES = ["block 0", "block 1", "wait + barrier", "block 2", "block 3 (+B reads, cursor)", "between tiles", "epilogue set-up", "epilogue slices"]

def report(title):
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 4 * 16))()
    assert fn(buf, 256 * 4 * 16) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 16).astype(np.float64)
    a = a[a[:, 0, 15] > 0]                                  # (NG_X3R_CUS: a grid smaller than the chip)
    tiles, items = a[..., 15], a[..., 14]
    print(title, f"  tiles per wave {tiles.mean():.1f}, items per wave {items.mean():.1f}")
    tot = 0.0
    for k in range(5):
        v = (a[..., k] / np.maximum(tiles, 1)).mean(); tot += v
        print(f"   {NAMES[k]:28s} {v:8.0f} cycles per K-tile   (min wave {(a[..., k] / np.maximum(tiles, 1)).min():.0f}, max {(a[..., k] / np.maximum(tiles, 1)).max():.0f})")
    v = (a[..., 5] / np.maximum(tiles, 1)).mean(); tot += v
    print(f"   {NAMES[5]:28s} {v:8.0f} cycles per K-tile")
    print(f"   = {tot:.0f} cycles per K-tile (MFMA issue alone: {192 * 16})")
    clk = (a[..., 12] / np.maximum(a[..., 13], 1)).mean() * 100.0
    print(f"   whole kernel: {a[..., 12].mean():.0f} cycles per wave, clock {clk:.0f} MHz (s_memtime / s_memrealtime)")
    print(f"   item switch (lookup + prepare)  {(a[..., 8] / np.maximum(items, 1)).mean():8.0f} cycles per item;  behind the last K-tile {(a[..., 9] / np.maximum(items, 1)).mean():8.0f} cycles per item")
    for k in (6, 7):
        print(f"   {NAMES[k]:28s} {(a[..., k] / np.maximum(items, 1)).mean():8.0f} cycles per item")

def planes(B, H, W, Cc, K):
    T = B * ((H + 5) // 6) * ((W + 5) // 6)
    V = torch.randn(64 * T * Cc, generator=g).to(dev); U = (torch.randn(64 * K * Cc, generator=g) * 0.05).to(dev)
    zero = torch.zeros(64, device=dev); plane = 64 * K * Cc
    U3 = torch.zeros(3 * plane, dtype=torch.bfloat16, device=dev)
    L.call("nirgan_split3", U.data_ptr(), U3.data_ptr(), plane, plane, None)
    M = torch.zeros(64 * T * K, device=dev)
    d = L.Wino6Desc(); d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, K
    d.U3, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U3.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
    d.algo = L.W6_X3_R4
    for _ in range(20): L.call("nirgan_wino6_gemm", C.byref(d), None)
    report(f"plane GEMM 64 x [{T} x {Cc}] x [{K}]")

planes(16, 64, 64, 256, 256)
planes(16, 64, 64, 512, 256)
