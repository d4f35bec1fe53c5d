#!/usr/bin/env python3
"""Where does the four-wave register-fed tile differ from the eight-wave tile?  Plane GEMMs of a few shapes, mismatch structure."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import lib as L
dev = "cuda:0"
g = torch.Generator().manual_seed(0)

def run(B, H, W, Cc, K):
    T = B * ((H + 5) // 6) * ((W + 5) // 6)
    V = torch.randn(64 * T * Cc, generator=g).to(dev)
    U = (torch.randn(64 * K * Cc, generator=g) * 0.05).to(dev)
    zero = torch.zeros(64, device=dev)
    plane = 64 * K * Cc
    U3 = torch.zeros(3 * plane, dtype=torch.bfloat16, device=dev)
    L.call("nirgan_split3", U.data_ptr(), U3.data_ptr(), plane, plane, None)
    outs = []
    for algo in (0, L.W6_X3_R4):
        M = torch.full((64 * T * K,), float("nan"), device=dev)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, K
        d.U3, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U3.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
        d.algo = algo
        L.call("nirgan_wino6_gemm", C.byref(d), None)
        torch.cuda.synchronize()
        outs.append(M.view(64, T, K).clone())
    ref = torch.einsum("ptc,pkc->ptk", V.view(64, T, Cc).double(), U.view(64, K, Cc).double())
    a, b = outs
    bad = (a != b) | torch.isnan(b)
    e_old = (a.double() - ref).abs().max().item(); e_new = (b.double() - ref).abs().max().item()
    print(f"T={T} C={Cc} K={K}: mismatching {bad.sum().item()} of {bad.numel()}   max err vs fp64: old {e_old:.3e} new {e_new:.3e}  nan in new: {torch.isnan(b).sum().item()}")
    if bad.any():
        idx = bad.nonzero()
        pl, rows, cols = idx[:, 0], idx[:, 1], idx[:, 2]
        print("   planes:", torch.unique(pl).tolist()[:20], " rows%16:", torch.bincount(rows % 16, minlength=16).tolist(), " rows//16 (first 16):", torch.bincount(rows // 16)[:16].tolist())
        print("   cols%16:", torch.bincount(cols % 16, minlength=16).tolist(), " cols//16:", torch.bincount(cols // 16).tolist())
        i = idx[0].tolist()
        print("   first:", i, a[i[0], i[1], i[2]].item(), b[i[0], i[1], i[2]].item(), ref[i[0], i[1], i[2]].item())

for case in [(1, 96, 96, 32, 128), (1, 96, 96, 64, 128), (1, 96, 96, 128, 128), (2, 16, 16, 128, 128), (1, 96, 96, 256, 256), (4, 96, 96, 128, 128)]:
    run(*case)

# odd extents (the 532 x 532 maps of configs[3]): T = 529 = two full row tiles + one of 17 rows, two items per workgroup
for case in [(1, 133, 133, 256, 256), (1, 133, 133, 128, 256), (2, 70, 70, 256, 256)]:
    run(*case)
