"""Times nirgan_wino6_gemm on the F(6x6,3x3) forward shape under NIRGAN_DIAG=1..4.  The DIAG variants were TEMPORARY template flags of
csrc/wino6.hip::w6_gemmp_body (1: no LDS-DMA inside the K loop, 2: no drain of the previous tile, 3: no s_waitcnt + barrier per K-step,
4: VALU stand-in for the MFMAs); they are not in the tree -- against the shipped library every row measures the shipped kernel.
Recorded result: profiles/r02_microbench_persistent_gemm_parts.txt."""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo/nir-gan_amd")
import torch
from nirgan_hip import lib as L
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
zero = torch.zeros(64, device=dev)
B, H, W, Cc, K = 16, 64, 64, 256, 256
T = 16 * 121
V = torch.randn(64 * T * Cc, device=dev); U = torch.randn(64 * K * Cc, device=dev) * 0.05; M = torch.zeros(64 * T * K, device=dev)
d = L.Wino6Desc(); d.r = 6; d.B, d.H, d.W, d.C, d.K = B, H, W, Cc, K
d.U, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
def run(tag, reps=40):
    for _ in range(5): L.call("nirgan_wino6_gemm", C.byref(d), st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): L.call("nirgan_wino6_gemm", C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{tag:40s} {ms*1e3:8.1f} us  {2.0*64*T*Cc*K/ms/1e9:6.1f} TF/s")
os.environ.pop("NIRGAN_DIAG", None)
run("warm-up (discard)", 80)
for tag, v in (("shipped", None), ("no LDS-DMA in the K loop", "1"), ("no drain of the previous tile", "2"), ("no wait + barrier per K-step", "3"), ("no MFMA (VALU stand-in)", "4"), ("shipped again", None)):
    if v is None: os.environ.pop("NIRGAN_DIAG", None)
    else: os.environ["NIRGAN_DIAG"] = v
    run(tag)
