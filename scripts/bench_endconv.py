"""Event-timed microbenchmark of the direct last-layer kernels (csrc/endconv.hip) at configs[1] size: B=16, 256x256, C=64."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nir-gan_amd"))
from nirgan_hip import lib as L  # noqa: E402

B, OH, OW, crop = 16, 256, 256, 0
if len(sys.argv) > 1:
    B, OH, OW, crop = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
be = L.backend()
hp, wp, H2, W2 = OH + 6, OW + 6, OH - 2 * crop, OW - 2 * crop
x = torch.randn(B, hp, wp, 64, device=dev)
w = torch.randn(49, 64, device=dev) * 0.03
bias = torch.zeros(1, device=dev)
out = torch.zeros(B, 1, H2, W2, device=dev)
dout = torch.randn(B, 1, H2, W2, device=dev)
dz = torch.zeros(be.nirgan_endconv_dz_elems(B, OH, OW), device=dev)
ws = torch.zeros(be.nirgan_endconv_ws_elems(B, OH, OW), device=dev)
gx = torch.zeros(B, hp, wp, 64, device=dev)
gw = torch.zeros(64 * 49, device=dev)
gb = torch.zeros(1, device=dev)
d = L.EndConvDesc()
d.x, d.x_hp, d.x_wp, d.B, d.OH, d.OW, d.crop, d.C, d.k = x.data_ptr(), hp, wp, B, OH, OW, crop, 64, 7
d.w, d.bias, d.act, d.out, d.dout = w.data_ptr(), bias.data_ptr(), L.ACT_TANH, out.data_ptr(), dout.data_ptr()
d.dz, d.dz_elems, d.gx, d.gw, d.gbias = dz.data_ptr(), dz.numel(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr()
d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
act_bytes = B * hp * wp * 64 * 4
fma = B * H2 * W2 * 49 * 64
print(f"B={B} {OH}x{OW} crop={crop}: activation {act_bytes / 1e6:.0f} MB, {fma / 1e9:.2f} G lane-FMAs (84 us at one v_fma per lane per 4 cycles)")
for fn in ("nirgan_endconv_fwd", "nirgan_endconv_dz", "nirgan_endconv_dgrad", "nirgan_endconv_wgrad"):
    f = getattr(be, fn)
    for _ in range(5):
        assert f(C.byref(d), st) == 0, be.nirgan_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        f(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{fn[7:]:16s} {us:8.1f} us   {act_bytes / us / 1e6:6.2f} TB/s of activation   {2 * fma / us / 1e6:6.1f} TFLOP/s")
