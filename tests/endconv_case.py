"""Shared body of the direct Conv2d(64, 1, 7) (+bias, tanh, crop) tests: the four C-ABI entry points against torch's conv2d and its
autograd in fp64 (model/networks.py:366-368).  Runs against whichever backend is active: the HIP library on the MI355X (-m gpu) or the
numpy emulation of the same ABI on the CPU (which is what the emulated trainer tests run on)."""
import ctypes as C

import torch
import torch.nn.functional as F

from nirgan_hip import lib as L


def run_endconv(device, B, OH, OW, crop, seed=0, tol=2e-5, act=L.ACT_TANH):
    g = torch.Generator().manual_seed(seed)
    be = L.backend()
    hp, wp, H2, W2 = OH + 6, OW + 6, OH - 2 * crop, OW - 2 * crop
    x = torch.randn(B, hp, wp, 64, generator=g)
    wt = torch.randn(1, 64, 7, 7, generator=g) * 0.03
    bias = torch.randn(1, generator=g) * 0.1
    dout = torch.randn(B, 1, H2, W2, generator=g)

    x64 = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w64, b64 = wt.double().requires_grad_(True), bias.double().requires_grad_(True)
    z = F.conv2d(x64, w64, b64)
    z = z[:, :, crop:OH - crop, crop:OW - crop]
    ref = torch.tanh(z) if act == L.ACT_TANH else z
    ref.backward(dout.double())

    dev = torch.device(device)
    xd = x.to(dev)
    wp_ = wt[0].permute(1, 2, 0).reshape(49, 64).contiguous().to(dev)          # [t][c], the tap-plane forward pack
    bd, dd = bias.to(dev), dout.to(dev)
    out = torch.full((B, 1, H2, W2), 7.0, device=dev)
    dz = torch.full((be.nirgan_endconv_dz_elems(B, OH, OW),), 3.0, device=dev)   # stale contents: the kernel rewrites the border
    ws = torch.zeros(be.nirgan_endconv_ws_elems(B, OH, OW), device=dev)
    gx = torch.full((B, hp, wp, 64), 5.0, device=dev)
    gw = torch.full((64 * 49,), 5.0, device=dev)
    gb = torch.full((1,), 0.25, device=dev)
    d = L.EndConvDesc()
    d.x, d.x_hp, d.x_wp, d.B, d.OH, d.OW, d.crop, d.C, d.k = xd.data_ptr(), hp, wp, B, OH, OW, crop, 64, 7
    d.w, d.bias, d.act, d.out, d.dout = wp_.data_ptr(), bd.data_ptr(), act, out.data_ptr(), dd.data_ptr()
    d.dz, d.dz_elems, d.gx, d.gw, d.gbias = dz.data_ptr(), dz.numel(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr()
    d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    st = None
    if dev.type == "cuda":
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for fn in ("nirgan_endconv_fwd", "nirgan_endconv_dz", "nirgan_endconv_wgrad", "nirgan_endconv_dgrad"):
        rc = getattr(be, fn)(C.byref(d), st)
        assert rc == 0, (fn, be.nirgan_last_error())
    if dev.type == "cuda":
        torch.cuda.synchronize()

    def close(a, b, what, t=tol):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        assert torch.isfinite(a).all(), what
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert err <= t * max(scale, 1e-20), f"{what}: err {err:.3e} of {scale:.3e}"

    close(out, ref, "forward")
    close(gx.permute(0, 3, 1, 2), x64.grad, "data gradient")
    close(gw.reshape(1, 64, 7, 7), w64.grad, "weight gradient", 1e-4)
    close(gb - 0.25, b64.grad, "bias gradient (accumulated)", 1e-4)
    # a second wgrad launch gives the same bits (fixed summation order)
    gw2 = torch.zeros_like(gw)
    d.gw = gw2.data_ptr()
    assert be.nirgan_endconv_wgrad(C.byref(d), st) == 0
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert torch.equal(gw2.cpu(), gw.cpu())
    # descriptor validation
    d.C = 32
    assert be.nirgan_endconv_fwd(C.byref(d), st) == -1 and b"Conv2d(64, 1, 7)" in be.nirgan_last_error()
