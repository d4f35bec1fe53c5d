"""MI355X-native counterpart of the reference's ``utils/losses.py``.

``ssim_loss(img1, img2, window_size=11)`` = 1 - kornia.metrics.ssim(img1, img2, window_size).mean()
(utils/losses.py:10-30), value AND gradient wrt ``img1`` on the device (csrc/ssimloss.hip, nirgan_ssim_loss): the term
model/pix2pix.py:233-237 adds to the generator objective when ``lambda_ssim > 0`` (0.0 in the shipped configs).
Only ``img1`` (the prediction) may require grad -- the reference differentiates it wrt ``pred`` alone (``nir`` is data).
``emd_loss(pred, target)`` (softmax + cumsum over C*H*W per sample, mean absolute CDF difference, :64-78) likewise: value and
gradient wrt ``pred`` (csrc/emdloss.hip, nirgan_emd_loss; scans in double).  The reference's training step never reaches it
(``lambda_hist > 0`` calls an undefined ``hist_loss``, pix2pix.py:239-243), so it is a library function here too.
"""
import ctypes as C

import torch

from nirgan_hip import lib as L


def _ssim_loss_device(img1: torch.Tensor, img2: torch.Tensor, window_size: int, want_grad: bool):
    if img1.shape != img2.shape or img1.dim() != 4:
        raise ValueError(f"ssim_loss needs equal-shaped [B, C, H, W] tensors, got {tuple(img1.shape)} and {tuple(img2.shape)}")
    if img1.device != img2.device or (img1.device.type != "cuda" and not L.is_emulated()):
        raise RuntimeError("nirgan_hip runs on MI355X (cuda device) only; there is no CPU path")
    p = img1.detach().to(torch.float32).contiguous()
    t = img2.detach().to(torch.float32).contiguous()
    B, Cc, H, W = p.shape
    be = L.backend()
    ws = torch.empty(int(be.nirgan_ssim_loss_ws_elems(B * Cc, H, W, int(window_size))), dtype=torch.float32, device=p.device)
    value = torch.zeros(1, dtype=torch.float32, device=p.device)
    grad = torch.zeros_like(p) if want_grad else None
    d = L.SsimLossDesc()
    d.pred, d.target, d.planes, d.H, d.W = p.data_ptr(), t.data_ptr(), B * Cc, H, W
    d.window, d.sigma, d.max_val, d.eps, d.weight = int(window_size), 1.5, 1.0, 1e-12, 1.0
    d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    d.loss, d.value, d.grad_pred = None, value.data_ptr(), (grad.data_ptr() if want_grad else None)
    st = torch.cuda.current_stream(p.device).cuda_stream if p.device.type == "cuda" else None
    L.check(be.nirgan_ssim_loss(C.byref(d), st), "ssim_loss")
    return value.reshape(()), grad


class _SsimLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, window_size):
        value, grad = _ssim_loss_device(img1, img2, window_size, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.dtype = img1.dtype
        return value

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (gout * grad).to(ctx.dtype), None, None


def ssim_loss(img1, img2, window_size=11):
    if img2.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("ssim_loss differentiates wrt img1 (the prediction) only, as the reference's generator step does")
    if img1.requires_grad and torch.is_grad_enabled():
        return _SsimLoss.apply(img1, img2, int(window_size))
    return _ssim_loss_device(img1, img2, int(window_size), want_grad=False)[0]


def _emd_loss_device(pred: torch.Tensor, target: torch.Tensor, want_grad: bool):
    if pred.shape != target.shape or pred.dim() < 2:
        raise ValueError(f"emd_loss needs equal-shaped [B, ...] tensors, got {tuple(pred.shape)} and {tuple(target.shape)}")
    if pred.device != target.device or (pred.device.type != "cuda" and not L.is_emulated()):
        raise RuntimeError("nirgan_hip runs on MI355X (cuda device) only; there is no CPU path")
    p = pred.detach().to(torch.float32).contiguous()
    t = target.detach().to(torch.float32).contiguous()
    # the reference asserts finiteness on the host (utils/losses.py:66-69): same contract, one fused check
    assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(t).all())
    B, N = p.shape[0], p[0].numel()
    be = L.backend()
    ws = torch.empty((int(be.nirgan_emd_loss_ws_bytes(B, N, 1 if want_grad else 0)) + 7) // 8, dtype=torch.float64, device=p.device)
    value = torch.zeros(1, dtype=torch.float32, device=p.device)
    grad = torch.zeros_like(p) if want_grad else None
    d = L.EmdLossDesc()
    d.pred, d.target, d.B, d.N, d.weight = p.data_ptr(), t.data_ptr(), B, N, 1.0
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 8
    d.loss, d.value, d.grad_pred = None, value.data_ptr(), (grad.data_ptr() if want_grad else None)
    st = torch.cuda.current_stream(p.device).cuda_stream if p.device.type == "cuda" else None
    L.check(be.nirgan_emd_loss(C.byref(d), st), "emd_loss")
    return value.reshape(()), grad


class _EmdLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        value, grad = _emd_loss_device(pred, target, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.dtype = pred.dtype
        return value

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (gout * grad).to(ctx.dtype), None


def emd_loss(pred, target):
    if target.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("emd_loss differentiates wrt pred only")
    if pred.requires_grad and torch.is_grad_enabled():
        return _EmdLoss.apply(pred, target)
    return _emd_loss_device(pred, target, want_grad=False)[0]
