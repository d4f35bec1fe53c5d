"""MI355X-native counterpart of the reference's ``utils/losses.py``.

``ssim_loss(img1, img2, window_size=11)`` = 1 - kornia.metrics.ssim(img1, img2, window_size).mean()
(utils/losses.py:10-30), value AND gradient wrt ``img1`` on the device (csrc/ssimloss.hip, nirgan_ssim_loss): the term
model/pix2pix.py:233-237 adds to the generator objective when ``lambda_ssim > 0`` (0.0 in the shipped configs).
Only ``img1`` (the prediction) may require grad -- the reference differentiates it wrt ``pred`` alone (``nir`` is data).
``emd_loss`` (softmax + cumsum over H*W, :64-78) is dead code in the reference (training_step calls an undefined
``hist_loss`` when ``lambda_hist > 0``, pix2pix.py:239-243) and is not on the MI355X path.
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


def emd_loss(pred, target):
    raise NotImplementedError("emd_loss (lambda_hist) is not on the MI355X path: the reference's training step calls an undefined "
                              "hist_loss for it (model/pix2pix.py:239-243), 0.0 in every shipped config")
