"""MI355X-native counterpart of the reference's ``utils/calculate_metrics.py``.

``calculate_metrics(pred, target, phase)`` keeps the reference's signature and result
(utils/calculate_metrics.py:5-36): a dict ``{phase/L1, phase/L2, phase/PSNR, phase/SSIM}`` of Python floats
(F.l1_loss, F.mse_loss, kornia.metrics.psnr(.., 1.0), kornia.metrics.ssim(window_size=5, max_val=1.).mean()).
The reference moves both tensors to the CPU every 10th batch for this (model/pix2pix.py:183-186); here it is ONE
fused pass on the device (nirgan_image_metrics).  ``image_metrics_device`` returns the three means as a device
tensor without synchronising, for callers that log asynchronously.
"""
import ctypes as C
import math

import torch

from nirgan_hip import lib as L


def image_metrics_device(pred: torch.Tensor, target: torch.Tensor, window_size: int = 5, max_val: float = 1.0,
                         sigma: float = 1.5, eps: float = 1e-12) -> torch.Tensor:
    """[mean |d|, mean d^2, mean SSIM map] as a 3-element fp32 tensor on the inputs' device (no host sync)."""
    if pred.shape != target.shape or pred.dim() != 4:
        raise ValueError(f"pred/target must be equal-shaped [B, C, H, W] tensors, got {tuple(pred.shape)} and {tuple(target.shape)}")
    if pred.device != target.device or (pred.device.type != "cuda" and not L.is_emulated()):
        raise RuntimeError("nirgan_hip runs on MI355X (cuda device) only; there is no CPU path")
    p = pred.detach().to(torch.float32).contiguous()
    t = target.detach().to(torch.float32).contiguous()
    B, Cc, H, W = p.shape
    be = L.backend()
    ws = torch.empty(int(be.nirgan_image_metrics_ws_elems(B * Cc, H, W)), dtype=torch.float32, device=p.device)
    means = torch.empty(3, dtype=torch.float32, device=p.device)
    d = L.MetricsDesc()
    d.pred, d.target, d.planes, d.H, d.W = p.data_ptr(), t.data_ptr(), B * Cc, H, W
    d.window, d.sigma, d.max_val, d.eps = int(window_size), float(sigma), float(max_val), float(eps)
    d.ws, d.ws_elems, d.means = ws.data_ptr(), ws.numel(), means.data_ptr()
    st = torch.cuda.current_stream(p.device).cuda_stream if p.device.type == "cuda" else None
    L.check(be.nirgan_image_metrics(C.byref(d), st), "image_metrics")
    return means


def calculate_metrics(pred, target, phase="train"):
    """
    Calculate PSNR, SSIM, L1 loss, and L2 loss between pred and target ([B, C, H, W]); dict of floats.
    """
    l1, l2, ssim = image_metrics_device(pred, target, window_size=5, max_val=1.0).tolist()
    psnr = 10.0 * math.log10(1.0 / l2) if l2 > 0.0 else float("inf")
    return {
        phase + '/L1': l1,
        phase + '/L2': l2,
        phase + '/PSNR': psnr,
        phase + '/SSIM': ssim,
    }
