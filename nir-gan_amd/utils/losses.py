"""MI355X-native counterpart of the forward values of the reference's ``utils/losses.py``.

``ssim_loss(img1, img2, window_size=11)`` = 1 - kornia.metrics.ssim(img1, img2, window_size).mean()
(utils/losses.py:11-30), evaluated by the fused device pass of ``utils/calculate_metrics``.  Every shipped config
has ``lambda_ssim = lambda_hist = 0`` (configs/config_px2px.yaml), so the value is a logged metric only: the result
carries no autograd graph, and ``Px2Px_PL`` refuses ``lambda_ssim > 0`` / ``lambda_hist > 0`` loudly.
``emd_loss`` (softmax + cumsum over H*W, :64-78) is not on the MI355X path.
"""
import torch

from utils.calculate_metrics import image_metrics_device


def ssim_loss(img1, img2, window_size=11):
    if (img1.requires_grad or img2.requires_grad) and torch.is_grad_enabled():
        raise NotImplementedError("ssim_loss has no backward on the MI355X path (lambda_ssim is 0.0 in every shipped config); "
                                  "call it under torch.no_grad() for the value")
    return 1.0 - image_metrics_device(img1.float(), img2.float(), window_size=window_size, max_val=1.0)[2]


def emd_loss(pred, target):
    raise NotImplementedError("emd_loss (lambda_hist) is not on the MI355X path (0.0 in every shipped config)")
