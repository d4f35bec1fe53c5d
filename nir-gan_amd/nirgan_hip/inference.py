"""Inference helpers around the generator forward.

The reference's create_synthetic_dataset.py (:100-118) runs ``model(hr)`` under no_grad on whole tiles,
then post-processes on the CPU and stores fp16 ``.npz`` files (:49-52, :117).  ``predict_tiled`` adds
what large scenes need on the GPU side: the scene is cut into tiles that overlap by twice the model's
reflect padding trick (model/pix2pix.py:91-93,107-108 hides tile-edge artefacts with pad-10 / crop); the
overlap is discarded, so every output pixel comes from a tile interior.
"""
from __future__ import annotations

import os

import numpy as np
import torch


@torch.no_grad()
def predict_tiled(model, rgb: torch.Tensor, tile: int = 512, margin: int = 16, batch: int = 8, embeds=None) -> torch.Tensor:
    """rgb: B x 3 x H x W (any H, W >= 4) -> B x 1 x H x W.  ``tile`` is the network input size (multiple of 4);
    ``margin`` pixels on every side of a tile are context only.

    On the device the reflect-padded scene is never materialised: one gather launch cuts a batch of overlapping tiles straight out
    of the scene (nirgan_tile_gather reflects at the borders like ``F.pad(mode='reflect')``), the model runs on the batch, one
    scatter launch writes the tiles' cores back (nirgan_tile_scatter) -- no per-tile Python.  Device tensors only (no CPU path; the
    plain-torch statement of the same tiling lives with the oracle: oracle/nirgan_oracle.py::predict_tiled)."""
    assert tile % 4 == 0 and 0 <= margin < tile // 2
    B, C3, H, W = rgb.shape
    core = tile - 2 * margin
    from . import lib as L
    if rgb.device.type != "cuda" and not L.is_emulated():
        raise RuntimeError("predict_tiled runs on MI355X (cuda tensors) only; there is no CPU path")
    be = L.backend()
    scene = rgb.detach().to(torch.float32).contiguous()
    total = int(be.nirgan_tile_count(B, H, W, tile, margin))
    if total <= 0:
        raise ValueError(f"predict_tiled: bad tiling (scene {H}x{W}, tile {tile}, margin {margin})")
    per_image = total // B
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=rgb.device)
    st = torch.cuda.current_stream(rgb.device).cuda_stream if rgb.device.type == "cuda" else None
    tiles = torch.empty(min(batch, total), C3, tile, tile, dtype=torch.float32, device=rgb.device)
    for first in range(0, total, batch):
        n = min(batch, total - first)
        L.check(be.nirgan_tile_gather(scene.data_ptr(), B, C3, H, W, tile, margin, first, n, tiles.data_ptr(), st), "tile_gather")
        x = tiles[:n]
        if embeds is None:
            pred = model(x)
        else:
            idx = torch.arange(first, first + n, device=embeds.device) // per_image      # tile -> scene it was cut from
            pred = model(x, embeds.index_select(0, idx))
        pred = pred.detach().to(torch.float32).contiguous()
        L.check(be.nirgan_tile_scatter(pred.data_ptr(), B, 1, H, W, tile, margin, first, n, out.data_ptr(), st), "tile_scatter")
    return out.to(rgb.dtype)


def save_nir_npz(pred_nir: torch.Tensor, out_path: str, name: str) -> str:
    """fp16 compressed .npz with key 'nir', as create_synthetic_dataset.py:49-52,115-118 writes it."""
    fn = os.path.join(out_path, f"{name}")
    np.savez_compressed(fn, nir=pred_nir.detach().to(torch.float16).cpu().numpy())
    return fn if fn.endswith(".npz") else fn + ".npz"


@torch.no_grad()
def resize_bilinear(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """F.interpolate(x, size=(H, W), mode='bilinear', align_corners=False) for [B, 1, h, w] on the device."""
    from . import lib as L
    dev = x.device
    st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    x = x.detach().to(torch.float32).contiguous()
    up = torch.empty(x.shape[0], 1, H, W, dtype=torch.float32, device=dev)
    L.check(L.backend().nirgan_bilinear_fwd(x.data_ptr(), x.shape[0], x.shape[-2], x.shape[-1], up.data_ptr(), H, W, st), "bilinear")
    return up


@torch.no_grad()
def histogram_match(image: torch.Tensor, reference: torch.Tensor) -> torch.Tensor:
    """create_synthetic_dataset.py:34-47 on the device: ``reference`` ([B, 1, h, w], e.g. the Sentinel-2 NIR band) is
    resized to the tile with F.interpolate(mode='bilinear', align_corners=False) semantics (nirgan_bilinear_fwd), then
    every tile of ``image`` ([B, 1, H, W]) is matched to its reference plane (skimage.exposure.match_histograms,
    channel_axis=None -> nirgan_hist_match).  Returns [B, 1, H, W] like the reference's helper; stays on the GPU."""
    import ctypes as C
    from . import lib as L
    if image.dim() != 4 or reference.dim() != 4 or image.shape[:2] != reference.shape[:2] or image.shape[1] != 1:
        raise ValueError(f"image/reference must be [B, 1, H, W] / [B, 1, h, w], got {tuple(image.shape)} and {tuple(reference.shape)}")
    if image.device != reference.device or (image.device.type != "cuda" and not L.is_emulated()):
        raise RuntimeError("nirgan_hip runs on MI355X (cuda device) only; there is no CPU path")
    dev = image.device
    be = L.backend()
    st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    img = image.detach().to(torch.float32).contiguous()
    ref = reference.detach().to(torch.float32).contiguous()
    B, _, H, W = img.shape
    if ref.shape[-2:] != (H, W):
        ref = resize_bilinear(ref, H, W)
    N = H * W
    ws = torch.empty(int(be.nirgan_hist_match_ws_bytes(B, N)) // 8, dtype=torch.int64, device=dev)
    out = torch.empty_like(img)
    d = L.HistMatchDesc()
    d.image, d.reference, d.B, d.N = img.data_ptr(), ref.data_ptr(), B, N
    d.ws, d.ws_bytes, d.out = ws.data_ptr(), ws.numel() * 8, out.data_ptr()
    L.check(be.nirgan_hist_match(C.byref(d), st), "hist_match")
    return out
