"""The fused two-optimizer Pix2Pix step on the HIP engines.

One call = one batch of the reference's loop (model/pix2pix.py:165-257 under Lightning's
two-optimizer schedule, configure_optimizers :485-492):

    optimizer 0 (D): pred = G(rgb);  loss_D = MSE(D(cat(rgb, pred.detach())), 0) + MSE(D(cat(rgb, nir)), 1);
                     backward; Adam(D)
    optimizer 1 (G): D frozen (already updated);  loss_G = l_GAN*MSE(D(cat(rgb, pred)), 1) + l_L1*L1(pred, nir)
                     [+ l_rs * sum_i w_i * crit(index_i(nir), index_i(pred))];  backward; Adam(G)

Differences from the reference that do not change results: the generator forward runs once
(its parameters do not change between the two optimizer passes, so the second forward would
reproduce ``pred`` bit for bit); the fake and real PatchGAN passes of the D step run as one
batch of 2B (InstanceNorm is per sample); torch.cat never materialises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L
from .flat import FlatParams
from .nets import DiscriminatorEngine, GeneratorEngine

RS_KEYS = ["ndvi", "ndwi", "gndvi", "savi", "msavi", "evi"]


class _ShapeState:
    """Engines, staging tensors and descriptors of one (B, H, W): built once per resolution bucket and kept
    (configs[4] draws 128/256/512 tiles step by step; 288 GB of HBM hold all buckets side by side)."""

    def __init__(self, tr: "Pix2PixTrainer", B, H, W):
        dev = tr.flatG.flat.device
        gp, gg = tr.flatG.param_views(), tr.flatG.grad_views()
        dp, dg = tr.flatD.param_views(), tr.flatD.grad_views()
        self.G = GeneratorEngine(gp, gg, tr.n_blocks, B, H, W, data_pad=tr.padding, inject=tr.inject, precision=tr.precision)
        self.D2 = DiscriminatorEngine(dp, dg, 2 * B, H, W, precision=tr.precision)
        self.D1 = DiscriminatorEngine(dp, dg, B, H, W, precision=tr.precision)
        self.rgb = torch.zeros(B, 3, H, W, dtype=torch.float32, device=dev)
        self.nir = torch.zeros(B, 1, H, W, dtype=torch.float32, device=dev)
        self.n_patch = self.D1.B * self.D1.out[0].numel()
        # pixel-loss descriptor (static pointers)
        d = L.PixLossDesc()
        d.rgb, d.nir, d.pred = self.rgb.data_ptr(), self.nir.data_ptr(), self.G.pred.data_ptr()
        d.B, d.H, d.W = B, H, W
        d.w_l1 = tr.lambda_l1
        for k in RS_KEYS:
            w = float(tr.rs_weights.get("lambda_" + k, 0.0)) if tr.lambda_rs > 0.0 else 0.0
            setattr(d, "w_" + k, tr.lambda_rs * w if w > 0.0 else 0.0)
        d.criterion, d.log_all = tr.rs_criterion, 0
        d.extra, d.extra_cs, d.extra_c, d.extra_scale = self.D1.gpred.data_ptr(), 1, 0, 1.0
        d.sums, d.grad_pred = tr.losses.data_ptr() + 3 * 4, self.G.dpred.data_ptr()
        self.pix = d


class Pix2PixTrainer:
    def __init__(self, netG: torch.nn.Module, netD: torch.nn.Module, *, n_blocks: int, lr=2e-4, beta1=0.5,
                 lambda_gan=1.0, lambda_l1=100.0, lambda_rs=0.0, rs_weights: Optional[Dict[str, float]] = None,
                 rs_criterion="l1", padding=0, inject: Optional[dict] = None, reducer=None, precision="fp32"):
        self.netG, self.netD = netG, netD
        self.flatG = netG._flat() if hasattr(netG, "_flat") else FlatParams(netG)
        self.flatD = netD._flat() if hasattr(netD, "_flat") else FlatParams(netD)
        self.n_blocks, self.padding, self.inject = n_blocks, padding, inject
        self.lr, self.beta1 = lr, beta1
        self.lr_d, self.lr_g = None, None   # per-network overrides (ReduceLROnPlateau steps them separately); None = self.lr
        self.lambda_gan, self.lambda_l1, self.lambda_rs = float(lambda_gan), float(lambda_l1), float(lambda_rs)
        self.rs_weights = rs_weights or {}
        if rs_criterion not in ("l1", "l2"):
            raise NotImplementedError(f"Criterion '{rs_criterion}' not implemented. 'l1' or 'l2' are supported.")
        self.rs_criterion = 0 if rs_criterion == "l1" else 1
        self.reducer = reducer            # parallel.GradReducer or None
        self.precision = precision        # 'fp32' | 'bf16' | 'bf16x3' (engine.precision_code)
        self._states: Dict[tuple, _ShapeState] = {}
        self._shape = None
        self.losses = None
        self.steps = 0

    # ------------------------------------------------------------------ engines for one shape
    def _prepare(self, B, H, W):
        rebuilt = self.flatG.ensure() | self.flatD.ensure()
        if rebuilt:
            self._states.clear()          # the flat ranges moved: every descriptor holds stale pointers
        if self.losses is None or rebuilt:
            # 0 D_fake 1 D_real 2 G_gan 3.. pix sums[7]
            self.losses = torch.zeros(16, dtype=torch.float32, device=self.flatG.flat.device)
        st = self._states.get((B, H, W))
        if st is None:
            st = self._states[(B, H, W)] = _ShapeState(self, B, H, W)
        self._shape = (B, H, W)
        self.G, self.D2, self.D1, self.rgb, self.nir = st.G, st.D2, st.D1, st.rgb, st.nir
        self._n_patch, self._pix = st.n_patch, st.pix

    # ------------------------------------------------------------------ one batch
    def step(self, rgb: torch.Tensor, nir: torch.Tensor, embeds: Optional[torch.Tensor] = None) -> "LossView":
        B, _, H, W = rgb.shape
        self._prepare(B, H, W)
        st = self.G.ctx.stream()
        be = L.backend()
        self.rgb.copy_(rgb)
        self.nir.copy_(nir)
        G, D2, D1 = self.G, self.D2, self.D1
        npatch = self._n_patch
        lp = self.losses.data_ptr()
        L.check(be.nirgan_fill(lp, 16, 0.0, st), "fill")
        # ---- generator forward (once)
        pred = G.forward(self.rgb, embeds, version=self.flatG.version)
        # ---- optimizer 0: discriminator on [fake ; real]
        D2.forward(parts=[(self.rgb, 0, 0), (pred, 0, 3), (self.rgb, B, 0), (self.nir, B, 3)], version=self.flatD.version)
        out, dout = D2.out.data_ptr(), D2.dout.data_ptr()
        L.check(be.nirgan_lsgan(out, npatch, 0.0, 1.0, lp, dout, st), "lsgan")
        L.check(be.nirgan_lsgan(out + npatch * 4, npatch, 1.0, 1.0, lp + 4, dout + npatch * 4, st), "lsgan")
        D2.backward(None, frozen=False, version=self.flatD.version)
        if self.reducer is not None:
            self.reducer.all_reduce_mean(self.flatD.grad)
        self.flatD.adam_step(self.lr if self.lr_d is None else self.lr_d, self.beta1, stream=st)
        # ---- optimizer 1: generator against the updated, frozen discriminator
        D1.forward(parts=[(self.rgb, 0, 0), (pred, 0, 3)], version=self.flatD.version)
        L.check(be.nirgan_lsgan(D1.out.data_ptr(), npatch, 1.0, self.lambda_gan, lp + 8, D1.dout.data_ptr(), st), "lsgan")
        D1.backward(None, frozen=True, version=self.flatD.version, pred_only=True)
        L.check(be.nirgan_pix_loss(C.byref(self._pix), st), "pix_loss")
        G.backward(None, version=self.flatG.version)
        if self.reducer is not None:
            self.reducer.all_reduce_mean(self.flatG.grad)
        self.flatG.adam_step(self.lr if self.lr_g is None else self.lr_g, self.beta1, stream=st)
        self.steps += 1
        return LossView(self, B * H * W)


class LossView:
    """Lazy view of the step's loss scalars (reading synchronises; the hot loop never does)."""

    def __init__(self, tr: Pix2PixTrainer, npix: int):
        self.tr, self.npix = tr, npix

    def as_dict(self) -> Dict[str, float]:
        tr = self.tr
        v = tr.losses.detach().cpu().tolist()
        out = {"loss_D_fake": v[0], "loss_D_real": v[1], "loss_D": v[0] + v[1]}
        gan_w = v[2]
        out["loss_G_gan"] = gan_w / tr.lambda_gan if tr.lambda_gan else 0.0
        out["loss_G_l1"] = v[3] / self.npix
        loss_g = gan_w + tr.lambda_l1 * out["loss_G_l1"]
        if tr.lambda_rs > 0.0:
            rs = 0.0
            for i, k in enumerate(RS_KEYS):
                w = float(tr.rs_weights.get("lambda_" + k, 0.0))
                if w > 0.0:
                    rs += w * v[4 + i] / self.npix
            out["loss_G_rs"] = rs
            loss_g += tr.lambda_rs * rs
        out["loss_G"] = loss_g
        return out
