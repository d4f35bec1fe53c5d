"""The fused two-optimizer Pix2Pix step on the HIP engines.

One call = one batch of the reference's loop (model/pix2pix.py:165-257 under Lightning's
two-optimizer schedule, configure_optimizers :485-492):

    optimizer 0 (D): pred = G(rgb);  loss_D = MSE(D(cat(rgb, pred.detach())), 0) + MSE(D(cat(rgb, nir)), 1);
                     backward; Adam(D)
    optimizer 1 (G): D frozen (already updated);  loss_G = l_GAN*MSE(D(cat(rgb, pred)), 1) + l_L1*L1(pred, nir)
                     [+ l_rs * sum_i w_i * crit(index_i(nir), index_i(pred))] [+ l_ssim * (1 - mean SSIM_11(pred, nir))];
                     backward; Adam(G)

Differences from the reference that do not change results: the generator forward runs once
(its parameters do not change between the two optimizer passes, so the second forward would
reproduce ``pred`` bit for bit); the fake and real PatchGAN passes of the D step run as one
batch of 2B (InstanceNorm is per sample); torch.cat never materialises.  Optionally
(``micro_batches=2``) the batch is cut into parts that run concurrently on separate HIP streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L
from .options import OPT
from .flat import FlatParams
from .nets import DiscriminatorEngine, GeneratorEngine

RS_KEYS = ["ndvi", "ndwi", "gndvi", "savi", "msavi", "evi"]


class _Micro:
    """Engines, staging tensors and descriptors of one micro-batch [lo, hi) of one (B, H, W)."""

    def __init__(self, tr: "Pix2PixTrainer", lo, hi, H, W, gradG, gradD, scale):
        dev = tr.flatG.flat.device
        B = hi - lo
        self.lo, self.hi, self.B, self.scale = lo, hi, B, scale
        gp, dp = tr.flatG.param_views(), tr.flatD.param_views()
        self.G = GeneratorEngine(gp, gradG, tr.n_blocks, B, H, W, data_pad=tr.padding, inject=tr.inject, precision=tr.precision)
        self.D2 = DiscriminatorEngine(dp, gradD, 2 * B, H, W, precision=tr.precision)
        self.D1 = DiscriminatorEngine(dp, gradD, B, H, W, precision=tr.precision)
        self.rgb = torch.zeros(B, 3, H, W, dtype=torch.float32, device=dev)
        self.nir = torch.zeros(B, 1, H, W, dtype=torch.float32, device=dev)
        self.n_patch = self.D1.B * self.D1.out[0].numel()
        # pixel-loss descriptor (static pointers); `scale` = this micro-batch's share of the batch mean
        d = L.PixLossDesc()
        d.rgb, d.nir, d.pred = self.rgb.data_ptr(), self.nir.data_ptr(), self.G.pred.data_ptr()
        d.B, d.H, d.W = B, H, W
        d.w_l1 = tr.lambda_l1 * scale
        for k in RS_KEYS:
            w = float(tr.rs_weights.get("lambda_" + k, 0.0)) if tr.lambda_rs > 0.0 else 0.0
            setattr(d, "w_" + k, tr.lambda_rs * w * scale if w > 0.0 else 0.0)
        d.criterion, d.log_all = tr.rs_criterion, 0
        d.extra, d.extra_cs, d.extra_c, d.extra_scale = self.D1.gpred.data_ptr(), 1, 0, 1.0
        d.sums, d.grad_pred = tr.losses.data_ptr() + 3 * 4, self.G.dpred.data_ptr()
        self.pix_ws = torch.zeros(L.PIX_LOSS_WS_ELEMS, dtype=torch.float32, device=dev)      # this micro-batch's own (streams run concurrently)
        d.ws, d.ws_elems = self.pix_ws.data_ptr(), self.pix_ws.numel()
        self.pix = d
        # optional SSIM term (model/pix2pix.py:233-237; lambda_ssim is 0.0 in the shipped configs): runs after the pixel losses and
        # ADDS its gradient to dpred, its weighted value to losses[10]
        self.ssim = None
        if tr.lambda_ssim > 0.0:
            be = L.backend()
            self.ssim_ws = torch.zeros(int(be.nirgan_ssim_loss_ws_elems(B, H, W, 11)), dtype=torch.float32, device=dev)
            sd = L.SsimLossDesc()
            sd.pred, sd.target, sd.planes, sd.H, sd.W = self.G.pred.data_ptr(), self.nir.data_ptr(), B, H, W
            sd.window, sd.sigma, sd.max_val, sd.eps = 11, 1.5, 1.0, 1e-12
            sd.weight = tr.lambda_ssim * scale
            sd.ws, sd.ws_elems = self.ssim_ws.data_ptr(), self.ssim_ws.numel()
            sd.loss, sd.value, sd.grad_pred = tr.losses.data_ptr() + 10 * 4, None, self.G.dpred.data_ptr()
            self.ssim = sd


class _ShapeState:
    """Everything one (B, H, W) needs, built once per resolution bucket and kept (configs[4] draws 128/256/512 tiles
    step by step; 288 GB of HBM hold all buckets side by side).  With micro-batches the batch is cut into equal
    parts that run on separate HIP streams: instance norm is per sample and every loss is a mean, so the sum of the
    parts' gradients (each scaled by its share) is the batch gradient; the HBM-bound kernels of one part run under
    the matrix-pipe kernels of the other."""

    def __init__(self, tr: "Pix2PixTrainer", B, H, W):
        n = tr._micro_count(B, H, W)
        self.n = n
        per = B // n
        self.extraG = [torch.zeros_like(tr.flatG.grad) for _ in range(n - 1)]
        self.extraD = [torch.zeros_like(tr.flatD.grad) for _ in range(n - 1)]

        def views(flat, buf):
            return {k: buf[o:o + cnt].view(shp) for k, (o, cnt, shp) in flat.slices.items()}
        self.micros = []
        for i in range(n):
            gG = tr.flatG.grad_views() if i == 0 else views(tr.flatG, self.extraG[i - 1])
            gD = tr.flatD.grad_views() if i == 0 else views(tr.flatD, self.extraD[i - 1])
            self.micros.append(_Micro(tr, i * per, (i + 1) * per, H, W, gG, gD, 1.0 / n))
        dev = tr.flatG.flat.device
        self.streams = [None] + [torch.cuda.Stream(dev) if dev.type == "cuda" else None for _ in range(n - 1)]
        # data parallel, single part: three gradient buckets per network -- the tail of the flat gradient goes to RCCL from inside
        # the backward plan, as soon as the launches that complete it are on the stream (parallel.GradReducer.begin), the middle in
        # front of the first layer's backward; only the first layer's gradient (the head) follows after the plan.  With micro-batches the side parts are added after the plans: one bucket, after the join.
        self.bucketed = tr.reducer is not None and n == 1 and OPT.dp_buckets
        self.headD = self.headG = None
        if self.bucketed:
            m = self.micros[0]
            self.headD = self._hook_tail(tr, m.D2, tr.flatD)
            self.headG = self._hook_tail(tr, m.G, tr.flatG)

    @staticmethod
    def _hook_tail(tr, eng, flat):
        """Insert the ``begin`` of the tail bucket and of the middle bucket into eng.bwd (three buckets per network: the tail as soon as
        the last layers' gradients are final, the middle -- everything else but the FIRST layer's gradient -- in front of the first
        layer's backward); returns what is left for after the plan: the first layer's slice of the flat gradient (16 / 38 KB)."""
        def span(first, last):
            lo = flat.slices[first][0]
            o, k, _ = flat.slices[last]
            hi = min(flat.total, o + -(-k // 4) * 4)
            assert 0 <= lo < hi <= flat.total
            return lo, hi
        index, first, last = eng.bwd_tail
        lo, hi = span(first, last)
        tail = flat.grad[lo:hi]
        eng.bwd.insert_hook(index, lambda: tr.reducer.begin(tail))
        rest = [(0, lo), (hi, flat.total)]                 # what the tail leaves, as index ranges
        mid = getattr(eng, "bwd_mid", None)
        if mid is None:
            return [flat.grad[a:b] for a, b in rest if b > a]
        mindex, hfirst, hlast = mid
        hlo, hhi = span(hfirst, hlast)
        assert mindex >= index and not (hlo < hi and lo < hhi), "the head bucket lies outside the tail"
        parts = []
        for a, b in rest:                                  # the ranges minus the head
            for c, d in ((a, min(b, hlo)), (max(a, hhi), b)):
                if d > c:
                    parts.append(flat.grad[c:d])
        if parts:
            eng.bwd.insert_hook(mindex, lambda: [tr.reducer.begin(p_) for p_ in parts])
        return [flat.grad[hlo:hhi]]


class _on_stream:
    """`with torch.cuda.stream(s)` that is a no-op for the launch stream itself (s is None) and on the CPU test seam."""

    def __init__(self, s):
        self.cm = torch.cuda.stream(s) if s is not None else None

    def __enter__(self):
        if self.cm is not None:
            self.cm.__enter__()

    def __exit__(self, *exc):
        if self.cm is not None:
            self.cm.__exit__(*exc)


class Pix2PixTrainer:
    def __init__(self, netG: torch.nn.Module, netD: torch.nn.Module, *, n_blocks: int, lr=2e-4, beta1=0.5,
                 lambda_gan=1.0, lambda_l1=100.0, lambda_rs=0.0, rs_weights: Optional[Dict[str, float]] = None,
                 rs_criterion="l1", padding=0, inject: Optional[dict] = None, reducer=None, precision="fp32",
                 micro_batches: int = 1, lambda_ssim: float = 0.0):
        self.netG, self.netD = netG, netD
        self.flatG = netG._flat() if hasattr(netG, "_flat") else FlatParams(netG)
        self.flatD = netD._flat() if hasattr(netD, "_flat") else FlatParams(netD)
        self.n_blocks, self.padding, self.inject = n_blocks, padding, inject
        self.lr, self.beta1 = lr, beta1
        self.lr_d, self.lr_g = None, None   # per-network overrides (ReduceLROnPlateau steps them separately); None = self.lr
        self.lambda_gan, self.lambda_l1, self.lambda_rs = float(lambda_gan), float(lambda_l1), float(lambda_rs)
        self.lambda_ssim = float(lambda_ssim)
        self.real_label, self.fake_label = 1.0, 0.0      # GANLoss's target_real_label / target_fake_label buffers (networks.py:229-230)
        self.rs_weights = rs_weights or {}
        if rs_criterion not in ("l1", "l2"):
            raise NotImplementedError(f"Criterion '{rs_criterion}' not implemented. 'l1' or 'l2' are supported.")
        self.rs_criterion = 0 if rs_criterion == "l1" else 1
        self.reducer = reducer            # parallel.GradReducer or None
        self.precision = precision        # 'fp32' | 'bf16' | 'bf16x3' (engine.precision_code)
        if int(micro_batches) < 1:
            raise ValueError("micro_batches must be >= 1")
        self.micro_batches = int(micro_batches)   # parts of the batch run on separate HIP streams (1 = off)
        self._states: Dict[tuple, _ShapeState] = {}
        self._shape = None
        self.losses = None
        self.steps = 0
        self._synced_ptrs = None

    def _sync_initial_weights(self):
        """What DDP does when it wraps a module (the reference: Lightning strategy "ddp", train.py:118-120): every rank starts
        from rank 0's parameters.  Redone when the flat ranges were re-materialised (``.to(device)``)."""
        if self.reducer is None:
            return
        ptrs = (self.flatG.flat.data_ptr(), self.flatD.flat.data_ptr())
        if ptrs != self._synced_ptrs:
            self.reducer.broadcast_params(self.flatG)
            self.reducer.broadcast_params(self.flatD)
            self._synced_ptrs = ptrs

    def _micro_count(self, B, H, W) -> int:
        n = self.micro_batches
        while n > 1 and B % n:
            n -= 1
        return n

    # ------------------------------------------------------------------ engines for one shape
    def _prepare(self, B, H, W):
        rebuilt = self.flatG.ensure() | self.flatD.ensure()
        self._sync_initial_weights()
        if rebuilt:
            self._states.clear()          # the flat ranges moved: every descriptor holds stale pointers
        if self.losses is None or rebuilt:
            # 0 D_fake 1 D_real 2 G_gan 3.. pix sums[7] 10 weighted SSIM term
            self.losses = torch.zeros(16, dtype=torch.float32, device=self.flatG.flat.device)
        st = self._states.get((B, H, W))
        if st is None:
            st = self._states[(B, H, W)] = _ShapeState(self, B, H, W)
        self._shape = (B, H, W)
        self._state = st
        m0 = st.micros[0]                 # single-part view (tests, probes): the first micro-batch's engines
        self.G, self.D2, self.D1, self.rgb, self.nir = m0.G, m0.D2, m0.D1, m0.rgb, m0.nir

    @property
    def pred(self) -> torch.Tensor:
        """The step's generator output for the whole batch (B x 1 x H x W)."""
        ms = self._state.micros
        return ms[0].G.pred if len(ms) == 1 else torch.cat([m.G.pred for m in ms], 0)

    # ------------------------------------------------------------------ the two optimizer passes of one micro-batch
    def _d_pass(self, m: _Micro, embeds):
        be, st = L.backend(), m.G.ctx.stream()
        lp = self.losses.data_ptr()
        B, npatch = m.B, m.n_patch
        pred = m.G.forward(m.rgb, embeds, version=self.flatG.values_version())              # generator forward (once)
        m.D2.forward(parts=[(m.rgb, 0, 0), (pred, 0, 3), (m.rgb, B, 0), (m.nir, B, 3)], version=self.flatD.values_version())
        out, dout = m.D2.out.data_ptr(), m.D2.dout.data_ptr()
        L.check(be.nirgan_lsgan(out, npatch, self.fake_label, m.scale, lp, dout, st), "lsgan")
        L.check(be.nirgan_lsgan(out + npatch * 4, npatch, self.real_label, m.scale, lp + 4, dout + npatch * 4, st), "lsgan")
        m.D2.backward(None, frozen=False, version=self.flatD.values_version())

    def _g_pass(self, m: _Micro):
        be, st = L.backend(), m.G.ctx.stream()
        lp = self.losses.data_ptr()
        m.D1.forward(parts=[(m.rgb, 0, 0), (m.G.pred, 0, 3)], version=self.flatD.values_version())
        L.check(be.nirgan_lsgan(m.D1.out.data_ptr(), m.n_patch, self.real_label, self.lambda_gan * m.scale, lp + 8, m.D1.dout.data_ptr(), st), "lsgan")
        m.D1.backward(None, frozen=True, version=self.flatD.values_version(), pred_only=True)
        L.check(be.nirgan_pix_loss(C.byref(m.pix), st), "pix_loss")
        if m.ssim is not None:
            L.check(be.nirgan_ssim_loss(C.byref(m.ssim), st), "ssim_loss")
        m.G.backward(None, version=self.flatG.values_version())

    def _fork(self, state: _ShapeState):
        for s in state.streams[1:]:
            if s is not None:
                s.wait_stream(torch.cuda.current_stream(s.device))

    def _join(self, state: _ShapeState, flat: FlatParams, extras):
        """Wait for the side streams, then add their gradient parts into the network's flat gradient (fixed order)."""
        for s in state.streams[1:]:
            if s is not None:
                torch.cuda.current_stream(s.device).wait_stream(s)
        st = self.G.ctx.stream()
        for buf in extras:
            L.call("nirgan_axpy", flat.grad.data_ptr(), buf.data_ptr(), flat.total, 1.0, st)

    def _reduce(self, state: _ShapeState, flat: FlatParams, head):
        """Average the network's gradient over the ranks before its Adam step (DDP's all-reduce during backward)."""
        if self.reducer is None:
            return
        if state.bucketed:
            for part in head:                 # the tail bucket has been in flight since the middle of the backward plan
                self.reducer.begin(part)
            self.reducer.finish()
        else:
            self.reducer.all_reduce_mean(flat.grad)

    # ------------------------------------------------------------------ one batch
    def step(self, rgb: torch.Tensor, nir: torch.Tensor, embeds: Optional[torch.Tensor] = None) -> "LossView":
        B, _, H, W = rgb.shape
        self._prepare(B, H, W)
        state = self._state
        st = self.G.ctx.stream()
        for m in state.micros:
            m.rgb.copy_(rgb[m.lo:m.hi])
            m.nir.copy_(nir[m.lo:m.hi])
        L.check(L.backend().nirgan_fill(self.losses.data_ptr(), 16, 0.0, st), "fill")
        # ---- optimizer 0: generator forward + discriminator on [fake ; real], per micro-batch
        self._fork(state)
        for m, s in zip(state.micros, state.streams):
            with _on_stream(s):
                self._d_pass(m, None if embeds is None else embeds[m.lo:m.hi])
        self._join(state, self.flatD, state.extraD)
        self._reduce(state, self.flatD, state.headD)
        self.flatD.adam_step(self.lr if self.lr_d is None else self.lr_d, self.beta1, stream=st)
        # ---- optimizer 1: generator against the updated, frozen discriminator
        self._fork(state)
        for m, s in zip(state.micros, state.streams):
            with _on_stream(s):
                self._g_pass(m)
        self._join(state, self.flatG, state.extraG)
        self._reduce(state, self.flatG, state.headG)
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
        if tr.lambda_ssim > 0.0:
            out["loss_G_ssim"] = v[10] / tr.lambda_ssim
            loss_g += v[10]
        out["loss_G"] = loss_g
        return out
