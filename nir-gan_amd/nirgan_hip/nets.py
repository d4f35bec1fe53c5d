"""Network-level engines: the ResNet generator (plain and SatCLIP-inject) and the PatchGAN.

Each engine is bound to one parameter set (device tensors in the reference layouts, keyed by
the reference's state_dict names) and one shape.  ``forward`` consumes / produces NCHW tensors
at the boundary exactly like the reference modules (model/networks.py:372-374, :582-584;
model/generator_inject.py:105-135); ``backward`` takes the gradient wrt the output and fills
the parameter gradients (and, for the PatchGAN, the gradient wrt its input).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import geometry as G
from . import lib as L
from .options import OPT
from .engine import (ConvIN, Ctx, Halo, Plan, SlabPool, TapPlaneConv, Weights, _Scratch, drop_dead_fp32_stores, emit_conv, grad_halo,
                     emit_deferred_reduce_rows, emit_w6_deferred_finishes, emit_wgrad, wino_applicable)


def generator_layout(n_blocks: int) -> dict:
    """Indices of the parametrised modules in ResnetGenerator.model (model/networks.py:341-370)."""
    up0 = 10 + n_blocks
    return {"first": 1, "down": [4, 7], "blocks": list(range(10, 10 + n_blocks)), "up": [up0, up0 + 3], "last": up0 + 7}


class _Engine:
    def __init__(self, device, precision="fp32"):
        self.ctx = Ctx(device, precision)
        self.weights = Weights(self.ctx)
        self.slabs = SlabPool(self.ctx)
        self.scratch = _Scratch(self.ctx)
        self.pack_fwd, self.pack_bwd = Plan(self.ctx), Plan(self.ctx)
        self.fwd, self.bwd = Plan(self.ctx), Plan(self.ctx)
        self._packed_version = -1
        self._packed_bwd_version = -1

    def refresh_weights(self, version: int, backward: bool = False):
        """Re-pack weights when the parameters changed (version = optimizer step counter)."""
        if not getattr(self, "_packs_fused", False):
            for pl in (self.pack_fwd, self.pack_bwd):
                pl.fuse_packs()
                pl.fuse_wino6_weights()
            self._packs_fused = True
        if self._packed_version != version:
            self.pack_fwd.run()
            self._packed_version = version
        if backward and self._packed_bwd_version != version:
            self.pack_bwd.run()
            self._packed_bwd_version = version


class GeneratorEngine(_Engine):
    """ResnetGenerator(input_nc=3, output_nc=1, ngf, n_blocks, InstanceNorm, reflect padding)."""

    def __init__(self, params: Dict[str, torch.Tensor], grads: Optional[Dict[str, torch.Tensor]], n_blocks: int,
                 B: int, H: int, W: int, data_pad: int = 0, inject: Optional[dict] = None, need_backward: bool = True,
                 precision="fp32"):
        dev = params["model.1.weight"].device
        super().__init__(dev, precision)
        self.need_backward = need_backward
        ctx = self.ctx
        self.params, self.grads = params, grads
        self.n_blocks, self.B, self.H, self.W, self.data_pad = n_blocks, B, H, W, data_pad
        self.inject = inject
        lay = generator_layout(n_blocks)
        ngf = params["model.1.weight"].shape[0]
        in_nc = params["model.1.weight"].shape[1]
        assert in_nc <= 4 and params[f"model.{lay['last']}.weight"].shape[0] == 1, "path is built for RGB(+1) -> NIR"
        Hg, Wg = H + 2 * data_pad, W + 2 * data_pad
        if Hg % 4 or Wg % 4:
            raise ValueError(f"generator input {Hg}x{Wg} (tile + 2*padding) must be a multiple of 4")
        self.Hg, self.Wg, self.in_nc = Hg, Wg, in_nc

        def P(i, what):
            return params[f"model.{i}.{what}"]

        # ---------------- buffers + layers
        self.rgb_in = ctx.zeros(B, in_nc, H, W)
        self.x4 = Halo(ctx, B, Hg, Wg, 4, 3)
        self.L1 = ConvIN(self, "first", "rowpacked", self.x4, P(1, "weight"), P(1, "bias"), k=7, s=1, p=3, cout=ngf,
                         out_pad=1, cin_real=in_nc)
        if inject is None:
            self.L2 = ConvIN(self, "down0", "conv", self.L1.out, P(4, "weight"), P(4, "bias"), k=3, s=2, p=1, cout=2 * ngf, out_pad=1)
            l2_out = self.L2.out
        else:
            self.L2 = ConvIN(self, "down0", "conv", self.L1.out, P(4, "weight"), P(4, "bias"), k=3, s=2, p=1, cout=2 * ngf,
                             out_pad=0, keep_z=True)
            H2, W2 = self.L2.OH, self.L2.OW
            if H2 != W2:
                raise ValueError("SatCLIP injection needs square tiles (generator_inject.py:116 swaps H and W)")
            self.a2 = Halo(ctx, B, H2, W2, 2 * ngf, 1)
            self.emb_in = Halo(ctx, B, 1, 1, 256, 0)
            self.e128 = Halo(ctx, B, 1, 1, 128 * 128, 0)
            self.e_map = ctx.zeros(B, H2, W2)
            l2_out = self.a2
        last_border = L.BORDER_REFLECT if n_blocks > 0 else L.BORDER_KEEP
        self.L3 = ConvIN(self, "down1", "conv", l2_out, P(7, "weight"), P(7, "bias"), k=3, s=2, p=1, cout=4 * ngf,
                         out_pad=1, out_border=last_border)
        self.blocks = []
        u = self.L3.out
        for j, i in enumerate(lay["blocks"]):
            c1 = ConvIN(self, f"b{j}c1", "conv", u, P(i, "conv_block.1.weight"), P(i, "conv_block.1.bias"), k=3, s=1, p=1,
                        cout=4 * ngf, out_pad=1, out_border=L.BORDER_REFLECT)
            c2 = ConvIN(self, f"b{j}c2", "conv", c1.out, P(i, "conv_block.5.weight"), P(i, "conv_block.5.bias"), k=3, s=1, p=1,
                        cout=4 * ngf, act=L.ACT_NONE, residual=u, out_pad=1,
                        out_border=L.BORDER_REFLECT if j < n_blocks - 1 else L.BORDER_KEEP)
            # c1's normalised output has ONE reader, c2's Winograd input transform: fold InstanceNorm + ReLU + reflect pad into that
            # transform and never write the buffer.  Needs c2's V kept for its weight gradient (else the backward re-reads c1.out).
            if wino_applicable(self.ctx, c2.inp, 3, 1, 1, c2.cout, c2.OH, c2.OW) and OPT.fold_apply:
                c1.defer_apply = True
                c2.producer = c1
            self.blocks.append((i, c1, c2))
            u = c2.out
        i0, i1 = lay["up"]
        self.U1 = ConvIN(self, "up0", "convT", u, P(i0, "weight"), P(i0, "bias"), k=3, s=2, p=1, cout=2 * ngf, out_pad=1)
        self.U2 = ConvIN(self, "up1", "convT", self.U1.out, P(i1, "weight"), P(i1, "bias"), k=3, s=2, p=1, cout=ngf,
                         out_pad=3, out_border=L.BORDER_REFLECT)
        il = lay["last"]
        self.last = TapPlaneConv(self, "last", self.U2.out, P(il, "weight"), P(il, "bias"), k=7, p=3, act=L.ACT_TANH, crop=data_pad)
        self.pred = self.last.dst
        # generator_inject.py:97-100,133-134: the prediction times a learnable scalar behind the tanh
        self.post_correction = inject is not None and bool(inject.get("post_correction"))
        if self.post_correction:
            self.pred_raw = self.pred
            self.pred = torch.empty_like(self.pred_raw)
        self.lay = lay

        # ---------------- forward plan
        f, pk = self.fwd, self.pack_fwd
        f.add("nirgan_nchw_to_halo", self.rgb_in.data_ptr(), B, in_nc, H, W, self.x4.ptr, 4, 0, data_pad, 3, L.BORDER_REFLECT)
        self.L1.emit_fwd(f, pk)
        self.L2.emit_fwd(f, pk)
        if inject is not None:
            self._emit_inject_fwd(f)
        self.L3.emit_fwd(f, pk)
        for _, c1, c2 in self.blocks:
            c1.emit_fwd(f, pk)
            c2.emit_fwd(f, pk)
        self.U1.emit_fwd(f, pk)
        self.U2.emit_fwd(f, pk)
        self.last.emit_fwd(f, pk)
        if self.post_correction:
            f.add("nirgan_param_scale_fwd", self.pred_raw.data_ptr(), self.params["post_correction_param"].data_ptr(), self.pred.data_ptr(), self.pred.numel())
        if need_backward:
            self._build_backward()
        # bf16 operand mode: activations / output gradients that every reader takes from the bf16 twin are stored as bf16 only
        drop_dead_fp32_stores(getattr(self, "twinned", []))

    # ------------------------------------------------------------------ SatCLIP injection
    def _emit_inject_fwd(self, f: Plan):
        ctx, B = self.ctx, self.B
        p = self.params
        H2, W2 = self.L2.OH, self.L2.OW
        # e = fc(embeds): a 1x1 product with M = B rows; fc.weight is already [N][K]
        emit_conv(f, ctx, self.emb_in, G.Taps([0], [0], 256), p["fc.weight"], p["fc.bias"], self.e128, N=128 * 128, OH=1, OW=1)
        # interpolate(size=(W, H)) as the reference does (generator_inject.py:116)
        f.add("nirgan_bilinear_fwd", self.e128.ptr, B, 128, 128, self.e_map.data_ptr(), W2, H2)
        d = L.InjectFwdDesc()
        d.z, d.e = self.L2.out.ptr, self.e_map.data_ptr()
        use_scale = self.inject.get("use_scale", True)
        d.scale = p["scale_param"].data_ptr() if use_scale else None
        d.style = 0 if self.inject.get("style", "multiply") == "multiply" else 1
        d.B, d.H, d.W, d.C = B, H2, W2, self.L2.cout
        d.out, d.o_hp, d.o_wp, d.o_pad = self.a2.ptr, self.a2.hp, self.a2.wp, 1
        ctx.keep.append(d)
        f.add("nirgan_inject_fwd", C.byref(d))

    def _emit_inject_bwd(self, b: Plan, g_a2: Halo):
        ctx, B, p, gr = self.ctx, self.B, self.params, self.grads
        H2, W2, C2 = self.L2.OH, self.L2.OW, self.L2.cout
        self.dz2 = Halo(ctx, B, H2, W2, C2, 0)
        self.de_map = ctx.zeros(B, H2, W2)
        self.de128 = Halo(ctx, B, 1, 1, 128 * 128, 0)
        use_scale = self.inject.get("use_scale", True)
        d = L.InjectBwdDesc()
        d.g, d.a, d.a_hp, d.a_wp, d.a_pad = g_a2.ptr, self.a2.ptr, self.a2.hp, self.a2.wp, 1
        d.z, d.e = self.L2.out.ptr, self.e_map.data_ptr()
        d.scale = p["scale_param"].data_ptr() if use_scale else None
        d.style = 0 if self.inject.get("style", "multiply") == "multiply" else 1
        d.B, d.H, d.W, d.C = B, H2, W2, C2
        d.dz, d.de = self.dz2.ptr, self.de_map.data_ptr()
        d.dscale = gr["scale_param"].data_ptr() if use_scale else None
        self.inject_ws = ctx.zeros(2048)                   # per-block partial sums of dscale, added in block order
        d.ws, d.ws_elems = self.inject_ws.data_ptr(), self.inject_ws.numel()
        ctx.keep.append(d)
        if use_scale:
            b.add("nirgan_fill", gr["scale_param"].data_ptr(), 1, 0.0)
        b.add("nirgan_inject_bwd", C.byref(d))
        b.add("nirgan_bilinear_bwd", self.de_map.data_ptr(), B, W2, H2, self.de128.ptr, 128, 128)
        # fc: dW[n][k] = sum_b dE[b][n] * emb[b][k];  db = column sums of dE
        emit_wgrad(b, ctx, self.de128, self.emb_in, G.Taps([0], [0], 256), G.linear_pack(128 * 128, 256), gr["fc.weight"],
                   N=128 * 128, OH=1, OW=1, p_oh=0, p_ow=0, slabs_pool=self.slabs)
        b.add("nirgan_colsum", self.de128.ptr, B, 128 * 128, gr["fc.bias"].data_ptr(), 0)
        return self.dz2

    # ------------------------------------------------------------------ backward plan
    def _build_backward(self):
        ctx, gr, b, pk = self.ctx, self.grads, self.bwd, self.pack_bwd
        assert gr is not None
        B, lay = self.B, self.lay

        def GW(i, what="weight"):
            return gr[f"model.{i}.{what}"]

        ctx.rr_deferred = [] if OPT.batch_reduce else None      # the weight gradients' slab sums are collected and run as one launch per flush point
        self.last.alloc_bwd()
        self.dpred = self.last.dout
        if self.post_correction:
            # the loss writes the gradient wrt the CORRECTED prediction; the last layer's backward takes it times the parameter
            self.dpred = torch.zeros_like(self.last.dout)
            self.pc_ws = ctx.zeros(1024)
            gpc = gr["post_correction_param"]
            b.add("nirgan_fill", gpc.data_ptr(), 1, 0.0)
            b.add("nirgan_param_scale_bwd", self.dpred.data_ptr(), self.pred_raw.data_ptr(), self.params["post_correction_param"].data_ptr(),
                  self.last.dout.data_ptr(), gpc.data_ptr(), self.pc_ws.data_ptr(), self.pc_ws.numel(), self.dpred.numel())
        for layer in [self.L1]:
            layer.alloc_bwd(need_dgrad=False)
        for layer in [self.L2, self.L3, self.U1, self.U2] + [c for _, c1, c2 in self.blocks for c in (c1, c2)]:
            layer.alloc_bwd(need_dgrad=True)
        # the only live generator bias (last conv) is accumulated with atomics: zero it first.  Biases in front
        # of an InstanceNorm have gradient exactly 0 and are never written (their flat-gradient slots stay 0).
        il = lay["last"]
        b.add("nirgan_fill", GW(il, "bias").data_ptr(), 1, 0.0)
        self.last.emit_bwd(b, pk, GW(il), GW(il, "bias"))
        i0, i1 = lay["up"]
        g_u1 = grad_halo(ctx, B, self.U1.OH, self.U1.OW, self.U1.cout, 0)
        self.U2.emit_bwd(b, pk, g=self.last.gin, g_fold=True, gw=GW(i1), gb=GW(i1, "bias"), dgrad_out=g_u1)
        c4, H3, W3 = self.L3.cout, self.L3.OH, self.L3.OW
        g_top = Halo(ctx, B, H3, W3, c4, 0)                # gradient wrt the last block's output
        self.U1.emit_bwd(b, pk, g=g_u1, gw=GW(i0), gb=GW(i0, "bias"), dgrad_out=g_top)
        # residual chain: g_in (halo'd, to fold) + g_skip (dense) = gradient wrt u_{i+1}
        g_in, g_fold, g_skip = g_top, False, None
        dense = [Halo(ctx, B, H3, W3, c4, 0), Halo(ctx, B, H3, W3, c4, 0)] if self.blocks else []
        flip = 0
        ctx.w6_deferred = []      # collect the blocks' weight-gradient finishes
        for j in range(len(self.blocks) - 1, -1, -1):
            i, c1, c2 = self.blocks[j]
            gq = grad_halo(ctx, B, H3, W3, c4, 1)
            gp = grad_halo(ctx, B, H3, W3, c4, 1)
            if g_skip is None:
                gsum, skip_next = None, g_in           # top of the chain: g_in is already dense
            else:
                gsum, skip_next = dense[flip], dense[flip]
                flip ^= 1
            c2.emit_bwd(b, pk, g=g_in, g_fold=g_fold, g2=g_skip, gsum=gsum, gw=GW(i, "conv_block.5.weight"),
                        gb=GW(i, "conv_block.5.bias"), dgrad_out=gq, act=L.ACT_NONE)
            c1.emit_bwd(b, pk, g=gq, g_fold=True, gw=GW(i, "conv_block.1.weight"), gb=GW(i, "conv_block.1.bias"), dgrad_out=gp)
            g_in, g_fold, g_skip = gp, True, skip_next
        emit_w6_deferred_finishes(b, ctx)         # the blocks' Winograd weight gradients: one inverse-transform launch for all of them
        emit_deferred_reduce_rows(b, ctx, last=False)     # the slab sums of every weight gradient so far (last layer, decoder, blocks) as one launch
        # data parallel: from here on the gradients of [first residual block .. last conv] are final (28 of 31 MB for 6 blocks)
        first_tail = f"model.{lay['blocks'][0]}.conv_block.1.weight" if self.blocks else f"model.{i0}.weight"
        self.bwd_tail = (len(b.ops), first_tail, f"model.{il}.bias")
        # (read by the SatCLIP modulation's backward as fp32 when there is one)
        g_a2 = grad_halo(ctx, B, self.L2.OH, self.L2.OW, self.L2.cout, 0, phases=4) if self.inject is None else Halo(ctx, B, self.L2.OH, self.L2.OW, self.L2.cout, 0)
        self.L3.emit_bwd(b, pk, g=g_in, g_fold=g_fold, g2=g_skip, gw=GW(7), gb=GW(7, "bias"), dgrad_out=g_a2)
        g_a1 = grad_halo(ctx, B, self.L1.OH, self.L1.OW, self.L1.cout, 0, phases=4)
        if self.inject is None:
            self.L2.emit_bwd(b, pk, g=g_a2, gw=GW(4), gb=GW(4, "bias"), dgrad_out=g_a1)
        else:
            dz2 = self._emit_inject_bwd(b, g_a2)
            self.L2.emit_bwd(b, pk, g=dz2, gw=GW(4), gb=GW(4, "bias"), dgrad_out=g_a1, act=L.ACT_NONE)
        # data parallel: everything but the FIRST layer's gradient is final here (the two stride-2 layers, the SatCLIP projection): the
        # middle bucket goes out under the first layer's backward; only model.1.* (38 KB) is left for after the plan
        emit_deferred_reduce_rows(b, ctx, last=False)
        self.bwd_mid = (len(b.ops), "model.1.weight", "model.1.bias")
        self.L1.emit_bwd(b, pk, g=g_a1, gw=GW(1), gb=GW(1, "bias"), dgrad_out=None)
        emit_deferred_reduce_rows(b, ctx)         # the first layer's slab sum (the data-parallel head bucket)

    # ------------------------------------------------------------------ run
    def forward(self, rgb: torch.Tensor, embeds: Optional[torch.Tensor] = None, version: int = 0) -> torch.Tensor:
        self.refresh_weights(version)
        self.rgb_in.copy_(rgb)
        if self.inject is not None:
            self.emb_in.t.view(self.B, 256).copy_(embeds)
        self.fwd.run()
        return self.pred

    def backward(self, dpred: Optional[torch.Tensor], version: int = 0) -> None:
        """dpred: gradient wrt the (cropped) prediction; None when the caller wrote self.dpred in place."""
        self.refresh_weights(version, backward=True)
        if dpred is not None:
            self.dpred.copy_(dpred)
        self.bwd.run()


class DiscriminatorEngine(_Engine):
    """NLayerDiscriminator(input_nc=4, ndf, n_layers=3, InstanceNorm): 70x70 PatchGAN."""

    def __init__(self, params: Dict[str, torch.Tensor], grads: Optional[Dict[str, torch.Tensor]], B: int, H: int, W: int,
                 need_backward: bool = True, precision="fp32"):
        dev = params["model.0.weight"].device
        super().__init__(dev, precision)
        ctx = self.ctx
        self.params, self.grads, self.B, self.H, self.W = params, grads, B, H, W
        ndf, in_nc = params["model.0.weight"].shape[0], params["model.0.weight"].shape[1]
        assert in_nc == 4, "PatchGAN input is cat(rgb, nir) = 4 channels on this path"

        def P(i, what):
            return params[f"model.{i}.{what}"]

        self.x_in = ctx.zeros(B, 4, H, W)
        self.x4 = Halo(ctx, B, H, W, 4, 1)
        self.C1 = ConvIN(self, "c1", "rowpacked", self.x4, P(0, "weight"), P(0, "bias"), k=4, s=2, p=1, cout=ndf, norm=False,
                         act=L.ACT_LRELU, out_pad=1, cin_real=4)
        self.C2 = ConvIN(self, "c2", "conv", self.C1.out, P(2, "weight"), P(2, "bias"), k=4, s=2, p=1, cout=2 * ndf, act=L.ACT_LRELU, out_pad=1)
        self.C3 = ConvIN(self, "c3", "conv", self.C2.out, P(5, "weight"), P(5, "bias"), k=4, s=2, p=1, cout=4 * ndf, act=L.ACT_LRELU, out_pad=1)
        self.C4 = ConvIN(self, "c4", "conv", self.C3.out, P(8, "weight"), P(8, "bias"), k=4, s=1, p=1, cout=8 * ndf, act=L.ACT_LRELU, out_pad=1)
        self.C5 = TapPlaneConv(self, "c5", self.C4.out, P(11, "weight"), P(11, "bias"), k=4, p=1)
        self.out = self.C5.dst
        f, pk = self.fwd, self.pack_fwd
        self.in_plan = Plan(ctx)
        self.in_plan.add("nirgan_nchw_to_halo", self.x_in.data_ptr(), B, 4, H, W, self.x4.ptr, 4, 0, 0, 1, L.BORDER_KEEP)
        for layer in (self.C1, self.C2, self.C3, self.C4, self.C5):
            layer.emit_fwd(f, pk)
        self._part_plans: dict = {}
        if need_backward:
            self._build_backward()
        drop_dead_fp32_stores(getattr(self, "twinned", []))          # bf16 operand mode: see GeneratorEngine

    def _build_backward(self):
        ctx, b, pk, gr, B = self.ctx, self.bwd, self.pack_bwd, self.grads, self.B
        self.C5.alloc_bwd()
        self.dout = self.C5.dout
        for layer in (self.C1, self.C2, self.C3, self.C4):
            layer.alloc_bwd(need_dgrad=True)
        self.g3p = grad_halo(ctx, B, self.C3.OH, self.C3.OW, self.C3.cout, 1)
        self.g2 = grad_halo(ctx, B, self.C2.OH, self.C2.OW, self.C2.cout, 0, phases=4)
        self.g1 = grad_halo(ctx, B, self.C1.OH, self.C1.OW, self.C1.cout, 0, phases=4)
        self.gx4 = Halo(ctx, B, self.H, self.W, 4, 0)
        self.gpred = ctx.zeros(B, self.H, self.W)
        self.bwd_frozen = Plan(ctx)       # parameters frozen: gradient wrt the whole 4-channel input (autograd bridge)
        self.bwd_pred = Plan(ctx)         # parameters frozen: gradient wrt channel 3 (pred) only (fused generator step)
        for plan, mode in ((b, "train"), (self.bwd_frozen, "input"), (self.bwd_pred, "pred")):
            frozen = mode != "train"

            def GW(i, what="weight"):
                return None if frozen else gr[f"model.{i}.{what}"]
            if not frozen:   # live biases (first and last conv) accumulate with atomics
                for k in ("model.0.bias", "model.11.bias"):
                    plan.add("nirgan_fill", gr[k].data_ptr(), gr[k].numel(), 0.0)
                ctx.rr_deferred = [] if OPT.batch_reduce else None      # the weight gradients' slab sums: one launch per flush point
            self.C5.emit_bwd(plan, pk, GW(11), GW(11, "bias"))
            self.C4.emit_bwd(plan, pk, g=self.C5.gin, g_fold=False, gw=GW(8), gb=GW(8, "bias"), dgrad_out=self.g3p)
            if not frozen:   # data parallel: the two last layers' gradients (8.4 of 11 MB) are final here
                emit_deferred_reduce_rows(plan, ctx, last=False)
                self.bwd_tail = (len(plan.ops), "model.8.weight", "model.11.bias")
            self.C3.emit_bwd(plan, pk, g=self.g3p, g_fold=False, gw=GW(5), gb=GW(5, "bias"), dgrad_out=self.g2)
            self.C2.emit_bwd(plan, pk, g=self.g2, gw=GW(2), gb=GW(2, "bias"), dgrad_out=self.g1)
            if not frozen:   # data parallel: all but the first layer's gradient (16 KB) is final here -- the middle bucket
                emit_deferred_reduce_rows(plan, ctx, last=False)
                self.bwd_mid = (len(plan.ops), "model.0.weight", "model.0.bias")
            self.C1.emit_bwd(plan, pk, g=self.g1, gw=GW(0), gb=GW(0, "bias"), dgrad_out=self.gx4 if mode == "input" else None)
            if not frozen:
                emit_deferred_reduce_rows(plan, ctx)
            if mode == "pred":
                d = L.ChanDgradDesc()
                dy, w = self.C1.dy, self.params["model.0.weight"]
                d.dy, d.dy_hp, d.dy_wp, d.dy_pad, d.C = dy.ptr, dy.hp, dy.wp, dy.pad, dy.C
                d.w, d.cin, d.k, d.stride, d.pad, d.channel = w.data_ptr(), 4, 4, 2, 1, 3
                d.B, d.H, d.W, d.out = B, self.H, self.W, self.gpred.data_ptr()
                ctx.keep.append(d)
                plan.add("nirgan_conv_channel_dgrad", C.byref(d))

    def input_plan(self, parts) -> Plan:
        """parts: list of (NCHW tensor [nb, Cs, H, W], b0, c0) written straight into the halo'd input
        (the torch.cat of model/pix2pix.py:197,202,216 never materialises)."""
        key = tuple((t.data_ptr(), t.shape[0], t.shape[1], b0, c0) for t, b0, c0 in parts)
        if key not in self._part_plans:
            pl = Plan(self.ctx)
            for t, b0, c0 in parts:
                nb, cs = t.shape[0], t.shape[1]
                assert t.is_contiguous() and t.shape[2] == self.H and t.shape[3] == self.W and b0 + nb <= self.B
                dst = self.x4.ptr + b0 * self.x4.hp * self.x4.wp * 4 * 4
                pl.add("nirgan_nchw_to_halo", t.data_ptr(), nb, cs, self.H, self.W, dst, 4, c0, 0, 1, L.BORDER_KEEP)
            self._part_plans[key] = pl
        return self._part_plans[key]

    def forward(self, x: Optional[torch.Tensor] = None, parts=None, version: int = 0) -> torch.Tensor:
        self.refresh_weights(version)
        if parts is not None:
            self.input_plan(parts).run()
        else:
            self.x_in.copy_(x)
            self.in_plan.run()
        self.fwd.run()
        return self.out

    def backward(self, dout: Optional[torch.Tensor], frozen: bool = False, version: int = 0, pred_only: bool = False) -> Optional[torch.Tensor]:
        """frozen=False: parameter gradients.  frozen=True: gradient wrt the input ([B][H][W][4]), or wrt its
        channel 3 only ([B][H][W]) when pred_only."""
        self.refresh_weights(version, backward=True)
        if dout is not None:
            self.dout.copy_(dout)
        if not frozen:
            self.bwd.run()
            return None
        if pred_only:
            self.bwd_pred.run()
            return self.gpred
        self.bwd_frozen.run()
        return self.gx4.t
