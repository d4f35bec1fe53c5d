"""torch.autograd bridges: the HIP engines behind ordinary differentiable calls.

The reference's callers (model/pix2pix.py:165-257 under Lightning, create_synthetic_dataset.py:107)
treat netG / netD / the losses as autograd-differentiable modules.  These Functions keep that
contract: forward runs an engine's forward plan, backward its backward plan; parameter
gradients are returned to autograd so ``loss.backward()`` + any optimizer works unchanged.
Engines are leased per call (a second forward before the first backward -- D(fake) and
D(real) in one graph -- gets its own buffer set) and returned when the graph node dies.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from .flat import FlatParams
from .nets import DiscriminatorEngine, GeneratorEngine


def _require_device(t: torch.Tensor, what: str):
    if t.device.type != "cuda" and not L.is_emulated():
        raise RuntimeError(f"{what}: tensors must live on the MI355X (cuda); the HIP path has no CPU fallback")


class _Lease:
    def __init__(self, pool: "EnginePool", key, eng):
        self.pool, self.key, self.eng = pool, key, eng
        self.bound = False          # an autograd node owns this lease: its backward runs the engine's backward plan and releases it

    def release(self):
        if self.eng is not None:
            self.pool.free.setdefault(self.key, []).append(self.eng)
            self.eng = None

    def __del__(self):
        self.release()


class EnginePool:
    """Engines of one network keyed by shape; rebuilt when the flat parameter storage moves."""

    def __init__(self, flat: FlatParams, factory):
        self.flat, self.factory = flat, factory
        self.free: Dict[tuple, list] = {}
        self._bound_ptr = None

    def lease(self, key) -> _Lease:
        self.flat.ensure()
        ptr = self.flat.flat.data_ptr()
        if ptr != self._bound_ptr:
            self.free.clear()
            self._bound_ptr = ptr
        lst = self.free.get(key, [])
        eng = lst.pop() if lst else self.factory(key)
        return _Lease(self, key, eng)


class _ForwardRecord:
    """What the generator's last training-mode forward ran on: the input tensor OBJECTS (held, so their storage cannot be handed to
    another tensor) with torch's version counters, the parameter version, and the engine lease that still holds every activation.
    The reference runs the generator forward in BOTH optimizer passes of a batch with unchanged parameters (model/pix2pix.py:
    178-180), which reproduces ``pred`` bit for bit; a second forward on the same tensors with the same parameters re-uses the
    first one's activations instead (and gets a backward-capable graph even when the first pass ran with the generator frozen,
    as Lightning's toggle_optimizer does).  Dropped by the next forward, by a backward through the lease, by any change.
    A lease is handed to at most ONE autograd node: once a forward has bound it (its graph will run the engine's backward and return
    the engine to the pool), a further forward on the same tensors runs on a fresh engine -- two backward-capable calls G(x), G(x) give two
    independent graphs, as with the reference's nn.Module.  Changes are detected through torch's version counters, the same evidence
    autograd itself uses for its saved tensors: writes that bypass them (``x.data.copy_``, numpy views of the storage) are not seen."""

    def __init__(self, rgb, embeds, ver, key, lease):
        self.rgb, self.rgb_v = rgb, rgb._version
        self.embeds, self.emb_v = embeds, (None if embeds is None else embeds._version)
        self.ver, self.key, self.lease = ver, key, lease

    def matches(self, rgb, embeds, ver, key) -> bool:
        return (self.lease.eng is not None and not self.lease.bound and rgb is self.rgb and rgb._version == self.rgb_v and embeds is self.embeds
                and (embeds is None or embeds._version == self.emb_v) and ver == self.ver and key == self.key)


class GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, grad_mode, rgb, embeds, *params):
        _require_device(rgb, "generator")
        B, _, H, W = rgb.shape
        wants = grad_mode and any(ctx.needs_input_grad[4:])
        # training mode with autograd on: run on the backward-capable engine and remember the forward, so that the second
        # optimizer pass of the batch does not repeat it; inference (no_grad / eval): a forward-only engine without backward buffers
        need_bwd = wants or (grad_mode and net.training)
        ver = net._flat().values_version()
        key = (B, H, W, net.data_pad, need_bwd)
        rec = net.__dict__.get("_fwd_record")
        net.__dict__["_fwd_record"] = None
        if rec is not None and need_bwd and rec.matches(rgb, embeds, ver, key):
            lease = rec.lease                                           # same tensors, same parameters: the activations are there
            net.__dict__["_fwd_reused"] = net.__dict__.get("_fwd_reused", 0) + 1
        else:
            rec = None
            lease = net._pool().lease(key)
            lease.eng.forward(rgb.detach().contiguous().float(), None if embeds is None else embeds.detach().contiguous().float(), version=ver)
        out = lease.eng.pred.clone()
        if need_bwd:
            net.__dict__["_fwd_record"] = rec if rec is not None else _ForwardRecord(rgb, embeds, ver, key, lease)
        if wants:
            lease.bound = True
            ctx.lease, ctx.net, ctx.ver, ctx.has_emb = lease, net, ver, embeds is not None
        return out

    @staticmethod
    def backward(ctx, dpred):
        lease, net = ctx.lease, ctx.net
        rec = net.__dict__.get("_fwd_record")
        if rec is not None and rec.lease is lease:
            net.__dict__["_fwd_record"] = None                          # the engine goes back to the pool below
        eng = lease.eng
        if eng is None:
            raise RuntimeError("generator backward: this graph's activations were already consumed by an earlier backward "
                               "(the engine went back to the pool); call backward once per forward, as with retain_graph=False")
        eng.backward(dpred.contiguous(), version=ctx.ver)
        flat = net._flat()
        # ONE buffer in the flat layout; the per-parameter gradients are views of it.  autograd keeps them as the .grad tensors
        # (first accumulation) and HipAdam.step recognises the layout and reads the buffer in place (optim.py)
        g = flat.grad.clone()
        grads = tuple(g[o:o + k].view(s) for (o, k, s) in (flat.slices[n] for n in flat.names))
        lease.release()
        return (None, None, None, None) + grads


class DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, grad_mode, x, *params):
        _require_device(x, "discriminator")
        B, _, H, W = x.shape
        need_w = grad_mode and any(ctx.needs_input_grad[3:])
        need_x = grad_mode and ctx.needs_input_grad[2]
        lease = net._pool().lease((B, H, W, need_w or need_x))
        eng = lease.eng
        ver = net._flat().values_version()
        out = eng.forward(x.detach().contiguous().float(), version=ver).clone()
        if need_w or need_x:
            ctx.lease, ctx.net, ctx.ver, ctx.need_w, ctx.need_x = lease, net, ver, need_w, need_x
        else:
            lease.release()
        return out

    @staticmethod
    def backward(ctx, dout):
        lease, net = ctx.lease, ctx.net
        eng = lease.eng
        if eng is None:
            raise RuntimeError("discriminator backward: this graph's activations were already consumed by an earlier backward")
        dout = dout.contiguous()
        flat = net._flat()
        gx = None
        grads = tuple(None for _ in flat.names)
        if ctx.need_w:
            eng.backward(dout, frozen=False, version=ctx.ver)
            g = flat.grad.clone()
            grads = tuple(g[o:o + k].view(s) for (o, k, s) in (flat.slices[n] for n in flat.names))
        if ctx.need_x:
            gx4 = eng.backward(dout, frozen=True, version=ctx.ver)
            gx = gx4.permute(0, 3, 1, 2).contiguous()
        lease.release()
        return (None, None, gx) + grads


class LsganFn(torch.autograd.Function):
    """GANLoss('lsgan'): MSELoss(prediction, label.expand_as(prediction)) (model/networks.py:233,258-276)."""

    @staticmethod
    def forward(ctx, pred, target: float):
        _require_device(pred, "GANLoss")
        p = pred.detach().contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        st = torch.cuda.current_stream(p.device).cuda_stream if p.device.type == "cuda" else None
        L.call("nirgan_lsgan", p.data_ptr(), p.numel(), float(target), 1.0, loss.data_ptr(), grad.data_ptr(), st)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad * gout, None


class PixLossFn(torch.autograd.Function):
    """weights[0]*L1(pred, nir) + sum_i weights[i]*crit(index_i(nir), index_i(pred)); differentiable in pred."""

    @staticmethod
    def forward(ctx, rgb, nir, pred, weights: Tuple[float, ...], criterion: int):
        _require_device(pred, "pixel losses")
        B, _, H, W = pred.shape
        nir_, pred_ = (t.detach().contiguous().float() for t in (nir, pred))
        rgb_ = None if rgb is None else rgb.detach().contiguous().float()      # None: plain L1 (no index term)
        sums = torch.zeros(8, dtype=torch.float32, device=pred_.device)
        grad = torch.empty_like(pred_)
        d = L.PixLossDesc()
        d.rgb = None if rgb_ is None else rgb_.data_ptr()
        d.nir, d.pred, d.B, d.H, d.W = nir_.data_ptr(), pred_.data_ptr(), B, H, W
        (d.w_l1, d.w_ndvi, d.w_ndwi, d.w_gndvi, d.w_savi, d.w_msavi, d.w_evi) = [float(w) for w in weights]
        d.criterion, d.log_all = criterion, 0
        d.sums, d.grad_pred = sums.data_ptr(), grad.data_ptr()
        ws = torch.empty(L.PIX_LOSS_WS_ELEMS, dtype=torch.float32, device=pred_.device)      # per call: block partial sums (fixed-order finish)
        d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
        st = torch.cuda.current_stream(pred_.device).cuda_stream if pred_.device.type == "cuda" else None
        L.call("nirgan_pix_loss", C.byref(d), st)
        ctx.save_for_backward(grad)
        w = torch.tensor(list(weights) + [0.0], dtype=torch.float32, device=pred_.device)
        return (sums * w).sum() / float(B * H * W)

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return None, None, grad * gout, None, None


def index_sums(rgb, nir, pred, criterion: int) -> torch.Tensor:
    """All seven per-pixel criterion sums (L1 + six indices), forward only: the 'logging_dict' mode."""
    _require_device(pred, "pixel losses")
    B, _, H, W = pred.shape
    rgb_, nir_, pred_ = (t.detach().contiguous().float() for t in (rgb, nir, pred))
    sums = torch.zeros(8, dtype=torch.float32, device=pred_.device)
    d = L.PixLossDesc()
    d.rgb, d.nir, d.pred, d.B, d.H, d.W = rgb_.data_ptr(), nir_.data_ptr(), pred_.data_ptr(), B, H, W
    d.criterion, d.log_all = criterion, 1
    d.sums = sums.data_ptr()
    ws = torch.empty(L.PIX_LOSS_WS_ELEMS, dtype=torch.float32, device=pred_.device)
    d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    st = torch.cuda.current_stream(pred_.device).cuda_stream if pred_.device.type == "cuda" else None
    L.call("nirgan_pix_loss", C.byref(d), st)
    return sums[:7] / float(B * H * W)
