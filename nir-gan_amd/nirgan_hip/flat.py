"""Flat parameter / gradient / Adam-state storage for one network.

All parameters of a network live in ONE fp32 device buffer (each tensor starts on a 16-byte
boundary), gradients in a second one of the same layout.  The nn.Parameter objects of the
module alias slices of it, so ``state_dict()`` keeps the reference's keys and layouts
(model/networks.py Sequential indices), Adam is a single streaming kernel over the range
(nirgan_adam) and data parallelism is a single RCCL all-reduce per network per step.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import lib as L


class FlatParams:
    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.names: List[str] = []
        self.slices: Dict[str, Tuple[int, int, torch.Size]] = {}
        self.flat = self.grad = self.m = self.v = None
        self.step_count = 0
        self.version = 0            # bumped whenever parameter VALUES change through this object
        self._bind()

    def _bind(self):
        named = list(self.module.named_parameters())
        assert named, "module has no parameters"
        dev = named[0][1].device
        off = 0
        self.names, self.slices = [], {}
        for n, p in named:
            assert p.dtype == torch.float32 and p.device == dev
            self.names.append(n)
            self.slices[n] = (off, p.numel(), p.shape)
            off += -(-p.numel() // 4) * 4
        same_layout = getattr(self, "total", None) == off and self.m is not None
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        # Adam moments follow the parameters across a re-bind (``.to(device)`` after training started); step_count stays with them
        if same_layout:
            self.m, self.v = self.m.to(dev).clone(), self.v.to(dev).clone()
        else:
            self.m = self.v = None
            self.step_count = 0
        with torch.no_grad():
            for n, p in named:
                o, k, shp = self.slices[n]
                view = self.flat[o:o + k].view(shp)
                view.copy_(p.data)
                p.data = view
        self._ptr0 = named[0][1].data_ptr()
        self.device = dev
        self.version += 1

    def ensure(self) -> bool:
        """Re-bind when the module was moved / re-materialised (e.g. ``.to(device)``). True if rebuilt."""
        p0 = next(self.module.parameters())
        if p0.data_ptr() != self._ptr0 or p0.device != self.device:
            self._bind()
            return True
        return False

    def touch(self) -> None:
        """Parameter values were written behind the nn.Parameters' backs (broadcast, checkpoint copy into ``flat``): packed /
        Winograd-domain weight caches keyed on values_version() must rebuild."""
        self.version += 1

    def moments(self):
        """(m, v) of the fused Adam, allocated on first use."""
        if self.m is None:
            self.m = torch.zeros_like(self.flat)
            self.v = torch.zeros_like(self.flat)
        return self.m, self.v

    def param_views(self) -> Dict[str, torch.Tensor]:
        return {n: self.flat[o:o + k].view(s) for n, (o, k, s) in self.slices.items()}

    def grad_views(self) -> Dict[str, torch.Tensor]:
        return {n: self.grad[o:o + k].view(s) for n, (o, k, s) in self.slices.items()}

    def values_version(self) -> int:
        """Changes whenever any parameter was modified in place (torch version counters) or re-bound."""
        return self.version * 1000003 + sum(p._version for p in self.module.parameters())

    def adam_step(self, lr: float, beta1: float, beta2: float = 0.999, eps: float = 1e-8, stream=None, grad_ptr=None, ranges=None):
        """torch.optim.Adam(lr, betas=(beta1, 0.999)) on the whole network (model/pix2pix.py:486-487).
        grad_ptr: read the gradients from another buffer of the flat layout (the autograd bridge's) instead of ``self.grad``;
        ranges: [(lo, hi)] element ranges to update (tensors without a gradient are left alone); default the whole range."""
        self.moments()
        self.step_count += 1
        gp = self.grad.data_ptr() if grad_ptr is None else grad_ptr
        for lo, hi in (ranges if ranges is not None else [(0, self.total)]):
            if hi > lo:
                L.call("nirgan_adam", self.flat.data_ptr() + 4 * lo, gp + 4 * lo, self.m.data_ptr() + 4 * lo, self.v.data_ptr() + 4 * lo,
                       hi - lo, lr, beta1, beta2, eps, self.step_count, stream)
        self.version += 1
