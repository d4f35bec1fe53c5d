"""torch.optim front end of the fused HIP Adam (so Lightning / ReduceLROnPlateau keep working).

``HipAdam(net.parameters(), lr, betas)`` has torch.optim.Adam's interface
(model/pix2pix.py:486-487); ``step()`` gathers the autograd gradients into the network's flat
gradient range and runs one nirgan_adam launch over the flat parameter range.
"""
from __future__ import annotations

import torch

from .flat import FlatParams


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, net: torch.nn.Module = None):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if net is None:
            raise ValueError("HipAdam needs net= (the module whose parameters it owns)")
        self.net = net

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flat: FlatParams = self.net._flat()
        g = self.param_groups[0]
        views = flat.grad_views()
        have = False
        for n, p in self.net.named_parameters():
            if p.grad is None:
                views[n].zero_()
            else:
                views[n].copy_(p.grad)
                have = True
        if have:
            st = torch.cuda.current_stream(flat.device).cuda_stream if flat.device.type == "cuda" else None
            flat.adam_step(g["lr"], g["betas"][0], g["betas"][1], g["eps"], stream=st)
        return loss
