"""torch.optim front end of the fused HIP Adam (so Lightning / ReduceLROnPlateau keep working).

``HipAdam(net.parameters(), lr, betas)`` has torch.optim.Adam's interface
(model/pix2pix.py:486-487); ``step()`` gathers the autograd gradients into the network's flat
gradient range and runs one nirgan_adam launch over the flat parameter range.

The moments and the step count live in the network's ``FlatParams`` (shared with the fused
``Pix2PixTrainer``).  ``state_dict()`` / ``load_state_dict()`` speak torch.optim.Adam's format
(per parameter ``step``, ``exp_avg``, ``exp_avg_sq``; the param-group keys torch writes), so the
``optimizer_states`` of a Lightning checkpoint (train.py:66-70,126 ``resume_from_checkpoint``)
load here and a checkpoint written here loads into ``torch.optim.Adam``.
"""
from __future__ import annotations

import torch

from .flat import FlatParams


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, net: torch.nn.Module = None):
        params = list(params)
        # the extra keys are what torch.optim.Adam keeps in its param groups: written so that its load_state_dict accepts ours
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None))
        if net is None:
            raise ValueError("HipAdam needs net= (the module whose parameters it owns)")
        if len(self.param_groups) != 1:
            raise ValueError("HipAdam runs one launch over the whole network: a single param group")
        self.net = net
        named = {id(p): n for n, p in net.named_parameters()}
        self._names = [named[id(p)] for p in self.param_groups[0]["params"]]
        if len(self._names) != len(named):
            raise ValueError("HipAdam must own every parameter of net (the fused kernel updates the flat range)")

    # ------------------------------------------------------------------ state <-> flat moments
    def _bind_state(self) -> None:
        """self.state[p] = views into the flat moments (nothing before the first step, like torch)."""
        flat: FlatParams = self.net._flat()
        self.state.clear()
        if flat.step_count == 0 or flat.m is None:
            return
        for n, p in zip(self._names, self.param_groups[0]["params"]):
            o, k, shp = flat.slices[n]
            self.state[p] = {"step": torch.tensor(float(flat.step_count)),
                             "exp_avg": flat.m[o:o + k].view(shp), "exp_avg_sq": flat.v[o:o + k].view(shp)}

    def state_dict(self):
        self._bind_state()
        sd = super().state_dict()
        # snapshots, not live views of the flat range
        sd["state"] = {i: {k: v.detach().clone() for k, v in st.items()} for i, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)       # casts to the parameters' device / dtype, restores lr & betas
        flat: FlatParams = self.net._flat()
        loaded = dict(self.state)
        if loaded:
            m, v = flat.moments()
            m.zero_()
            v.zero_()
            steps = set()
            for n, p in zip(self._names, self.param_groups[0]["params"]):
                st = loaded.get(p)
                if st is None:
                    continue
                o, k, _ = flat.slices[n]
                m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) != 1:
                raise ValueError(f"HipAdam keeps one step count per network; the checkpoint has {sorted(steps)}")
            flat.step_count = steps.pop()
        else:
            flat.step_count = 0
            if flat.m is not None:
                flat.m.zero_()
                flat.v.zero_()
        self._bind_state()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flat: FlatParams = self.net._flat()
        g = self.param_groups[0]
        st = torch.cuda.current_stream(flat.device).cuda_stream if flat.device.type == "cuda" else None
        params = self.param_groups[0]["params"]
        grads = [p.grad for p in params]
        if all(gr is None for gr in grads):
            return loss                                   # torch.optim.Adam: nothing to do, no step counted
        if all(gr is not None for gr in grads):
            # the autograd bridge (functional.GeneratorFn / DiscriminatorFn.backward) hands autograd views of ONE buffer in the flat
            # layout and autograd keeps them as the .grad tensors: Adam then reads that buffer in place -- no per-parameter copy
            base = self._aliased_flat_gradient(flat, grads)
            if base is not None:
                flat.adam_step(g["lr"], g["betas"][0], g["betas"][1], g["eps"], stream=st, grad_ptr=base)
                return loss
            views = flat.grad_views()
            for n, gr in zip(self._names, grads):
                views[n].copy_(gr)
            flat.adam_step(g["lr"], g["betas"][0], g["betas"][1], g["eps"], stream=st)
            return loss
        # some parameters have no gradient: torch.optim.Adam leaves those tensors (values AND moments) untouched.  The fused kernel
        # runs over the contiguous runs of tensors that do have one (the step count stays one per network).
        views = flat.grad_views()
        runs, cur = [], None
        for n, gr in zip(self._names, grads):
            o, k, _ = flat.slices[n]
            if gr is None:
                cur = None
                continue
            views[n].copy_(gr)
            end = o + -(-k // 4) * 4
            if cur is not None and cur[1] == o:
                cur[1] = end
            else:
                cur = [o, end]
                runs.append(cur)
        flat.adam_step(g["lr"], g["betas"][0], g["betas"][1], g["eps"], stream=st, ranges=[(a, min(b, flat.total)) for a, b in runs])
        return loss

    def _aliased_flat_gradient(self, flat: FlatParams, grads):
        """Address of a buffer that holds every gradient at its flat-layout offset (fp32, contiguous), or None."""
        base = None
        for n, gr in zip(self._names, grads):
            o, k, _ = flat.slices[n]
            if gr.dtype != torch.float32 or gr.device != flat.device or not gr.is_contiguous():
                return None
            b = gr.data_ptr() - 4 * o
            if base is None:
                base = b
            elif b != base:
                return None
        if base is None or base % 16:
            return None
        # the padding elements between tensors must be readable: the views come from one allocation of flat.total floats
        s0 = grads[0].untyped_storage()
        lo, hi = s0.data_ptr(), s0.data_ptr() + s0.nbytes()
        return base if (lo <= base and base + 4 * flat.total <= hi) else None
