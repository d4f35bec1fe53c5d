"""MI355X-native counterpart of the reference's ``model/generator_inject.py``.

``define_G_inject(config)`` returns a generator with the reference's parameters and
``state_dict`` keys (``model.<idx>...``, ``fc.weight/bias``, ``scale_param``,
``post_correction_param``) whose ``forward(input, embeds)`` injects the SatCLIP embedding
after the first stride-2 conv + InstanceNorm and before its ReLU
(reference: model/generator_inject.py:105-135, factory :145-200).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from model.networks import Linear, _HipNet, _is_instance, _resnet_sequence, get_norm_layer, init_net
from nirgan_hip import functional as HF
from nirgan_hip.nets import GeneratorEngine


class ResnetGenerator_inject(_HipNet):
    def __init__(self, config, norm_layer, n_blocks=9):
        super().__init__()
        base, sat = config.base_configs, config.satclip
        self.inject_style = sat.satclip_inject_style
        self.post_correction = sat.post_correction
        self.post_correction_init = sat.post_correction_init
        self.scaling_param = sat.scaling_param
        self.scaling_param_init = sat.scaling_param_init
        assert (n_blocks >= 0)
        if not _is_instance(norm_layer):
            raise NotImplementedError('only norm="instance" runs on the MI355X path')
        if self.inject_style not in ("add", "multiply"):
            raise NotImplementedError(f"inject style '{self.inject_style}' not recognized: 'add' or 'multiply'")
        if self.inject_style == "add" and not self.scaling_param:
            raise AttributeError("inject style 'add' reads scale_param (generator_inject.py:123); enable scaling_param")
        self.n_blocks, self.data_pad = n_blocks, 0
        model = _resnet_sequence(base.input_nc, base.output_nc, base.ngf, norm_layer, not base.no_dropout, n_blocks, 'reflect')
        self.embed_fc_ou_square = 128
        self.fc = Linear(in_features=256, out_features=self.embed_fc_ou_square * self.embed_fc_ou_square)
        if self.scaling_param:
            print("Setting learned scale Parameter with init value: ", self.scaling_param_init)
            self.scale_param = nn.Parameter(torch.tensor(float(self.scaling_param_init)))
        if self.post_correction:
            print("Setting Post-Correction Parameter with init value: ", self.post_correction_init)
            self.post_correction_param = nn.Parameter(torch.tensor(float(self.post_correction_init)))
        self.model = nn.Sequential(*model)

    def _make_engine(self, key):
        B, H, W, pad, need_bwd = key
        f = self._flat()
        cfg = {"style": self.inject_style, "use_scale": bool(self.scaling_param), "post_correction": bool(self.post_correction)}
        return GeneratorEngine(f.param_views(), f.grad_views(), self.n_blocks, B, H, W, data_pad=pad, inject=cfg,
                               need_backward=need_bwd, precision=getattr(self, "precision", "fp32"))

    def forward(self, input, embeds):
        return HF.GeneratorFn.apply(self, torch.is_grad_enabled(), input, embeds, *self.parameters())


def define_G_inject(config):
    """generator_inject.py:145-200: only resnet_9blocks."""
    base = config.base_configs
    norm_layer = get_norm_layer(norm_type=base.norm)
    if base.netG == 'resnet_9blocks':
        net = ResnetGenerator_inject(config, norm_layer=norm_layer, n_blocks=9)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized. Only resnet_9blocks for SatCLIP.' % base.netG)
    return init_net(net, base.init_type, base.init_gain, [])
