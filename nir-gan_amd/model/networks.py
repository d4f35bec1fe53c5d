"""MI355X-native counterpart of the reference's ``model/networks.py`` (same import path and API).

``define_G`` / ``define_D`` / ``GANLoss`` / ``get_norm_layer`` / ``init_weights`` / ``init_net`` keep the
reference's signatures, error behaviour, ``state_dict`` keys (``model.<idx>...``) and parameter
layouts (reference: model/networks.py:18-36, 68-117, 120-204, 210-276, 316-434, 539-584), so
checkpoints and callers (train.py:48, create_synthetic_dataset.py:21-28) drop in.  The modules
are parameter containers: ``forward`` runs the hand-written HIP engines of ``nirgan_hip``
(there is no torch.nn compute and no CPU fallback).

Supported on this path: ``netG`` resnet_6blocks / resnet_9blocks, ``netD`` basic / n_layers(3),
``norm='instance'``, no dropout, ``gan_mode='lsgan'`` -- everything the shipped configs select
(configs/config_px2px.yaml:13-21).  Other names the reference knows raise NotImplementedError.
"""
from __future__ import annotations

import functools
import math

import torch
import torch.nn as nn
from torch.nn import init

from nirgan_hip import functional as HF
from nirgan_hip.flat import FlatParams
from nirgan_hip.nets import DiscriminatorEngine, GeneratorEngine


# --------------------------------------------------------------------------------------------
# parameter containers.  Their constructors draw from the global torch RNG exactly as the
# torch.nn layers of the reference do (kaiming_uniform_(a=sqrt(5)) on the weight, then the
# bias), so that ``torch.manual_seed(s); define_G(...)`` yields the reference's weights.
# --------------------------------------------------------------------------------------------
class _ConvBase(nn.Module):
    def __init__(self, weight_shape, bias_len, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*weight_shape))
        self.bias = nn.Parameter(torch.empty(bias_len)) if bias else None
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)

    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container of the MI355X path; call the enclosing network")


class Conv2d(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        super().__init__((out_channels, in_channels, kernel_size, kernel_size), out_channels, bias)


class ConvTranspose2d(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.output_padding = kernel_size, stride, padding, output_padding
        super().__init__((in_channels, out_channels, kernel_size, kernel_size), out_channels, bias)


class Linear(_ConvBase):
    def __init__(self, in_features, out_features, bias=True):
        self.in_features, self.out_features = in_features, out_features
        super().__init__((out_features, in_features), out_features, bias)


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type='instance'):
    """networks.py:18-36.  Only 'instance' is executable on the HIP path; the names are all accepted."""
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == 'none':
        def norm_layer(x):
            return Identity()
        return norm_layer
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def _is_instance(norm_layer) -> bool:
    f = norm_layer.func if type(norm_layer) == functools.partial else norm_layer
    return f == nn.InstanceNorm2d


def init_weights(net, init_type='normal', init_gain=0.02):
    """networks.py:68-99: N(0, gain) on every Conv*/Linear weight, bias 0 (class-name match, as there)."""
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, 'weight') and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:102-117.  gpu_ids moves the net to the first listed device; multi-GPU is one
    process per GPU with RCCL (nirgan_hip.parallel), not nn.DataParallel."""
    if len(gpu_ids) > 0:
        assert (torch.cuda.is_available())
        net.to(gpu_ids[0])
    init_weights(net, init_type, init_gain=init_gain)
    return net


class _HipNet(nn.Module):
    """Shared plumbing: flat parameter storage and engine pool, created lazily on the device."""

    def _flat(self) -> FlatParams:
        f = self.__dict__.get("_flat_obj")
        if f is None:
            f = FlatParams(self)
            self.__dict__["_flat_obj"] = f
        else:
            f.ensure()
        return f

    @property
    def precision(self) -> str:
        """Operand precision of the MFMA contractions: 'fp32' (default, the reference's arithmetic), 'bf16', 'bf16x3'
        (nirgan_hip/engine.py::precision_code).  Not part of the reference's API; assigning rebuilds the engines."""
        return self.__dict__.get("_precision", "fp32")

    @precision.setter
    def precision(self, value: str):
        from nirgan_hip.engine import precision_code
        precision_code(value)
        self.__dict__["_precision"] = value
        self.__dict__["_pool_obj"] = None

    def _pool(self) -> HF.EnginePool:
        p = self.__dict__.get("_pool_obj")
        if p is None:
            p = HF.EnginePool(self._flat(), self._make_engine)
            self.__dict__["_pool_obj"] = p
        return p


class ResnetBlock(nn.Module):
    """Parameter layout of the reference ResnetBlock (networks.py:377-434): conv_block.1 / conv_block.5."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        if padding_type != 'reflect':
            raise NotImplementedError('padding [%s] is not implemented on the MI355X path' % padding_type)
        if use_dropout:
            raise NotImplementedError('dropout is not on the MI355X path (no shipped config enables it)')
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), Conv2d(dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), Conv2d(dim, dim, kernel_size=3, padding=0, bias=use_bias), norm_layer(dim))


def _resnet_sequence(input_nc, output_nc, ngf, norm_layer, use_dropout, n_blocks, padding_type):
    """The module list of ResnetGenerator.__init__ (networks.py:341-370), same indices."""
    use_bias = _is_instance(norm_layer)
    model = [nn.ReflectionPad2d(3), Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias), norm_layer(ngf), nn.ReLU(True)]
    for i in range(2):
        mult = 2 ** i
        model += [Conv2d(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=1, bias=use_bias),
                  norm_layer(ngf * mult * 2), nn.ReLU(True)]
    for _ in range(n_blocks):
        model += [ResnetBlock(ngf * 4, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout, use_bias=use_bias)]
    for i in range(2):
        mult = 2 ** (2 - i)
        model += [ConvTranspose2d(ngf * mult, int(ngf * mult / 2), kernel_size=3, stride=2, padding=1, output_padding=1, bias=use_bias),
                  norm_layer(int(ngf * mult / 2)), nn.ReLU(True)]
    model += [nn.ReflectionPad2d(3), Conv2d(ngf, output_nc, kernel_size=7, padding=0), nn.Tanh()]
    return model


class ResnetGenerator(_HipNet):
    """ResNet encoder-decoder generator (networks.py:316-374) on the HIP engines."""

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False, n_blocks=6, padding_type='reflect'):
        assert (n_blocks >= 0)
        super().__init__()
        if not _is_instance(norm_layer):
            raise NotImplementedError('only norm="instance" runs on the MI355X path (the shipped configs use it)')
        if output_nc != 1 or input_nc > 4 or ngf % 4:
            raise NotImplementedError('the MI355X path covers RGB(+1) -> 1 band with ngf % 4 == 0')
        self.n_blocks, self.data_pad = n_blocks, 0
        self.model = nn.Sequential(*_resnet_sequence(input_nc, output_nc, ngf, norm_layer, use_dropout, n_blocks, padding_type))

    def _make_engine(self, key):
        B, H, W, pad, need_bwd = key
        f = self._flat()
        return GeneratorEngine(f.param_views(), f.grad_views(), self.n_blocks, B, H, W, data_pad=pad, need_backward=need_bwd,
                               precision=getattr(self, "precision", "fp32"))

    def forward(self, input):
        return HF.GeneratorFn.apply(self, torch.is_grad_enabled(), input, None, *self.parameters())


class NLayerDiscriminator(_HipNet):
    """70x70 PatchGAN (networks.py:539-584) on the HIP engines."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d):
        super().__init__()
        if not _is_instance(norm_layer):
            raise NotImplementedError('only norm="instance" runs on the MI355X path (the shipped configs use it)')
        if n_layers != 3 or input_nc != 4 or ndf % 4:
            raise NotImplementedError('the MI355X path covers the basic PatchGAN: n_layers=3 on cat(rgb, nir)')
        use_bias = True
        kw, padw = 4, 1
        sequence = [Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        sequence += [Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = nn.Sequential(*sequence)

    def _make_engine(self, key):
        B, H, W, need_bwd = key
        f = self._flat()
        return DiscriminatorEngine(f.param_views(), f.grad_views(), B, H, W, need_backward=need_bwd,
                                   precision=getattr(self, "precision", "fp32"))

    def forward(self, input):
        return HF.DiscriminatorFn.apply(self, torch.is_grad_enabled(), input, *self.parameters())


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:120-160."""
    norm_layer = get_norm_layer(norm_type=norm)
    if netG == 'resnet_9blocks':
        net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout, n_blocks=9)
    elif netG == 'resnet_6blocks':
        net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout, n_blocks=6)
    elif netG in ('unet_128', 'unet_256'):
        raise NotImplementedError('Generator model name [%s] is not on the MI355X path (no shipped config selects it)' % netG)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:163-204."""
    norm_layer = get_norm_layer(norm_type=norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer)
    elif netD == 'pixel':
        raise NotImplementedError('Discriminator model name [pixel] is not on the MI355X path (no shipped config selects it)')
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


class GANLoss(nn.Module):
    """networks.py:210-276.  'lsgan' runs the fused HIP loss; the label mask is the reference's
    0-dim fp32 buffer expanded to the prediction's shape (bit-exact 1.0 / 0.0)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super(GANLoss, self).__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode == 'lsgan':
            self.loss = None
        elif gan_mode in ('vanilla', 'wgangp'):
            raise NotImplementedError('gan mode %s is not on the MI355X path (the shipped configs use lsgan)' % gan_mode)
        else:
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        self._label_cache = {}

    def _label_value(self, target_is_real: bool) -> float:
        """The float the fused loss kernel takes, read from the registered buffer (networks.py:229-230) -- a checkpoint that carries
        other ``criterionGAN.real_label / fake_label`` values (train.py:61-65) or an in-place edit must reach the kernel.  One host
        read per CHANGE of the buffer (keyed on storage and torch's version counter), none in the steady state."""
        t = self.real_label if target_is_real else self.fake_label
        key = (t.data_ptr(), t._version, t.device)
        hit = self._label_cache.get(bool(target_is_real))
        if hit is None or hit[0] != key:
            hit = (key, float(t.detach().float().cpu().item()))
            self._label_cache[bool(target_is_real)] = hit
        return hit[1]

    def get_target_tensor(self, prediction, target_is_real):
        target_tensor = self.real_label if target_is_real else self.fake_label
        return target_tensor.expand_as(prediction)

    def __call__(self, prediction, target_is_real):
        return HF.LsganFn.apply(prediction, self._label_value(target_is_real))
