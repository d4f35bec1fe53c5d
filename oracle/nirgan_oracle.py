"""CPU oracle for the NIR-GAN Pix2Pix hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch fp32 functional ops) of the reference's
RGB->NIR Pix2Pix train/inference path.  It is the checker the HIP path is compared with;
it is never the thing measured or shipped.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``nir-gan_amd/``) never imports anything under ``oracle/``.

Parity pin: the reference has no tests or golden vectors of its own, so
this oracle is pinned by outputs of the reference itself: ``oracle/make_golden.py`` imports
``/root/reference/model/networks.py``, ``model/generator_inject.py`` and
``utils/remote_sensing_indices.py`` in the build container, runs them on seeded inputs and
commits inputs + expected outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.  Parts of the path that live in
``model/pix2pix.py`` (not importable: needs pytorch_lightning/omegaconf/wandb/kornia) are
restated from the source text and are "parity unpinned" beyond the fixtures of the
functions they call; they are marked [text] below.

Parameters are plain dicts keyed by the reference's ``state_dict`` names
(``model.1.weight`` ...), tensors in the reference's layouts (Conv2d ``Cout,Cin,kh,kw``,
ConvTranspose2d ``Cin,Cout,kh,kw``, Linear ``out,in``).

All ``file:line`` citations are relative to the reference repository root.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

IN_EPS = 1e-5  # torch.nn.InstanceNorm2d default eps (model/networks.py:30)


# --------------------------------------------------------------------------------------
# key layout of the reference Sequential (model/networks.py:341-374, 559-580)
# --------------------------------------------------------------------------------------
def generator_keys(n_blocks: int) -> dict:
    """Indices of the parametrised entries of ResnetGenerator.model (networks.py:341-370)."""
    first = 1                      # [0] ReflectionPad2d(3) [1] Conv7x7 [2] IN [3] ReLU
    down = [4, 7]                  # Conv3x3 s2, IN, ReLU  (x2)
    blocks = list(range(10, 10 + n_blocks))
    up0 = 10 + n_blocks            # ConvT, IN, ReLU (x2)
    up = [up0, up0 + 3]
    last = up0 + 7                 # ReflectionPad2d(3) at up0+6, Conv7x7 at up0+7, Tanh
    return {"first": first, "down": down, "blocks": blocks, "up": up, "last": last}


DISC_CONV_IDX = [0, 2, 5, 8, 11]   # NLayerDiscriminator(n_layers=3) (networks.py:559-580)


# ---------------------------------------------------------------------------------------------------------
# Operand precision of the contractions.  The reference computes everything in fp32 ('fp32', the default and the
# only mode pinned by the golden fixtures).  'bf16' restates the build's opt-in bf16-MFMA mode (BASELINE.json
# configs[4]): BOTH operands of every Conv2d / ConvTranspose2d / Linear contraction -- forward, data gradient and
# weight gradient -- are rounded to bf16 (nearest even), products and sums are fp32, everything else is fp32.
# The reference has no bf16 path: this mode is parity UNPINNED against the reference (it converges to the fp32
# path as the rounding is removed, which tests/ check by construction: same code, rounding function = identity).
# `y_bf16_min_pixels` (bf16 mode only) restates the build's storage rule for convolution outputs in front of an InstanceNorm: on maps of at
# least that many pixels per sample (and H*W -- per sub-pixel phase for a transposed convolution -- a multiple of 128: the build's tiles)
# the statistics come from the fp32 result, the tensor itself is kept rounded to bf16, and both the normalisation and its backward
# use the rounded values (_StoredBf16Norm).  None = every tensor stays fp32.  `y_bf16_min_tiles`: the build stores bf16 only from launches
# of at least that many 128 x 128 output tiles (B * H * W / phases pixels x C channels; smaller problems divide K over several
# workgroups and keep fp32): 400 in the build, 0 = no such limit.
# `g_bf16_min_tiles` (bf16 mode only; None = off) restates the same for the data gradients that feed an InstanceNorm backward: the
# launch that computes the gradient wrt a layer's input stores it as bf16 (from that many output tiles on), i.e. dx of the marked
# contractions (argument g16 of _conv2d / _conv_transpose2d) is rounded once more.
_PRECISION = "fp32"
_Y16_MIN_PIXELS = None
_G16_MIN_TILES = None


class operand_precision:
    def __init__(self, mode: str, y_bf16_min_pixels=None, y_bf16_min_tiles=0, g_bf16_min_tiles=None):
        assert mode in ("fp32", "bf16"), mode
        assert (y_bf16_min_pixels is None and g_bf16_min_tiles is None) or mode == "bf16"
        self.mode, self.y16 = mode, (None if y_bf16_min_pixels is None else (y_bf16_min_pixels, y_bf16_min_tiles))
        self.g16 = g_bf16_min_tiles

    def __enter__(self):
        global _PRECISION, _Y16_MIN_PIXELS, _G16_MIN_TILES
        self.prev, _PRECISION = (_PRECISION, _Y16_MIN_PIXELS, _G16_MIN_TILES), self.mode
        _Y16_MIN_PIXELS, _G16_MIN_TILES = self.y16, self.g16

    def __exit__(self, *exc):
        global _PRECISION, _Y16_MIN_PIXELS, _G16_MIN_TILES
        _PRECISION, _Y16_MIN_PIXELS, _G16_MIN_TILES = self.prev


def _launch_tiles(pixels: int, channels: int) -> int:
    """128 x 128 output tiles of a launch over `pixels` x `channels` (the build's 64-wide tile for <= 64 channels)"""
    return -(-pixels // 128) * (-(-channels // 128) if channels > 64 else 1)


def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundedContraction(torch.autograd.Function):
    """y = fn(bf(x), bf(w)); dx, dw = vjp of fn at (bf(x), bf(w)) applied to bf(dy)."""

    @staticmethod
    def forward(ctx, x, w, fn, g16=None):
        ctx.fn, ctx.g16 = fn, g16
        ctx.save_for_backward(x, w)
        with torch.no_grad():
            return fn(_bf(x), _bf(w))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xr, wr = _bf(x).detach().requires_grad_(True), _bf(w).detach().requires_grad_(True)
            y = ctx.fn(xr, wr)
            dx, dw = torch.autograd.grad(y, (xr, wr), _bf(dy))
        if ctx.g16 is not None and _G16_MIN_TILES is not None and dx.dim() == 4 and dx.shape[1] % 4 == 0:
            phases, pad = ctx.g16           # producer = `phases` sub-problems; x carried a halo of `pad` the build does not count
            pixels = dx.shape[0] * (dx.shape[2] - 2 * pad) * (dx.shape[3] - 2 * pad) // phases
            if _launch_tiles(pixels, dx.shape[1]) >= _G16_MIN_TILES:
                dx = _bf(dx)
        return dx, dw, None, None


def _rounded(fn, x, w, bias, bias_shape, g16=None):
    y = _RoundedContraction.apply(x, w, fn, g16)
    return y if bias is None else y + bias.view(bias_shape)


def _conv2d(x, w, b=None, stride=1, padding=0, g16=None):
    """g16 (bf16 mode's storage rule for data gradients, see _G16_MIN_TILES): halo of x that the build's gradient buffer does not count
    (1 behind F.pad(.., 1), else 0) when the gradient wrt x feeds an InstanceNorm backward; None where it does not."""
    if _PRECISION == "fp32":
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    return _rounded(lambda a, k: F.conv2d(a, k, None, stride=stride, padding=padding), x, w, b, (1, -1, 1, 1),
                    None if g16 is None else (stride * stride, g16))


def _conv_transpose2d(x, w, b=None, stride=1, padding=0, output_padding=0, g16=None):
    if _PRECISION == "fp32":
        return F.conv_transpose2d(x, w, b, stride=stride, padding=padding, output_padding=output_padding)
    return _rounded(lambda a, k: F.conv_transpose2d(a, k, None, stride=stride, padding=padding, output_padding=output_padding),
                    x, w, b, (1, -1, 1, 1), None if g16 is None else (1, g16))


def _linear(x, w, b=None):
    if _PRECISION == "fp32":
        return F.linear(x, w, b)
    return _rounded(lambda a, k: F.linear(a, k), x, w, b, (1, -1))


# --------------------------------------------------------------------------------------
# Kinks (test / diagnostic seam).  ReLU, LeakyReLU and |.| are the only non-smooth points of the step.  An element within fp32
# rounding of a kink takes different branches in two CORRECT evaluations (26 M activations per tile: a few hundred always do), and
# one flipped activation moves a gradient tensor by ~1e-3 of its norm -- which is what limits any fp32-vs-fp64 gradient comparison
# at full size.  `record_kinks()` collects the branch decisions of an evaluation in call order; `forced_kinks(masks)` makes an
# evaluation TAKE recorded decisions (from another evaluation of the same step, e.g. the device's), after which the comparison is
# between smooth functions and can be tight.
# --------------------------------------------------------------------------------------
_KINKS = None        # None | ("record", list) | ("force", iterator)


class record_kinks:
    def __enter__(self):
        global _KINKS
        self.masks = []
        _KINKS = ("record", self.masks)
        return self.masks

    def __exit__(self, *exc):
        global _KINKS
        _KINKS = None


class forced_kinks:
    def __init__(self, masks):
        self.masks = list(masks)

    def __enter__(self):
        global _KINKS
        self.it = iter(self.masks)
        _KINKS = ("force", self.it)
        return self

    def __exit__(self, *exc):
        global _KINKS
        _KINKS = None
        if exc[0] is None:
            assert next(self.it, None) is None, "forced_kinks: more recorded decisions than kinks evaluated"


def _branch(x: torch.Tensor) -> torch.Tensor:
    """1 where the kink's upper branch is taken (x > 0): from x itself, or the recorded decision."""
    if _KINKS is None:
        return x > 0
    if _KINKS[0] == "record":
        _KINKS[1].append((x > 0).detach())
        return x > 0
    m = next(_KINKS[1])
    assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
    return m


def _relu(x: torch.Tensor) -> torch.Tensor:
    return F.relu(x) if _KINKS is None else x * _branch(x).to(x.dtype)


def _lrelu(x: torch.Tensor, slope: float) -> torch.Tensor:
    if _KINKS is None:
        return F.leaky_relu(x, slope)
    m = _branch(x).to(x.dtype)
    return x * (m + slope * (1 - m))


class _StoredBf16Norm(torch.autograd.Function):
    """InstanceNorm2d of a tensor that is STORED as bf16 (build's bf16 mode, see _Y16_MIN_PIXELS): mean / variance of the fp32 values,
    z = (bf(y) - mean) * rstd; backward by the instance-norm rule evaluated at that z:
    dy = rstd * (g - mean(g) - z * mean(g * z))."""

    @staticmethod
    def forward(ctx, y):
        mean = y.mean((2, 3), keepdim=True)
        var = y.var((2, 3), unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + IN_EPS)
        z = (_bf(y) - mean) * rstd
        ctx.save_for_backward(z, rstd)
        return z

    @staticmethod
    def backward(ctx, g):
        z, rstd = ctx.saved_tensors
        return rstd * (g - g.mean((2, 3), keepdim=True) - z * (g * z).mean((2, 3), keepdim=True))


def _inorm(x: torch.Tensor, phases: int = 1) -> torch.Tensor:
    # InstanceNorm2d(affine=False, track_running_stats=False): networks.py:30.  phases = 4 behind a stride-2 transposed convolution
    # (the build computes it as four sub-pixel problems: its storage rule looks at the pixels of one)
    hw, c = x.shape[-2] * x.shape[-1], x.shape[1]
    if _PRECISION == "bf16" and _Y16_MIN_PIXELS is not None and hw >= _Y16_MIN_PIXELS[0] and (hw // phases) % 128 == 0 and c % 4 == 0:
        tiles = -(-(x.shape[0] * hw // phases) // 128) * (-(-c // 128) if c > 64 else 1)
        if tiles >= _Y16_MIN_PIXELS[1]:
            return _StoredBf16Norm.apply(x)
    return F.instance_norm(x, eps=IN_EPS)


# --------------------------------------------------------------------------------------
# generator (model/networks.py:316-374, ResnetBlock :377-434)
# --------------------------------------------------------------------------------------
def generator_trunk_head(p: Params, x: torch.Tensor) -> torch.Tensor:
    """model[:6] of the reference: pad3, conv7, IN, ReLU, conv3 s2, IN (generator_inject.py:107)."""
    x = _conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), p["model.1.weight"], p["model.1.bias"])
    x = _relu(_inorm(x))
    x = _conv2d(x, p["model.4.weight"], p["model.4.bias"], stride=2, padding=1, g16=0)
    return _inorm(x)


def generator_trunk_tail(p: Params, x: torch.Tensor, n_blocks: int, modulated: bool = False) -> torch.Tensor:
    """model[6:] of the reference: ReLU, conv3 s2, IN, ReLU, blocks, 2x convT, pad3, conv7, tanh."""
    k = generator_keys(n_blocks)
    x = _relu(x)
    # (modulated: the gradient wrt x goes to the SatCLIP modulation's backward, an fp32 tensor in the build)
    x = _conv2d(x, p["model.7.weight"], p["model.7.bias"], stride=2, padding=1, g16=(None if modulated else 0))
    x = _relu(_inorm(x))
    for i in k["blocks"]:          # ResnetBlock.forward: out = x + conv_block(x) (networks.py:430-434)
        h = _conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"),
                     p[f"model.{i}.conv_block.1.weight"], p[f"model.{i}.conv_block.1.bias"], g16=1)
        h = _relu(_inorm(h))
        h = _conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"),
                     p[f"model.{i}.conv_block.5.weight"], p[f"model.{i}.conv_block.5.bias"], g16=1)
        x = x + _inorm(h)
    for i in k["up"]:              # ConvTranspose2d k3 s2 p1 op1 (networks.py:360-363)
        # (the gradient wrt the first up-convolution's input is the skip path's dense fp32 tensor in the build: no g16 there)
        x = _conv_transpose2d(x, p[f"model.{i}.weight"], p[f"model.{i}.bias"],
                               stride=2, padding=1, output_padding=1, g16=(None if i == k["up"][0] else 0))
        x = _relu(_inorm(x, phases=4))
    i = k["last"]
    x = _conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), p[f"model.{i}.weight"], p[f"model.{i}.bias"])
    return torch.tanh(x)


def generator_forward(p: Params, x: torch.Tensor, n_blocks: int) -> torch.Tensor:
    """ResnetGenerator.forward (networks.py:372-374)."""
    return generator_trunk_tail(p, generator_trunk_head(p, x), n_blocks)


def inject_modulation(p: Params, x: torch.Tensor, embeds: torch.Tensor, style: str = "multiply",
                      use_scale: bool = True) -> torch.Tensor:
    """The SatCLIP injection of ResnetGenerator_inject.forward (generator_inject.py:110-127).

    x: B x C x H x W (post-IN, pre-ReLU).  embeds: B x 256.
    """
    e = _linear(embeds, p["fc.weight"], p["fc.bias"])            # :110
    e = e.view(-1, 1, 128, 128)                                  # :113
    # NB the reference passes size=(x.shape[-1], x.shape[-2]) i.e. (W, H)            # :116
    e = F.interpolate(e, size=(x.shape[-1], x.shape[-2]), mode="bilinear", align_corners=False)
    e = e.repeat(1, x.shape[-3], 1, 1)                            # :119
    if style == "add":                                            # :122-123
        return x + p["scale_param"] * e
    if style == "multiply" and use_scale:                         # :124-125
        return x * (1 + p["scale_param"] * e)
    if style == "multiply":                                       # :126-127
        return x * e
    raise NotImplementedError(style)


def generator_inject_forward(p: Params, x: torch.Tensor, embeds: torch.Tensor, n_blocks: int = 9,
                             style: str = "multiply", use_scale: bool = True,
                             post_correction: bool = False) -> torch.Tensor:
    """ResnetGenerator_inject.forward (generator_inject.py:105-135)."""
    h = generator_trunk_head(p, x)
    h = inject_modulation(p, h, embeds, style, use_scale)
    out = generator_trunk_tail(p, h, n_blocks, modulated=True)
    if post_correction:                                           # :133-134
        out = out * p["post_correction_param"]
    return out


# --------------------------------------------------------------------------------------
# discriminator (model/networks.py:539-584)
# --------------------------------------------------------------------------------------
def discriminator_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """NLayerDiscriminator(n_layers=3).forward: 70x70 PatchGAN, no sigmoid."""
    x = _lrelu(_conv2d(x, p["model.0.weight"], p["model.0.bias"], stride=2, padding=1), 0.2)
    for i, s in ((2, 2), (5, 2), (8, 1)):
        x = _conv2d(x, p[f"model.{i}.weight"], p[f"model.{i}.bias"], stride=s, padding=1, g16=0)
        x = _lrelu(_inorm(x), 0.2)
    return _conv2d(x, p["model.11.weight"], p["model.11.bias"], stride=1, padding=1)


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def gan_target_tensor(pred: torch.Tensor, target_is_real: bool,
                      real_label: float = 1.0, fake_label: float = 0.0) -> torch.Tensor:
    """GANLoss.get_target_tensor (networks.py:241-256): 0-dim fp32 label expanded to pred's shape."""
    lab = torch.tensor(real_label if target_is_real else fake_label, dtype=torch.float32)
    return lab.to(pred.dtype).expand_as(pred)      # fp32 in the reference; the cast only serves fp64 diagnostics


def lsgan_loss(pred: torch.Tensor, target_is_real: bool) -> torch.Tensor:
    """GANLoss('lsgan').__call__ (networks.py:258-276): MSELoss, mean over all elements."""
    return F.mse_loss(pred, gan_target_tensor(pred, target_is_real))


def l1_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """torch.nn.L1Loss() (pix2pix.py:60, used :222)."""
    if _KINKS is None:
        return F.l1_loss(pred, target)
    d = pred - target
    m = _branch(d).to(d.dtype)
    return (d * (2 * m - 1)).mean()


def _crit(name: str):
    if name == "l1":
        return F.l1_loss
    if name == "l2":
        return F.mse_loss
    raise NotImplementedError(name)


def rs_index_pairs(rgb, nir, pred, mode: str = "loss") -> dict:
    """All six spectral indices for (nir, pred) (remote_sensing_indices.py:84-319).

    mode 'loss' uses the reference's epsilons (ndvi/ndwi/evi only); 'index' none.
    """
    eps = 1e-6 if mode == "loss" else 0.0
    red, green, blue = rgb[:, 0:1], rgb[:, 1:2], rgb[:, 2:3]
    out = {}
    out["ndvi"] = ((nir - red) / (nir + red + eps), (pred - red) / (pred + red + eps))           # :109-111
    out["ndwi"] = ((nir - green) / (nir + green + eps), (pred - green) / (pred + green + eps))   # :148-151
    nd, ndp = (nir - red) / (nir + red), (pred - red) / (pred + red)                             # :186-187
    out["gndvi"] = ((nir - green) / (nd + green), (pred - green) / (ndp + green))                # :190-191
    out["savi"] = (1.5 * (nir - red) / (nir + red + 0.5), 1.5 * (pred - red) / (pred + red + 0.5))  # :226-227
    out["msavi"] = ((2 * nir + 1 - torch.sqrt((2 * nir + 1) ** 2 - 8 * (nir - red))) / 2,
                    (2 * pred + 1 - torch.sqrt((2 * pred + 1) ** 2 - 8 * (pred - red))) / 2)     # :263-264
    l, c1, c2, g = 1, 6, 7.5, 2.5                                                                # :295
    if mode == "loss":
        den = (nir + c1) * (red - c2) * (blue + l) + 1e-6                                        # :304-305
        denp = (pred + c1) * (red - c2) * (blue + l) + 1e-6
    else:
        den = (nir + c1) * (red - c2) * (blue + l)                                               # :313-314
        denp = (pred + c1) * (red - c2) * (blue + l)
    out["evi"] = (g * ((nir - red) / den), g * ((pred - red) / denp))
    return out


RS_ORDER = ["ndvi", "ndwi", "gndvi", "savi", "msavi", "evi"]   # dict order of loss_fns (:45-52)
RS_DEFAULT = {"lambda_ndvi": 0.333, "lambda_ndwi": 0.333, "lambda_evi": 0.333,
              "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}   # :37-43


def rs_weighted_loss(rgb, nir, pred, loss_config: Optional[dict] = None, criterion: str = "l1"):
    """RemoteSensingIndices('loss', criterion).get_and_weight_losses(mode='loss') (:23-61)."""
    cfg = RS_DEFAULT if loss_config is None else loss_config
    crit = _crit(criterion)
    idx = rs_index_pairs(rgb, nir, pred, "loss")
    total = 0.0
    for name in RS_ORDER:
        w = cfg.get("lambda_" + name, 0.0)
        if w > 0.0:
            a, b = idx[name]
            if _KINKS is not None and criterion == "l1":          # |a - b| is a kink too (record / force it like the ReLUs)
                d = a - b
                total = total + w * (d * (2 * _branch(d).to(d.dtype) - 1)).mean()
            else:
                total = total + w * crit(a, b)
    return total


def rs_logging_dict(rgb, nir, pred, criterion: str = "l1") -> dict:
    """get_and_weight_losses(mode='logging_dict') (:63-68)."""
    crit = _crit(criterion)
    idx = rs_index_pairs(rgb, nir, pred, "loss")
    return {f"indices_loss/{n}_error": crit(*idx[n]) for n in RS_ORDER}


# --------------------------------------------------------------------------------------
# train / validation metrics: utils/calculate_metrics.py:5-36.  kornia==0.7.3 (requirements.txt:9)
# is NOT installed here and not vendored under /root/reference: its published algorithm is restated --
# kornia.metrics.ssim(img1, img2, window_size, max_val=1.0, eps=1e-12, padding='same'): 1-D Gaussian
# exp(-x^2 / (2*1.5^2)) normalised to sum 1, applied separably with border_type='reflect'; C1 = (0.01*max_val)^2,
# C2 = (0.03*max_val)^2; ssim = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2) + eps);
# kornia.metrics.psnr = 10 log10(max_val^2 / mse).  PARITY UNPINNED against kornia itself; pinned by known answers
# (tests/test_oracle_golden.py::test_metric_known_answers).
# --------------------------------------------------------------------------------------
def ssim_map(a: torch.Tensor, b: torch.Tensor, window_size: int = 5, max_val: float = 1.0, eps: float = 1e-12) -> torch.Tensor:
    r = window_size // 2
    x = torch.arange(window_size, dtype=a.dtype) - r
    k = torch.exp(-x * x / (2.0 * 1.5 ** 2))
    k = k / k.sum()
    C = a.shape[1]
    kh, kw = k.view(1, 1, -1, 1).repeat(C, 1, 1, 1), k.view(1, 1, 1, -1).repeat(C, 1, 1, 1)

    def filt(t):
        t = F.pad(t, (r, r, r, r), mode="reflect")
        return F.conv2d(F.conv2d(t, kw, groups=C), kh, groups=C)
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    mu1, mu2 = filt(a), filt(b)
    s1, s2, s12 = filt(a * a) - mu1 * mu1, filt(b * b) - mu2 * mu2, filt(a * b) - mu1 * mu2
    return ((2.0 * mu1 * mu2 + c1) * (2.0 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2) + eps)


def calculate_metrics(pred: torch.Tensor, target: torch.Tensor, phase: str = "train") -> dict:
    """utils/calculate_metrics.py:5-36."""
    l2 = F.mse_loss(pred, target).item()
    return {phase + "/L1": F.l1_loss(pred, target).item(), phase + "/L2": l2,
            phase + "/PSNR": 10.0 * math.log10(1.0 / l2) if l2 > 0 else float("inf"),
            phase + "/SSIM": ssim_map(pred, target, 5, 1.0).mean().item()}


# --------------------------------------------------------------------------------------
# SatCLIP location encoder, fp64 (load_lightweight.py:29).
#   spherical_harmonics: positional_encoding/spherical_harmonics.py:26-42 with the closed-form SH of
#     spherical_harmonics_closed_form.py:8-40 -- PINNED: oracle/make_golden.py imports that file and commits
#     tests/golden/f6_locenc.npz.  The 'analytic' variant (spherical_harmonics_ylm.py, the default and what the published
#     checkpoints record) is missing from the reference tree, but its generator spherical_harmonics_generate_ylms.py is there:
#     as its text reads it differs from closed-form by (-1)^m for m != 0 and by a factor pi for m == 0 (sh_factor,
#     'analytic-generator-text') -- PINNED: make_golden.py::f7 evaluates the generator's calc_ylm (sympy) and commits
#     f7_sh_analytic.npz.  The factor pi is an operator-precedence slip of that text which the published table does not show
#     (its Yl0_m0 is 0.28209...): the default 'analytic' keeps the orthonormal constant for m == 0 (pinned by f6 there, by f7 elsewhere).
#   siren_forward: location_encoder.py:73-151 restated from the text -- the module is NOT importable
#     (model/satclip/__init__.py pulls pytorch_lightning; positional_encoding/__init__.py needs the missing ylm file):
#     parity UNPINNED for the MLP part; it is three F.linear + sin calls.
# --------------------------------------------------------------------------------------
def _assoc_legendre(l: int, m: int, x: torch.Tensor) -> torch.Tensor:
    pmm = torch.ones_like(x)
    if m > 0:
        somx2 = torch.sqrt((1 - x) * (1 + x))
        fact = 1.0
        for _ in range(1, m + 1):
            pmm = pmm * (-fact) * somx2
            fact += 2.0
    if l == m:
        return pmm
    pmmp1 = x * (2.0 * m + 1.0) * pmm
    if l == m + 1:
        return pmmp1
    pll = torch.zeros_like(x)
    for ll in range(m + 2, l + 1):
        pll = ((2.0 * ll - 1.0) * x * pmmp1 - (ll + m - 1.0) * pmm) / (ll - m)
        pmm, pmmp1 = pmmp1, pll
    return pll


def sh_factor(l: int, m: int, calculation: str = "closed-form") -> float:
    """Constant in front of P_l^|m|(cos theta) * {1, cos(m phi), sin(|m| phi)}; P carries the Condon-Shortley phase in every variant.

    'closed-form' (spherical_harmonics_closed_form.py:25-40): sqrt2 (m != 0) * sqrt((2l+1)(l-|m|)! / (4 pi (l+|m|)!)).
    'analytic-generator-text' (spherical_harmonics_generate_ylms.py:19-36, the generator of the missing spherical_harmonics_ylm.py,
    exactly as its text reads): the same normalisation times (-1)**m for m != 0 (sympy's assoc_legendre already holds the phase, the
    script multiplies it in again), and for m == 0 the script's ``sqrt((2*l + 1) / 4 * pi)`` = sqrt((2l+1) pi / 4) (operator
    precedence: pi is multiplied).  PINNED for every feature by fixture f7.
    'analytic' (the build's default for published checkpoints): the generator's sign on m != 0, the ORTHONORMAL constant on m == 0
    (the published table's first entry is 0.28209... = sqrt(1/(4 pi)), not 0.886...) -- pinned by f7 on m != 0 and by f6 on m == 0."""
    am = abs(m)
    k = math.sqrt((2.0 * l + 1.0) * math.factorial(l - am) / (4 * math.pi * math.factorial(l + am)))
    if calculation == "closed-form":
        return k if m == 0 else math.sqrt(2.0) * k
    if calculation in ("analytic", "analytic-generator-text"):
        if m != 0:
            return (-1.0) ** am * math.sqrt(2.0) * k
        return k if calculation == "analytic" else math.sqrt((2 * l + 1) / 4 * math.pi)
    raise NotImplementedError(calculation)


def spherical_harmonics(lonlat: torch.Tensor, legendre_polys: int = 10, calculation: str = "closed-form") -> torch.Tensor:
    lon, lat = lonlat[:, 0], lonlat[:, 1]
    phi, theta = torch.deg2rad(lon + 180), torch.deg2rad(lat + 90)
    ct = torch.cos(theta)
    Y = []
    for l in range(legendre_polys):
        for m in range(-l, l + 1):
            am = abs(m)
            k = sh_factor(l, m, calculation)
            if m == 0:
                y = k * _assoc_legendre(l, 0, ct)
            elif m > 0:
                y = k * torch.cos(m * phi) * _assoc_legendre(l, m, ct)
            else:
                y = k * torch.sin(-m * phi) * _assoc_legendre(l, am, ct)
            Y.append(y)
    return torch.stack(Y, dim=-1)


def siren_forward(p: Params, x: torch.Tensor, num_layers: int, w0: float = 1.0, w0_initial: float = 30.0) -> torch.Tensor:
    """SirenNet.forward in eval mode (dropout off): keys nnet.layers.{i}.weight/bias, nnet.last_layer.weight/bias."""
    for i in range(num_layers):
        x = torch.sin((w0_initial if i == 0 else w0) * F.linear(x, p[f"nnet.layers.{i}.weight"], p.get(f"nnet.layers.{i}.bias")))
    return F.linear(x, p["nnet.last_layer.weight"], p.get("nnet.last_layer.bias"))


def location_encoder_forward(p: Params, lonlat: torch.Tensor, legendre_polys: int, num_layers: int,
                             calculation: str = "closed-form") -> torch.Tensor:
    """LocationEncoder.forward (location_encoder.py:267-274) on fp64 lon/lat."""
    return siren_forward(p, spherical_harmonics(lonlat.double(), legendre_polys, calculation), num_layers)


# --------------------------------------------------------------------------------------
# Histogram matching: create_synthetic_dataset.py:34-47.  scikit-image is NOT installed here and not
# vendored under /root/reference (the reference pins no version for it): the published float path of
# skimage.exposure.match_histograms(image, reference, channel_axis=None) -- _match_cumulative_cdf -- is restated with
# numpy.  PARITY UNPINNED against scikit-image itself; pinned by known answers (tests/test_oracle_golden.py).
# --------------------------------------------------------------------------------------
def match_histograms_plane(source, template):
    import numpy as np
    src_values, src_lookup, src_counts = np.unique(source.reshape(-1), return_inverse=True, return_counts=True)
    tmpl_values, tmpl_counts = np.unique(template.reshape(-1), return_counts=True)
    src_quantiles = np.cumsum(src_counts) / source.size
    tmpl_quantiles = np.cumsum(tmpl_counts) / template.size
    interp_a_values = np.interp(src_quantiles, tmpl_quantiles, tmpl_values)
    return interp_a_values[src_lookup].reshape(source.shape).astype(source.dtype, copy=False)


def histogram_match(image: torch.Tensor, reference: torch.Tensor) -> torch.Tensor:
    """create_synthetic_dataset.py:34-47: resize the reference to the tile (bilinear), match each tile; [B, 1, H, W]."""
    reference = F.interpolate(reference, size=image.shape[-2:], mode="bilinear", align_corners=False)
    matched = []
    for img, ref in zip(image, reference):
        m = match_histograms_plane(img.squeeze().cpu().numpy(), ref.squeeze().cpu().numpy())
        matched.append(torch.from_numpy(m).unsqueeze(0))
    return torch.stack(matched, dim=0)


# --------------------------------------------------------------------------------------
# Px2Px_PL orchestration  [text: model/pix2pix.py is not importable here]
# --------------------------------------------------------------------------------------
def px_forward(pG: Params, rgb: torch.Tensor, n_blocks: int, padding: int = 0,
               embeds: Optional[torch.Tensor] = None, inject_cfg: Optional[dict] = None) -> torch.Tensor:
    """Px2Px_PL.forward (pix2pix.py:88-110): reflect-pad -> netG -> crop."""
    x = rgb
    if padding:
        x = F.pad(x, (padding,) * 4, mode="reflect")                      # :91-93
    if embeds is None:
        y = generator_forward(pG, x, n_blocks)                            # :97
    else:
        c = inject_cfg or {}
        y = generator_inject_forward(pG, x, embeds, n_blocks, c.get("style", "multiply"),
                                     c.get("use_scale", True), c.get("post_correction", False))  # :102
    if padding:
        y = y[..., padding:-padding, padding:-padding]                    # :107-108
    return y


def d_step_loss(pD: Params, rgb, nir, pred) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """optimizer_idx 0 (pix2pix.py:195-212): loss_D = MSE(D(fake.detach()),0) + MSE(D(real),1), no 0.5."""
    pred_fake = discriminator_forward(pD, torch.cat((rgb, pred), 1).detach())
    loss_fake = lsgan_loss(pred_fake, False)
    pred_real = discriminator_forward(pD, torch.cat((rgb, nir), 1))
    loss_real = lsgan_loss(pred_real, True)
    return loss_fake + loss_real, loss_fake, loss_real


def ssim_loss(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """utils/losses.py:10-30: 1 - kornia.metrics.ssim(img1, img2, window_size).mean() (restated by ssim_map above)."""
    return 1.0 - ssim_map(img1.float(), img2.float(), window_size).mean()


def emd_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """utils/losses.py:64-78."""
    pred, target = pred.reshape(pred.shape[0], -1), target.reshape(target.shape[0], -1)
    return torch.mean(torch.abs(torch.cumsum(F.softmax(pred, dim=1), dim=1) - torch.cumsum(F.softmax(target, dim=1), dim=1)))


def g_step_loss(pD: Params, rgb, nir, pred, lambda_gan=1.0, lambda_l1=100.0, lambda_rs=0.0,
                rs_weights: Optional[dict] = None, rs_criterion: str = "l1", lambda_ssim: float = 0.0) -> Tuple[torch.Tensor, dict]:
    """optimizer_idx 1 (pix2pix.py:215-257)."""
    pred_fake = discriminator_forward(pD, torch.cat((rgb, pred), 1))
    l_gan = lsgan_loss(pred_fake, True)                                   # :220
    l_l1 = l1_loss(pred, nir)                                             # :222
    loss = l_gan * lambda_gan + l_l1 * lambda_l1                          # :226-229
    parts = {"gan": l_gan, "l1": l_l1}
    if lambda_ssim > 0.0:                                                 # :233-237
        l_ssim = ssim_loss(pred, nir)
        loss = loss + l_ssim * lambda_ssim
        parts["ssim"] = l_ssim
    if lambda_rs > 0.0:                                                   # :246-251
        l_rs = rs_weighted_loss(rgb, nir, pred, rs_weights, rs_criterion)
        loss = loss + l_rs * lambda_rs
        parts["rs"] = l_rs
    return loss, parts


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr=2e-4, b1=0.5, b2=0.999, eps=1e-8) -> None:
    """torch.optim.Adam single-tensor update (pix2pix.py:486-487; torch defaults eps=1e-8, wd=0).

    In place on p, m, v; ``step`` is the 1-based step count AFTER increment.
    """
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class OracleTrainer:
    """One batch of the reference's two-optimizer loop, restated on CPU.  [text]

    Order (pix2pix.py:165-257, :485-492 and Lightning 1.9 multi-optimizer loop):
    optimizer 0 = D: G forward, loss_D, backward (grads of D only: fake is detached), Adam(D);
    optimizer 1 = G: G forward again, D frozen (toggle_optimizer), loss_G, backward, Adam(G).
    """

    def __init__(self, pG: Params, pD: Params, n_blocks: int, padding: int = 0,
                 lr=2e-4, beta1=0.5, lambda_gan=1.0, lambda_l1=100.0, lambda_rs=0.0,
                 rs_weights: Optional[dict] = None, rs_criterion="l1", inject_cfg: Optional[dict] = None, lambda_ssim: float = 0.0,
                 d_loss_scale: float = 1.0):
        self.d_loss_scale = d_loss_scale      # 0.5 = the legacy Pix2PixModel.backward_D (model/pix2pix_model.py:128)
        self.pG = {k: v.detach().clone().requires_grad_(True) for k, v in pG.items()}
        self.pD = {k: v.detach().clone().requires_grad_(True) for k, v in pD.items()}
        self.n_blocks, self.padding = n_blocks, padding
        self.lr, self.beta1 = lr, beta1
        self.lam = (lambda_gan, lambda_l1, lambda_rs)
        self.lambda_ssim = lambda_ssim
        self.rs_weights, self.rs_criterion = rs_weights, rs_criterion
        self.inject_cfg = inject_cfg
        self.state = {id(t): (torch.zeros_like(t), torch.zeros_like(t)) for t in
                      list(self.pG.values()) + list(self.pD.values())}
        self.step_count = 0
        self.last = {}

    def _adam(self, params: Params, grads: dict):
        for k, t in params.items():
            g = grads[k]
            if g is None:
                continue
            m, v = self.state[id(t)]
            with torch.no_grad():
                adam_step(t, g, m, v, self.step_count, self.lr, self.beta1)

    def step(self, rgb, nir, embeds=None) -> dict:
        self.step_count += 1
        lg, l1w, lrs = self.lam
        # ---- optimizer_idx 0: discriminator
        pred = px_forward(self.pG, rgb, self.n_blocks, self.padding, embeds, self.inject_cfg)
        loss_d, lf, lr_ = d_step_loss(self.pD, rgb, nir, pred)
        loss_d = loss_d * self.d_loss_scale if self.d_loss_scale != 1.0 else loss_d
        gD = torch.autograd.grad(loss_d, list(self.pD.values()))
        gD = dict(zip(self.pD.keys(), gD))
        self._adam(self.pD, gD)
        # ---- optimizer_idx 1: generator (D frozen, already updated)
        pred = px_forward(self.pG, rgb, self.n_blocks, self.padding, embeds, self.inject_cfg)
        pD_frozen = {k: v.detach() for k, v in self.pD.items()}
        loss_g, parts = g_step_loss(pD_frozen, rgb, nir, pred, lg, l1w, lrs, self.rs_weights, self.rs_criterion, self.lambda_ssim)
        gG = torch.autograd.grad(loss_g, list(self.pG.values()), allow_unused=True)
        gG = dict(zip(self.pG.keys(), gG))
        self._adam(self.pG, gG)
        self.last = {"pred": pred.detach(), "grads_D": gD, "grads_G": gG}
        out = {"loss_D": loss_d.detach(), "loss_D_fake": lf.detach(), "loss_D_real": lr_.detach(),
               "loss_G": loss_g.detach()}
        out.update({"loss_G_" + k: v.detach() for k, v in parts.items()})
        return out


# --------------------------------------------------------------------------------------
# weight construction in the reference's RNG order (networks.py:68-117)
# --------------------------------------------------------------------------------------
def shadowed_bias_keys(net: str, n_blocks: int = 6) -> set:
    """Biases that feed an InstanceNorm directly (mathematically dead; SURVEY section 7 hard parts)."""
    if net == "D":
        return {"model.2.bias", "model.5.bias", "model.8.bias"}
    k = generator_keys(n_blocks)
    keys = {f"model.{k['first']}.bias"} | {f"model.{i}.bias" for i in k["down"] + k["up"]}
    for i in k["blocks"]:
        keys |= {f"model.{i}.conv_block.1.bias", f"model.{i}.conv_block.5.bias"}
    return keys


# --------------------------------------------------------------------------------------
# tiled inference: restatement of the product's tiling for the parity tests
# --------------------------------------------------------------------------------------
def predict_tiled(model, rgb, tile=512, margin=16, batch=8, embeds=None):
    """Tiled inference stated with torch ops (pad the scene once with reflect, slice overlapping tiles, keep each tile's core): what
    nirgan_hip.inference.predict_tiled does with its gather / scatter kernels (create_synthetic_dataset.py:100-118 per tile;
    model/pix2pix.py:91-93,107-108 for the pad / crop idea).  Checker only."""
    B, _, H, W = rgb.shape
    core = tile - 2 * margin
    ph, pw = (-H) % core, (-W) % core
    x = torch.nn.functional.pad(rgb, (margin, margin + pw, margin, margin + ph), mode="reflect")
    out = torch.empty(B, 1, H + ph, W + pw, dtype=rgb.dtype, device=rgb.device)
    coords = [(b, i, j) for b in range(B) for i in range(0, H + ph, core) for j in range(0, W + pw, core)]
    for k in range(0, len(coords), batch):
        chunk = coords[k:k + batch]
        tiles = torch.stack([x[b, :, i:i + tile, j:j + tile] for b, i, j in chunk])
        e = None if embeds is None else torch.stack([embeds[b] for b, _, _ in chunk])
        pred = model(tiles) if e is None else model(tiles, e)
        for (b, i, j), p in zip(chunk, pred):
            out[b, :, i:i + core, j:j + core] = p[:, margin:margin + core, margin:margin + core]
    return out[:, :, :H, :W]
