"""MI355X-native counterpart of the reference's ``model/satclip/location_encoder.py`` for the one combination the
published SatCLIP checkpoints use: ``le_type='sphericalharmonics'`` + ``pe_type='siren'``.

Same class names, constructor arguments, parameter names and initialisation as the reference
(location_encoder.py:73-151, :205-274; positional_encoding/spherical_harmonics.py:9-42), so
``get_satclip_loc_encoder`` loads the ``nnet.*`` tensors of a SatCLIP checkpoint unchanged.  ``LocationEncoder.forward``
is ONE HIP launch (nirgan_location_encoder): harmonics and every Siren layer fused, fp64 like the reference.
Forward only and evaluation mode only -- the reference uses the encoder frozen, under ``torch.no_grad()``
(satclip_wrapper.py:30-35, model/pix2pix.py:481-484).
"""
import ctypes as C
import math

import torch
from torch import nn

from nirgan_hip import lib as L


HARMONICS_VARIANTS = ("analytic", "closed-form", "analytic-generator-text")


def sh_normalisation(legendre_polys: int, harmonics_calculation: str = "closed-form") -> torch.Tensor:
    """Per-feature constant K[l*l+l+m] in front of P_l^|m|(cos theta) * {1, cos(m phi), sin(|m| phi)} (P with the
    Condon-Shortley phase, as the device recurrence computes it).

    'closed-form' (spherical_harmonics_closed_form.py:25-40): [sqrt 2 if m != 0] * sqrt((2l+1)(l-|m|)! / (4 pi (l+|m|)!)).
    'analytic' (the tabulated spherical_harmonics_ylm.py -- NOT in the reference tree; what published SatCLIP checkpoints record):
      the orthonormal constant for m == 0 (= closed-form) and (-1)**m times the closed-form constant for m != 0: the table's
      generator (spherical_harmonics_generate_ylms.py:19-36) multiplies (-1)**m onto sympy's assoc_legendre, which already
      carries that phase, so odd orders flip sign against closed-form.
    'analytic-generator-text': the generator script EXACTLY as its text in the reference reads, i.e. with its m == 0 branch
      ``sqrt((2*l + 1) / 4 * pi)`` = sqrt((2l+1) pi / 4) -- pi times the orthonormal constant (operator precedence).  The
      published table starts with Yl0_m0 = 0.2820947917... = sqrt(1/(4 pi)), i.e. it was generated with the orthonormal
      constant, so this variant is NOT the default: a checkpoint trained on the table would get its ten zonal inputs scaled by
      pi.  It is kept as an explicit option because it is the only variant the reference tree itself pins for every feature
      (fixture f7 = this script evaluated with sympy); 'analytic' is pinned by f7 on the m != 0 features and by the
      reference's closed-form file (fixture f6) on the m == 0 features."""
    if harmonics_calculation not in HARMONICS_VARIANTS:
        raise NotImplementedError(f"harmonics_calculation [{harmonics_calculation}] is not on the MI355X path {HARMONICS_VARIANTS}")
    out = []
    for l in range(legendre_polys):
        for m in range(-l, l + 1):
            am = abs(m)
            k = math.sqrt((2.0 * l + 1.0) * math.factorial(l - am) / (4 * math.pi * math.factorial(l + am)))
            if harmonics_calculation == "closed-form":
                out.append(k if m == 0 else math.sqrt(2.0) * k)
            elif m != 0:
                out.append((-1.0) ** am * math.sqrt(2.0) * k)
            else:
                out.append(k if harmonics_calculation == "analytic" else math.sqrt((2 * l + 1) / 4 * math.pi))
    return torch.tensor(out, dtype=torch.float64)


class SphericalHarmonics(nn.Module):
    """Positional encoding of lon/lat by real spherical harmonics of degree < legendre_polys
    (positional_encoding/spherical_harmonics.py:9-42).

    The device kernel evaluates the associated-Legendre recurrence for both variants; 'analytic' and 'closed-form' differ
    in the per-feature constant only (sh_normalisation).  Evaluated inside LocationEncoder's fused launch; called on its
    own it runs the same kernel with no layers."""

    def __init__(self, legendre_polys: int = 10, harmonics_calculation="analytic"):
        super().__init__()
        self.L, self.M = int(legendre_polys), int(legendre_polys)
        self.embedding_dim = self.L * self.M
        self.harmonics_calculation = harmonics_calculation
        self.register_buffer("sh_norm", sh_normalisation(self.L, harmonics_calculation), persistent=False)

    def forward(self, lonlat):
        return _run(lonlat, self, [], [], [])[1]


class Sine(nn.Module):
    def __init__(self, w0=1.):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


class Siren(nn.Module):
    """One sine layer; parameters ``weight`` [dim_out, dim_in], ``bias`` [dim_out] with the SIREN initialisation
    (location_encoder.py:119-151)."""

    def __init__(self, dim_in, dim_out, w0=1., c=6., is_first=False, use_bias=True, activation=None, dropout=False):
        super().__init__()
        self.dim_in, self.dim_out, self.is_first, self.dropout = dim_in, dim_out, is_first, dropout
        weight = torch.zeros(dim_out, dim_in)
        bias = torch.zeros(dim_out) if use_bias else None
        w_std = (1 / dim_in) if is_first else (math.sqrt(c / dim_in) / w0)
        weight.uniform_(-w_std, w_std)
        if bias is not None:
            bias.uniform_(-w_std, w_std)
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias) if use_bias else None
        self.activation = Sine(w0) if activation is None else activation


class SirenNet(nn.Module):
    def __init__(self, dim_in, dim_hidden, dim_out, num_layers, w0=1., w0_initial=30., use_bias=True,
                 final_activation=None, degreeinput=False, dropout=True):
        super().__init__()
        if final_activation is not None or degreeinput:
            raise NotImplementedError("SirenNet on the MI355X path: Identity output, harmonics input (what get_neural_network builds)")
        self.num_layers, self.dim_hidden, self.degreeinput = num_layers, dim_hidden, degreeinput
        self.layers = nn.ModuleList([])
        for ind in range(num_layers):
            is_first = ind == 0
            self.layers.append(Siren(dim_in=dim_in if is_first else dim_hidden, dim_out=dim_hidden,
                                     w0=w0_initial if is_first else w0, use_bias=use_bias, is_first=is_first, dropout=dropout))
        self.last_layer = Siren(dim_in=dim_hidden, dim_out=dim_out, w0=w0, use_bias=use_bias, activation=nn.Identity(), dropout=False)

    def linear_stack(self):
        layers = list(self.layers) + [self.last_layer]
        w0 = [float(l.activation.w0) for l in self.layers] + [0.0]
        return [l.weight for l in layers], [l.bias for l in layers], w0


def _run(lonlat, posenc: SphericalHarmonics, weights, biases, w0):
    """One launch of nirgan_location_encoder; returns (embedding or None, harmonics)."""
    if lonlat.dim() != 2 or lonlat.shape[1] != 2:
        raise ValueError(f"lonlat must be [B, 2] (lon, lat in degrees), got {tuple(lonlat.shape)}")
    dev = lonlat.device
    if dev.type != "cuda" and not L.is_emulated():
        raise RuntimeError("nirgan_hip runs on MI355X (cuda device) only; there is no CPU path")
    x = lonlat.detach().to(torch.float64).contiguous()
    B, nf = x.shape[0], posenc.embedding_dim
    keep = [x, posenc.sh_norm.to(dev)]
    feats = torch.empty(B, nf, dtype=torch.float64, device=dev)
    n = len(weights)
    d = L.LocEncDesc()
    d.lonlat, d.B, d.L, d.sh_norm = x.data_ptr(), B, posenc.L, keep[1].data_ptr()
    dims = [nf]
    wp, bp = (L.fp * max(n, 1))(), (L.fp * max(n, 1))()
    for i, (w, b) in enumerate(zip(weights, biases)):
        w64 = w.detach().to(device=dev, dtype=torch.float64).contiguous()
        keep.append(w64)
        wp[i] = w64.data_ptr()
        if b is not None:
            b64 = b.detach().to(device=dev, dtype=torch.float64).contiguous()
            keep.append(b64)
            bp[i] = b64.data_ptr()
        else:
            bp[i] = None
        dims.append(w64.shape[0])
    out = torch.empty(B, dims[-1], dtype=torch.float64, device=dev)
    if n == 0:      # harmonics only: an identity "layer" is not needed, the kernel copies the features out
        eye = torch.eye(nf, dtype=torch.float64, device=dev)
        keep.append(eye)
        wp[0], bp[0] = eye.data_ptr(), None
        dims, w0, n = [nf, nf], [0.0], 1
        out = torch.empty(B, nf, dtype=torch.float64, device=dev)
    d.nlayers = n
    d.weights, d.biases = C.cast(wp, C.POINTER(L.fp)), C.cast(bp, C.POINTER(L.fp))
    d.dims = (L.i32 * len(dims))(*dims)
    d.w0 = (C.c_double * n)(*w0)
    d.out, d.features = out.data_ptr(), feats.data_ptr()
    st = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    L.check(L.backend().nirgan_location_encoder(C.byref(d), st), "location_encoder")
    del keep
    return (out if weights else None), feats


class LocationEncoder(nn.Module):
    def __init__(self, posenc, nnet):
        super().__init__()
        self.posenc, self.nnet = posenc, nnet

    def forward(self, x):
        if self.training and any(l.dropout for l in self.nnet.layers):
            raise NotImplementedError("LocationEncoder runs frozen in evaluation mode on the MI355X path (call .eval()); "
                                      "the reference never trains it (satclip_wrapper.py:30-35)")
        weights, biases, w0 = self.nnet.linear_stack()
        return _run(x, self.posenc, weights, biases, w0)[0]


def get_positional_encoding(name, legendre_polys=10, harmonics_calculation='analytic', min_radius=1, max_radius=360, frequency_num=10):
    if name == "sphericalharmonics" and harmonics_calculation != "discretized":
        return SphericalHarmonics(legendre_polys=legendre_polys, harmonics_calculation=harmonics_calculation)
    if name in ("direct", "cartesian3d", "sphericalharmonics", "theory", "wrap", "grid", "spherec", "spherecplus", "spherem", "spheremplus"):
        raise NotImplementedError(f"positional encoding [{name}/{harmonics_calculation}] is not on the MI355X path "
                                  "(the published SatCLIP checkpoints use spherical harmonics)")
    raise ValueError(f"{name} not a known positional encoding.")


def get_neural_network(name, input_dim, num_classes=256, dim_hidden=256, num_layers=2):
    if name == "siren":
        return SirenNet(dim_in=input_dim, dim_hidden=dim_hidden, num_layers=num_layers, dim_out=num_classes)
    if name in ("linear", "mlp", "fcnet"):
        raise NotImplementedError(f"neural network [{name}] is not on the MI355X path (the published SatCLIP checkpoints use siren)")
    raise ValueError(f"{name} not a known neural networks.")
