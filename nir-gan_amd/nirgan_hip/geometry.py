"""Host-side geometry of the implicit-GEMM descriptors (pure Python, no device access).

Every convolution-like layer of the path is lowered to the two generic MFMA kernels of
libnirgan_hip (``nirgan_conv_igemm``, ``nirgan_wgrad_igemm``) over halo'd NHWC buffers.
This module holds the index arithmetic: tap lists, origins, sub-pixel phase decomposition
of stride-2 data-gradients / transposed convolutions, and the int32 maps between the
reference weight layouts (Conv2d ``Cout,Cin,kh,kw`` -- model/networks.py:342,349,405-427,
559-579; ConvTranspose2d ``Cin,Cout,kh,kw`` -- networks.py:360-363) and the packed
``[N][ntaps*run]`` layout the kernels consume.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np


@dataclass
class PackSpec:
    """packed[n][k] = src[n*row_stride + index_map[k]]  (index_map[k] < 0 -> 0)."""
    N: int
    K: int
    row_stride: int
    index_map: np.ndarray            # int32 [K]
    key: tuple = ()
    run: int = 0                     # contiguous floats per tap the consuming contraction reads (0: unknown)


@dataclass
class Taps:
    dh: List[int]
    dw: List[int]
    run: int

    @property
    def n(self):
        return len(self.dh)


def conv_out(h: int, k: int, s: int, p: int) -> int:
    return (h + 2 * p - k) // s + 1


# ------------------------------------------------------------------ Conv2d, weight [Cout][Cin][k][k]
def conv_fwd_taps(k: int, cin: int) -> Taps:
    return Taps([kh for kh in range(k) for _ in range(k)], [kw for _ in range(k) for kw in range(k)], cin)


def conv_fwd_pack(cout: int, cin: int, k: int) -> PackSpec:
    """packed[co][(kh*k+kw)*cin + ci] = W[co][ci][kh][kw]; also the wgrad reduce map."""
    t = np.arange(k * k)[:, None]
    ci = np.arange(cin)[None, :]
    return PackSpec(cout, k * k * cin, cin * k * k, (ci * (k * k) + t).astype(np.int32).reshape(-1), ("cf", cout, cin, k), cin)


def conv_rowpacked_taps(k: int, cs: int) -> Taps:
    """One tap per kernel ROW: k*cs contiguous floats of the NHWC(cs) input cover kw x channels."""
    return Taps(list(range(k)), [0] * k, k * cs)


def conv_rowpacked_pack(cout: int, cin: int, k: int, cs: int) -> PackSpec:
    """packed[co][kh*(k*cs) + kw*cs + c] = W[co][c][kh][kw] for c < cin, else 0."""
    idx = np.full((k, k, cs), -1, dtype=np.int32)
    for kh in range(k):
        for kw in range(k):
            for c in range(cin):
                idx[kh, kw, c] = c * k * k + kh * k + kw
    return PackSpec(cout, k * k * cs, cin * k * k, idx.reshape(-1), ("cr", cout, cin, k, cs), k * cs)


def conv_rowpacked_pair_pack(cout: int, cin: int, k: int, cs: int, q: int) -> PackSpec:
    """Row-packed first convolution with TWO adjacent output pixels per GEMM row (stride 1): a tap is one kernel row over the k + 1
    pixels both windows cover; pixel q of the pair sees W[co][c][kh][kw] at window pixel kw + q.
    packed[co][kh*((k+1)*cs) + px*cs + c] = W[co][c][kh][px - q] for 0 <= px - q < k and c < cin, else 0."""
    idx = np.full((k, k + 1, cs), -1, dtype=np.int32)
    for kh in range(k):
        for kw in range(k):
            for c in range(cin):
                idx[kh, kw + q, c] = c * k * k + kh * k + kw
    return PackSpec(cout, k * (k + 1) * cs, cin * k * k, idx.reshape(-1), ("crp", cout, cin, k, cs, q), (k + 1) * cs)


def conv_dgrad_s1_taps(k: int, cout: int) -> Taps:
    """Full correlation over dY with a zero halo of k-1: dXp[a] = sum_kh Z[a + (k-1-kh)] W[kh]."""
    return Taps([k - 1 - kh for kh in range(k) for _ in range(k)], [k - 1 - kw for _ in range(k) for kw in range(k)], cout)


def conv_dgrad_pack(cout: int, cin: int, k: int, taps_hw: List[Tuple[int, int]]) -> PackSpec:
    """packed[ci][t*cout + co] = W[co][ci][kh_t][kw_t]  (rows are INPUT channels)."""
    idx = np.empty((len(taps_hw), cout), dtype=np.int32)
    for t, (kh, kw) in enumerate(taps_hw):
        idx[t, :] = np.arange(cout) * (cin * k * k) + kh * k + kw
    return PackSpec(cin, len(taps_hw) * cout, k * k, idx.reshape(-1), ("cd", cout, cin, k, tuple(taps_hw)), cout)


@dataclass
class Phase:
    """One sub-pixel phase of a stride-2 gather: out rows oh*2 + out_o, input rows oh + in_o + dh."""
    n_h: int
    n_w: int
    out_oh: int
    out_ow: int
    in_oh: int
    in_ow: int
    taps_hw: List[Tuple[int, int]]        # kernel (kh, kw) of each tap
    dh: List[int] = field(default_factory=list)
    dw: List[int] = field(default_factory=list)


def _phase_axis_dgrad(H: int, k: int, p: int, par: int):
    """Stride-2 conv data-gradient along one axis, for interior rows h with (h+p)%2 == par.

    dXp[a] = sum_{kh = par+2j} dY[(a-kh)/2] W[kh], a = h+p = 2a'+par, dY row a'-j; dY lives in a
    zero-halo-1 buffer (row index +1).  Returns (count, out_origin, in_origin, [(kh, dh)]).
    """
    a_min = -(-(p - par) // 2)                 # ceil((p-par)/2)
    a_max = (H + p - 1 - par) // 2
    cnt = a_max - a_min + 1
    js = [(par + 2 * j, -j) for j in range((k - par + 1) // 2)]
    return cnt, 2 * a_min + par - p, a_min + 1, js


def conv_dgrad_s2_phases(H: int, W: int, k: int, p: int) -> List[Phase]:
    out = []
    for pa in range(2):
        nh, ooh, ioh, jh = _phase_axis_dgrad(H, k, p, pa)
        for pb in range(2):
            nw, oow, iow, jw = _phase_axis_dgrad(W, k, p, pb)
            if nh <= 0 or nw <= 0 or not jh or not jw:
                continue
            ph = Phase(nh, nw, ooh, oow, ioh, iow, [(kh, kw) for kh, _ in jh for kw, _ in jw])
            ph.dh = [d for _, d in jh for _ in jw]
            ph.dw = [d for _ in jh for _, d in jw]
            out.append(ph)
    return out


# ------------------------------------------------------------------ ConvTranspose2d k3 s2 p1 op1, weight [Cin][Cout][k][k]
def _phase_axis_convT(Hin: int, k: int, p: int, par: int):
    """y[oy] = sum_{iy,kh: oy = 2 iy - p + kh} x[iy] W[kh]; oy = 2a'+par, kh = kh0+2j, iy = a'+d0-j.

    x lives in a zero-halo-1 buffer (row index +1).
    """
    kh0 = (par + p) % 2
    d0 = (par + p - kh0) // 2
    js = [(kh0 + 2 * j, -j) for j in range((k - kh0 + 1) // 2)]
    return Hin, par, d0 + 1, js


def convT_fwd_phases(Hin: int, Win: int, k: int = 3, p: int = 1) -> List[Phase]:
    out = []
    for pa in range(2):
        nh, ooh, ioh, jh = _phase_axis_convT(Hin, k, p, pa)
        for pb in range(2):
            nw, oow, iow, jw = _phase_axis_convT(Win, k, p, pb)
            ph = Phase(nh, nw, ooh, oow, ioh, iow, [(kh, kw) for kh, _ in jh for kw, _ in jw])
            ph.dh = [d for _, d in jh for _ in jw]
            ph.dw = [d for _ in jh for _, d in jw]
            out.append(ph)
    return out


@dataclass
class PhasePair:
    """Two sub-pixel phases of ONE output row (pixels 2 ow + out_ow and 2 ow + out_ow + 1) as one problem over the union of their taps:
    GEMM row (oh, ow) reads input rows in_oh + oh + dh[t], columns in_ow + ow + dw[t]; taps_hw[q][t] = the kernel element pixel q
    multiplies tap t with, or None (a zero block of the packed weights)."""
    n_h: int
    n_w: int
    out_oh: int
    out_ow: int
    in_oh: int
    in_ow: int
    dh: List[int]
    dw: List[int]
    taps_hw: List[list]


def pair_row_phases(phases: List[Phase]):
    """The four phases of a stride-2 gather as two PhasePairs (one per output-row parity, the shorter tap list first), or None when
    they do not pair up (odd sizes drop or shrink a phase)."""
    if len(phases) != 4:
        return None
    rows = {}
    for ph in phases:
        rows.setdefault(ph.out_oh, []).append(ph)
    if len(rows) != 2 or any(len(v) != 2 for v in rows.values()):
        return None
    out = []
    for ooh in sorted(rows):
        a, b = sorted(rows[ooh], key=lambda ph: ph.out_ow)
        if b.out_ow != a.out_ow + 1 or (a.n_h, a.n_w) != (b.n_h, b.n_w):
            return None
        pos = sorted({(ph.in_oh + h, ph.in_ow + w) for ph in (a, b) for h, w in zip(ph.dh, ph.dw)})
        ioh, iow = min(h for h, _ in pos), min(w for _, w in pos)
        maps = []
        for ph in (a, b):
            own = {(ph.in_oh + h, ph.in_ow + w): hw for h, w, hw in zip(ph.dh, ph.dw, ph.taps_hw)}
            maps.append([own.get(q) for q in pos])
        out.append(PhasePair(a.n_h, a.n_w, a.out_oh, a.out_ow, ioh, iow, [h - ioh for h, _ in pos], [w - iow for _, w in pos], maps))
    out.sort(key=lambda pr: len(pr.dh))
    return out


def masked_pack(spec_fn, taps_hw: list) -> PackSpec:
    """spec_fn(list of (kh, kw)) -> PackSpec with one run per tap; the taps given as None become zero blocks (index -1)."""
    spec = spec_fn([hw if hw is not None else (0, 0) for hw in taps_hw])
    idx = spec.index_map.reshape(len(taps_hw), -1).copy()
    for t, hw in enumerate(taps_hw):
        if hw is None:
            idx[t, :] = -1
    return PackSpec(spec.N, spec.K, spec.row_stride, idx.reshape(-1), spec.key + ("mask", tuple(hw is not None for hw in taps_hw)), spec.run)


def convT_fwd_pack(cin: int, cout: int, k: int, taps_hw: List[Tuple[int, int]]) -> PackSpec:
    """packed[co][t*cin + ci] = W[ci][co][kh_t][kw_t]."""
    idx = np.empty((len(taps_hw), cin), dtype=np.int32)
    for t, (kh, kw) in enumerate(taps_hw):
        idx[t, :] = np.arange(cin) * (cout * k * k) + kh * k + kw
    return PackSpec(cout, len(taps_hw) * cin, k * k, idx.reshape(-1), ("tf", cin, cout, k, tuple(taps_hw)), cin)


def convT_dgrad_taps(k: int, cout: int) -> Taps:
    """dX[iy] = sum_kh dY[2 iy - p + kh] W[kh]: a stride-2 gather over dY (zero halo 1 when p = 1)."""
    return Taps([kh for kh in range(k) for _ in range(k)], [kw for _ in range(k) for kw in range(k)], cout)


def convT_dgrad_pack(cin: int, cout: int, k: int) -> PackSpec:
    """packed[ci][(kh*k+kw)*cout + co] = W[ci][co][kh][kw]; also the convT wgrad reduce map."""
    t = np.arange(k * k)[:, None]
    co = np.arange(cout)[None, :]
    return PackSpec(cin, k * k * cout, cout * k * k, (co * (k * k) + t).astype(np.int32).reshape(-1), ("td", cin, cout, k), cout)


# ------------------------------------------------------------------ Conv2d(C, 1, k): tap planes
def tapplane_fwd_pack(cin: int, k: int) -> PackSpec:
    """packed[t][ci] = W[0][ci][kh][kw], t = kh*k+kw; also the wgrad reduce map."""
    return PackSpec(k * k, cin, 1, (np.arange(cin) * (k * k)).astype(np.int32), ("pf", cin, k), cin)


def tapplane_dgrad_pack(cin: int, k: int, qcs: int) -> PackSpec:
    """packed[ci][t] = W[0][ci][t] for t < k*k, 0 for the padding planes."""
    idx = np.full(qcs, -1, dtype=np.int32)
    idx[:k * k] = np.arange(k * k)
    return PackSpec(cin, qcs, k * k, idx, ("pd", cin, k, qcs), qcs)


def linear_pack(nout: int, nin: int) -> PackSpec:
    return PackSpec(nout, nin, nin, np.arange(nin, dtype=np.int32), ("ln", nout, nin), nin)


def wgrad_split(M: int, tiles: int, target_blocks: int = 1024, row_mult: int = 32) -> Tuple[int, int]:
    """(nsplit, rows_per_split) for the weight-gradient GEMM.

    256 CUs x 2 resident blocks = 512 slots: tiles*nsplit is kept AT OR BELOW a whole number of rounds
    (<= target_blocks) so that no nearly-empty trailing round appears; rows are a multiple of 32."""
    want = max(1, target_blocks // max(tiles, 1))
    rows = max(row_mult, -(-M // want))
    rows = -(-rows // row_mult) * row_mult           # whole K-steps per split (64 pixels when both operands are bf16 twins)
    nsplit = -(-M // rows)
    return nsplit, rows


CUS = 256          # MI355X; engine.Ctx overwrites it with the device's own count: the persistent launches start one workgroup per CU
                   # (csrc/igemm_wgrad.hip::ng_cu_count, csrc/igemm_conv.hip::ng_cu_count_conv read the same property)
# cost figures of the 256-wide items in tenths of a convolution K-tile (csrc/igemm_wgrad.hip::pair256_split uses the same ones; measured,
# profiles/r04_tile256_stamps.txt): convolution K-tile 10 (2 500 cycles) + 65 per tile, weight-gradient K-tile 12 (3 010 cycles) + 40 per unit
T256_CONV_KT, T256_CONV_FIXED, T256_WGRAD_KT, T256_WGRAD_FIXED = 10, 65, 12, 40


def wgrad256_ok(M: int, OH: int, OW: int, N: int, K: int, run: int) -> bool:
    """The shapes csrc/igemm_tile256.h::wgrad_tile256_ok accepts (both operands bf16 twins is the caller's business)."""
    return (N % 256 == 0 and K % 256 == 0 and run % 8 == 0 and (OW % 64 == 0 or (OW <= 64 and 64 % OW == 0))
            and (OH * OW) % 64 == 0 and M % 64 == 0)


def pair256_plan(M: int, units_per_split: int, conv_tiles: int = 0, conv_nk: int = 0, cus: int = 0) -> Tuple[int, int]:
    """(nsplit, rows_per_split) of a weight gradient on the 256-wide persistent tiles, alone (conv_tiles = 0) or fused with the data
    gradient's conv_tiles tiles of conv_nk K-tiles each: the split count that minimises the longer walk -- data-gradient workgroups
    ceil(conv_tiles / x) items, weight-gradient workgroups ceil(units / (cus - x)) units of M / 64 / nsplit K-tiles, at the measured cost
    figures above -- and, among equals, the fewest slabs."""
    kt = M // 64
    cus = cus or CUS
    best = None
    for ns in range(1, min(kt, 64) + 1):
        per = -(-kt // ns)
        if (ns - 1) * per >= kt:
            continue                                  # an empty last split
        units = units_per_split * ns
        wcost = per * T256_WGRAD_KT + T256_WGRAD_FIXED
        if conv_tiles:
            cost = min(max(-(-conv_tiles // x) * (conv_nk * T256_CONV_KT + T256_CONV_FIXED), -(-units // (cus - x)) * wcost) for x in range(1, cus))
        else:
            cost = -(-units // cus) * wcost
        if best is None or cost < best[0]:
            best = (cost, ns, per * 64)
    return best[1], best[2]
