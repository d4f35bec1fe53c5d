"""Descriptor precision 3 (csrc/igemm_x3.h): fp32 operands as three bf16 terms, six bf16 products per fp32 product -- every kernel of the
family against float64 at the tolerance of the exact-fp32 tile it replaces (model/networks.py:349,360-363,405-427,559-574 through
train.py:29: the reference's arithmetic is fp32).  The bound is relative to the exact tile's own error against float64 on the same
operands (<= 2x), plus an absolute floor of a few fp32 ulps of the output's maximum."""
import ctypes as C

import numpy as np
import pytest
import torch

from emu_backend import split3_planes
from nirgan_hip import geometry as G
from nirgan_hip import lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv, emit_wgrad
from nirgan_hip.options import OPT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _split3(ctx, t):
    n = t.numel()
    plane = (n + 7) // 8 * 8
    tw = torch.zeros(3 * plane, dtype=torch.bfloat16, device=DEV)
    L.call("nirgan_split3", t.data_ptr(), tw.data_ptr(), n, plane, None)
    ctx.keep.append(tw)
    return tw, plane


def _err(got, ref):
    e = (got.double() - ref).abs()
    s = ref.abs().max().item()
    return e.max().item() / s, e.pow(2).mean().sqrt().item() / s


def test_split3_planes_are_the_exact_three_term_split():
    """x = h + m + l exactly, each term the nearest-even bf16 of what the previous ones left (the emulator's numpy statement of the rule)"""
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(4096, generator=g), torch.randn(2048, generator=g) * 1e-6, torch.randn(2040, generator=g) * 1e5, torch.zeros(8)]).to(DEV)
    ctx = Ctx(DEV)
    tw, plane = _split3(ctx, x)
    torch.cuda.synchronize()
    bits = tw.view(torch.int16).cpu().numpy().view(np.uint16).reshape(3, plane)[:, :x.numel()]
    want = split3_planes(x.cpu().numpy())
    for t in range(3):
        assert np.array_equal(bits[t], want[t]), f"term {t}"
    back = sum((bits[t].astype(np.uint32) << 16).view(np.float32).astype(np.float64) for t in range(3))
    assert np.array_equal(back, x.cpu().numpy().astype(np.float64))


CONVS = [  # B, H, W, cin, cout, k, stride, bias
    (2, 32, 32, 64, 128, 3, 2, True),          # N = 128, one partly filled M tile
    (3, 31, 31, 128, 256, 3, 1, False),        # odd extent: ragged last tile, two column tiles
    (2, 64, 64, 128, 64, 4, 2, True),          # N = 64 with K = 2048: the 256 x 64 tile
    (16, 64, 64, 256, 256, 3, 1, False),       # the benchmark's trunk shape (direct form), 512 full tiles
    (1, 16, 16, 32, 128, 3, 1, True),          # one K-tile per tap, a single item per workgroup
]


@pytest.mark.parametrize("case", CONVS)
def test_conv_x3_against_float64_at_the_fp32_tiles_error(case):
    B, H, W, cin, cout, k, s, with_bias = case
    g = torch.Generator().manual_seed(11)
    ctx = Ctx(DEV)
    OH, OW = G.conv_out(H, k, s, 1), G.conv_out(W, k, s, 1)
    x = Halo(ctx, B, H, W, cin, 1)
    x.interior().copy_(torch.randn(B, H, W, cin, generator=g).to(DEV))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(DEV)
    bias = (torch.randn(cout, generator=g)).to(DEV) if with_bias else None
    spec = G.conv_fwd_pack(cout, cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    ctx.keep.append(wp)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    tw, plane = _split3(ctx, wp)
    outs = {}
    for prec in (0, 3):
        y = Halo(ctx, B, OH, OW, cout, 0)
        y.t.fill_(float("nan"))
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, bias, y, N=cout, OH=OH, OW=OW, in_stride=s, allow_split=False)
        d.precision = prec
        if prec == 3:
            d.w_x3, d.w_x3_plane = tw.data_ptr(), plane
            assert L.backend().nirgan_conv_kernel_name(C.byref(d)).startswith(b"conv_x3"), "the split tile does not cover this case"
        L.call("nirgan_conv_igemm", C.byref(d), None)
        L.call("nirgan_conv_igemm", C.byref(d), None)          # (a second launch over the same buffers: same bits)
        torch.cuda.synchronize()
        outs[prec] = y.t.clone()
    ref = torch.nn.functional.conv2d(x.t.double().permute(0, 3, 1, 2), w.double(), None if bias is None else bias.double(), stride=s).permute(0, 2, 3, 1)
    e0, e3 = _err(outs[0], ref), _err(outs[3], ref)
    assert torch.isfinite(outs[3]).all()
    assert e3[0] <= 2.0 * e0[0] + 2e-7 and e3[1] <= 2.0 * e0[1] + 5e-8, (e0, e3)


WGRADS = [  # B, H, cin, cout, k, stride
    (2, 64, 64, 128, 3, 2),        # N = 128, K = 576: a partly filled fifth column tile
    (2, 64, 128, 256, 3, 2),       # N = 256: 256-row units (OW = 32: the scalar pixel walk needs whole 32-pixel K-tiles per image row)
    (4, 64, 64, 128, 4, 2),        # K = 1024
]


@pytest.mark.parametrize("case", WGRADS)
def test_wgrad_x3_against_float64_at_the_fp32_tiles_error(case, monkeypatch):
    B, H, cin, cout, k, s = case
    g = torch.Generator().manual_seed(12)
    ctx = Ctx(DEV)
    OH = G.conv_out(H, k, s, 1)
    x = Halo(ctx, B, H, H, cin, 1)
    x.interior().copy_(torch.randn(B, H, H, cin, generator=g).to(DEV))
    dy = Halo(ctx, B, OH, OH, cout, 0)
    dy.t.copy_(torch.randn(B, OH, OH, cout, generator=g).to(DEV))
    spec = G.conv_fwd_pack(cout, cin, k)
    got = {}
    for prec in (0, 3):
        monkeypatch.setattr(OPT, "split3", prec == 3)
        gw = ctx.zeros(cout, cin, k, k)
        ctx.keep.append(gw)
        plan = Plan(ctx)
        d = emit_wgrad(plan, ctx, dy, x, G.conv_fwd_taps(k, cin), spec, gw, N=cout, OH=OH, OW=OH, p_oh=0, p_ow=0, q_stride=s, q_oh=0, q_ow=0)
        assert d.precision == prec
        if prec == 3:
            assert L.backend().nirgan_wgrad_kernel_name(C.byref(d)).startswith(b"wgrad_x3_kernel")
        plan.run()
        torch.cuda.synchronize()
        got[prec] = gw.clone()
    xd, dyd = x.t.double(), dy.t.double().reshape(-1, cout)
    ref = torch.stack([torch.stack([dyd.T @ xd[:, kh:kh + (OH - 1) * s + 1:s, kw:kw + (OH - 1) * s + 1:s, :].reshape(-1, cin) for kw in range(k)], -1)
                       for kh in range(k)], -2)
    e0, e3 = _err(got[0], ref), _err(got[3], ref)
    # (the split tile runs one unit per CU: up to twice the pixels per split of the exact tile's launch, i.e. longer fp32 accumulation
    # chains -- sqrt(2) of its summation noise on top of the products')
    assert e3[0] <= 2.0 * e0[0] + 2e-7 and e3[1] <= 2.5 * e0[1] + 5e-8, (e0, e3)


PLANES = [  # variant, B, H, W, C, K
    (6, 2, 16, 16, 128, 128),      # T = 18: one partly filled tile per plane, a K-tile of 18 pixels in the weight gradient
    (6, 16, 64, 64, 256, 256),     # the benchmark's trunk: T = 1936
    (4, 4, 31, 31, 256, 512),      # the PatchGAN's F(4x4,4x4) layer: 49 planes, K = 512
    (6, 1, 12, 12, 64, 128),       # C = 64: two K-tiles per plane
]


@pytest.mark.parametrize("case", PLANES)
def test_wino6_plane_gemms_and_weight_gradient_x3(case):
    """M[f] = V[f] U[f]^T on the split tile (U from its three planes written by nirgan_wino6_weights_x3) and dU[f] = Yt[f]^T V[f] as
    wgrad_tile_x3 planes, against float64 and against the exact-fp32 launches of the same descriptors."""
    v, B, H, W, Cc, K = case
    r, mo = (3, 6) if v == 6 else (v, 4)
    NP = (mo + r - 1) ** 2
    be = L.backend()
    T = int(be.nirgan_wino6_tiles_r(B, H, W, v))
    g = torch.Generator().manual_seed(13)
    w = (torch.randn(K, Cc, r, r, generator=g) * 0.05).to(DEV)
    U = torch.zeros(NP * K * Cc, device=DEV)
    U3 = torch.zeros(3 * NP * K * Cc, dtype=torch.bfloat16, device=DEV)
    L.call("nirgan_wino6_weights_x3", w.data_ptr(), K, Cc, v, 0, U.data_ptr(), U3.data_ptr(), None)
    U_plain = torch.zeros_like(U)
    L.call("nirgan_wino6_weights_r", w.data_ptr(), K, Cc, v, 0, U_plain.data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.equal(U, U_plain)
    bits = U3.view(torch.int16).cpu().numpy().view(np.uint16).reshape(3, -1)
    back = sum((bits[t].astype(np.uint32) << 16).view(np.float32).astype(np.float64) for t in range(3))
    assert np.array_equal(back, U.cpu().numpy().astype(np.float64)), "U3 is not the three-term split of U"
    # the planes alone (U = NULL: what the engine asks for when the split tile takes the layer) are the same planes
    U3_only = torch.zeros_like(U3)
    L.call("nirgan_wino6_weights_x3", w.data_ptr(), K, Cc, v, 0, None, U3_only.data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.equal(U3_only.view(torch.int16), U3.view(torch.int16))
    V = torch.randn(NP * T * Cc, generator=g).to(DEV)
    zero = torch.zeros(64, device=DEV)
    Ms = {}
    for x3 in (False, True):
        M = torch.full((NP * T * K,), float("nan"), device=DEV)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = v, B, H, W, Cc, K
        d.U, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
        if x3:
            d.U3 = U3.data_ptr()
            assert be.nirgan_wino6_gemm_kernel_name(C.byref(d)).startswith(b"conv_x3")
        L.call("nirgan_wino6_gemm", C.byref(d), None)
        torch.cuda.synchronize()
        Ms[x3] = M
    # ... and the plane GEMM needs nothing else: the same launch without U is bitwise the one with it; without U AND planes it is refused
    M2 = torch.full((NP * T * K,), float("nan"), device=DEV)
    d = L.Wino6Desc()
    d.r, d.B, d.H, d.W, d.C, d.K = v, B, H, W, Cc, K
    d.V, d.V_elems, d.M, d.M_elems, d.zero_page, d.U3 = V.data_ptr(), V.numel(), M2.data_ptr(), M2.numel(), zero.data_ptr(), U3.data_ptr()
    L.call("nirgan_wino6_gemm", C.byref(d), None)
    torch.cuda.synchronize()
    assert torch.equal(M2, Ms[True])
    d.U3 = None
    assert be.nirgan_wino6_gemm(C.byref(d), None) != 0
    ref = torch.einsum("ftc,fkc->ftk", V.double().reshape(NP, T, Cc), U.double().reshape(NP, K, Cc)).reshape(-1)
    e0, e3 = _err(Ms[False], ref), _err(Ms[True], ref)
    assert torch.isfinite(Ms[True]).all()
    assert e3[0] <= 2.0 * e0[0] + 2e-7 and e3[1] <= 2.0 * e0[1] + 5e-8, ("plane GEMM", e0, e3)
    if K % 128:
        return
    # transform-domain weight gradient: P = Yt [NP][T][K], Q = V [NP][T][C] -> slabs [NP][nsplit][K][C]
    Yt = torch.randn(NP * T * K, generator=g).to(DEV)
    slabs = {}
    for prec in (0, 3):
        for nsplit in ((1, 2) if T > 64 else (1,)):
            rows = -(-(-(-T // nsplit)) // 32) * 32
            sl = torch.full((NP * nsplit * K * Cc,), float("nan"), device=DEV)
            d = L.WgradDesc()
            d.p, d.p_elems, d.p_hp, d.p_wp, d.p_cs, d.p_oh, d.p_ow = Yt.data_ptr(), NP * T * K, 1, T, K, 0, 0
            d.q, d.q_elems, d.q_hp, d.q_wp, d.q_cs = V.data_ptr(), NP * T * Cc, 1, T, Cc
            d.q_stride, d.q_oh, d.q_ow, d.run, d.ntaps = 1, 0, 0, Cc, 1
            d.B, d.OH, d.OW, d.N = 1, 1, T, K
            d.slabs, d.slab_elems, d.nsplit, d.rows_per_split = sl.data_ptr(), sl.numel(), nsplit, rows
            d.zero_page, d.precision = zero.data_ptr(), prec
            d.nplanes, d.p_plane, d.q_plane = NP, T * K, T * Cc
            if prec == 3:
                assert be.nirgan_wgrad_kernel_name(C.byref(d)).startswith(b"wgrad_x3_kernel")
            L.call("nirgan_wgrad_igemm", C.byref(d), None)
            torch.cuda.synchronize()
            slabs[(prec, nsplit)] = sl.reshape(NP, nsplit, K, Cc).sum(1)
    refw = torch.einsum("ftk,ftc->fkc", Yt.double().reshape(NP, T, K), V.double().reshape(NP, T, Cc))
    for nsplit in ((1, 2) if T > 64 else (1,)):
        e0, e3 = _err(slabs[(0, nsplit)], refw), _err(slabs[(3, nsplit)], refw)
        assert torch.isfinite(slabs[(3, nsplit)]).all()
        assert e3[0] <= 2.0 * e0[0] + 2e-7 and e3[1] <= 2.0 * e0[1] + 5e-8, ("plane weight gradient", nsplit, e0, e3)


def test_x3_launches_are_bitwise_reproducible_over_many_launches():
    """the persistent walk keeps LDS-DMA and stores in flight across barriers and items: 100 launches of one problem, every output bit equal"""
    g = torch.Generator().manual_seed(14)
    ctx = Ctx(DEV)
    B, H, cin, cout, k, s = 8, 64, 128, 256, 3, 2
    OH = G.conv_out(H, k, s, 1)
    x = Halo(ctx, B, H, H, cin, 1)
    x.interior().copy_(torch.randn(B, H, H, cin, generator=g).to(DEV))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(DEV)
    spec = G.conv_fwd_pack(cout, cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    ctx.keep.append(wp)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    tw, plane = _split3(ctx, wp)
    y = Halo(ctx, B, OH, OH, cout, 0)
    d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, None, y, N=cout, OH=OH, OW=OH, in_stride=s, allow_split=False)
    d.precision, d.w_x3, d.w_x3_plane = 3, tw.data_ptr(), plane
    L.call("nirgan_conv_igemm", C.byref(d), None)
    torch.cuda.synchronize()
    first = y.t.clone()
    for i in range(100):
        y.t.fill_(float("nan"))
        L.call("nirgan_conv_igemm", C.byref(d), None)
        torch.cuda.synchronize()
        assert torch.equal(y.t, first), f"launch {i} differs"


R4_CONVS = [  # B, H, W, cin, cout, k, stride, pad, bias   (>= 192 tiles of 128 columns each: below that the library takes 64-column tiles)
    (16, 128, 128, 64, 128, 3, 2, 1, True),    # 256 full tiles, bias
    (6, 95, 93, 128, 256, 3, 1, 1, False),     # odd extents: a ragged last tile, two column tiles, the row walk wraps inside a lane pair
    (13, 64, 64, 96, 128, 1, 1, 0, False),     # THREE K-tiles per item: the least the cursor supports, an item crossing in every third tile
    (12, 64, 68, 128, 128, 1, 1, 0, True),     # four K-tiles, bias
    (16, 192, 192, 64, 128, 3, 2, 1, False),   # 576 tiles: two or three items per workgroup, an epilogue's stores behind the next item's fetches
    (4, 133, 101, 64, 128, 3, 1, 1, False),    # odd width and height
]


@pytest.mark.parametrize("case", R4_CONVS)
def test_four_wave_split_tile_is_bitwise_the_eight_wave_tile(case):
    """csrc/igemm_x3r.h (descriptor algo NIRGAN_CONV_X3_R4): same operands, same products, same accumulation order per output element as
    conv_x3_persist -- the outputs are compared BITWISE, over three launches each (the persistent walk keeps fetches in flight across items)"""
    B, H, W, cin, cout, k, s, pad, with_bias = case
    g = torch.Generator().manual_seed(21)
    ctx = Ctx(DEV)
    OH, OW = G.conv_out(H, k, s, pad), G.conv_out(W, k, s, pad)
    x = Halo(ctx, B, H, W, cin, pad)
    x.interior().copy_(torch.randn(B, H, W, cin, generator=g).to(DEV))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV) if with_bias else None
    spec = G.conv_fwd_pack(cout, cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    ctx.keep.append(wp)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    tw, plane = _split3(ctx, wp)
    outs = {}
    for algo in (0, L.CONV_X3_R4):
        y = Halo(ctx, B, OH, OW, cout, 0)
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, cin), wp, bias, y, N=cout, OH=OH, OW=OW, in_stride=s, allow_split=False)
        d.precision, d.w_x3, d.w_x3_plane, d.algo = 3, tw.data_ptr(), plane, algo
        name = L.backend().nirgan_conv_kernel_name(C.byref(d))
        assert name.startswith(b"conv_x3r_kernel" if algo else b"conv_x3_kernel"), name
        for _ in range(3):
            y.t.fill_(float("nan"))
            L.call("nirgan_conv_igemm", C.byref(d), None)
            torch.cuda.synchronize()
            if algo in outs:
                assert torch.equal(outs[algo], y.t), "a launch differs from the first one"
            outs[algo] = y.t.clone()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[L.CONV_X3_R4])
    ref = torch.nn.functional.conv2d(x.t.double().permute(0, 3, 1, 2), w.double(), None if bias is None else bias.double(), stride=s).permute(0, 2, 3, 1)
    assert _err(outs[L.CONV_X3_R4], ref)[0] < 2e-6


@pytest.mark.parametrize("shape", [(2, 16, 16, 128, 128), (16, 64, 64, 256, 256), (3, 40, 28, 96, 256)])
def test_four_wave_plane_gemm_is_bitwise_the_eight_wave_plane_gemm(shape):
    """the Winograd plane batches (nirgan_wino6_desc.algo NIRGAN_W6_X3_R4): 64 planes of [T x C] x [C x K]; T = 18 leaves one partly filled
    tile per plane, C = 96 is three K-tiles per item"""
    B, H, W, Cc, K = shape
    g = torch.Generator().manual_seed(22)
    T = B * ((H + 5) // 6) * ((W + 5) // 6)
    V = torch.randn(64 * T * Cc, generator=g).to(DEV)
    U = (torch.randn(64 * K * Cc, generator=g) * 0.05).to(DEV)
    zero = torch.zeros(64, device=DEV)
    plane = 64 * K * Cc
    U3 = torch.zeros(3 * plane, dtype=torch.bfloat16, device=DEV)
    L.call("nirgan_split3", U.data_ptr(), U3.data_ptr(), plane, plane, None)
    outs = {}
    for algo in (0, L.W6_X3_R4):
        M = torch.full((64 * T * K,), float("nan"), device=DEV)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, K
        d.U3, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U3.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
        d.algo = algo
        name = L.backend().nirgan_wino6_gemm_kernel_name(C.byref(d))
        assert name.startswith(b"conv_x3r_kernel" if algo else b"conv_x3_kernel"), name
        for _ in range(2):
            L.call("nirgan_wino6_gemm", C.byref(d), None)
        torch.cuda.synchronize()
        outs[algo] = M
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[L.W6_X3_R4])
    ref = torch.bmm(V.view(64, T, Cc).double(), U.view(64, K, Cc).double().transpose(1, 2)).reshape(-1)
    assert _err(outs[L.W6_X3_R4], ref)[0] < 2e-6
