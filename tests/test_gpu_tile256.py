"""The bf16 operand mode's 256 x 256 x 64 eight-phase tile (csrc/igemm_tile256.h) on the MI355X: the default launch of an
eligible nirgan_conv_desc against (a) the 128-row tile on the same descriptor (NIRGAN_CONV_TILE128: fp32 summation order only),
(b) torch's conv2d in float64 on the bf16-rounded operands (the reference's nn.Conv2d arithmetic, model/networks.py:405-427, with
the operand rounding of BASELINE.json configs[4]), and -- for the synchronisation structure -- repeated launches that must agree
bit for bit (a fragment read that beats its LDS-DMA shows up as a rare differing tile)."""
import ctypes as C

import os

import pytest
import torch

from nirgan_hip import geometry as G
from nirgan_hip import lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv

SOAK = int(os.environ.get("NIRGAN_TEST_SOAK", "1"))      # multiplies the launch counts of the race screens (a soak run: 25)
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _packed16(ctx, w, spec):
    buf = torch.zeros(spec.N, spec.K, dtype=torch.bfloat16, device=DEV)
    L.call("nirgan_pack_rows_bf16", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), buf.data_ptr(), spec.N, spec.K, None)
    torch.cuda.synchronize()
    ctx.keep.append(buf)
    return buf


def _problem(B, H, W, Cin, Cout, k, bias, out16, seed=0):
    ctx = Ctx(DEV, "bf16")
    g = torch.Generator().manual_seed(seed)
    p = (k - 1) // 2 if k != 4 else 1
    x = Halo(ctx, B, H, W, Cin, p, twin=True)
    x.t.copy_(torch.randn(x.t.shape, generator=g).to(DEV))
    x.t16.copy_(x.t.to(torch.bfloat16))
    OH, OW = H + 2 * p - k + 1, W + 2 * p - k + 1
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV) if bias else None
    wp = _packed16(ctx, w, G.conv_fwd_pack(Cout, Cin, k))
    outs, descs = [], []
    for algo in (0, L.CONV_TILE128):
        y = Halo(ctx, B, OH, OW, Cout, 0, bf16=out16)
        y.t.fill_(float("nan"))
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, Cin), wp, b, y, N=Cout, OH=OH, OW=OW, allow_split=False)
        assert d.in_bf16 == 1 and d.w_bf16 == 1 and d.out_bf16 == (1 if out16 else 0)
        d.algo = algo
        outs.append(y)
        descs.append(d)
    return ctx, x, w, b, outs, descs


CASES = [
    # B, H, W, Cin, Cout, k, bias, bf16 output
    (16, 64, 64, 256, 256, 3, False, False),     # the benchmark's residual-block convolution: 256 tiles, 36 K-tiles
    (16, 64, 64, 256, 256, 3, True, True),
    (11, 61, 63, 128, 256, 3, True, False),      # ragged last M tile (42 273 rows: 166 tiles), 18 K-tiles
    (10, 64, 64, 64, 256, 3, False, False),      # 9 K-tiles: the odd tail (one buffer only in the last round)
    (18, 33, 35, 64, 512, 1, True, False),       # one K-tile: prologue + four phases, two column tiles (164 tiles)
    (32, 31, 31, 256, 512, 4, False, False),     # the PatchGAN's stride-1 4x4 layer: 16 taps, 64 K-tiles, two column tiles (242 tiles)
    (10, 80, 52, 192, 256, 3, False, True),      # run = 192: three slices per tap (163 tiles)
]


@pytest.mark.parametrize("case", CASES)
def test_tile256_against_the_128_row_tile_and_float64(case):
    B, H, W, Cin, Cout, k, bias, out16 = case
    ctx, x, w, b, (y256, y128), (d256, d128) = _problem(*case)
    assert L.backend().nirgan_conv_kernel_name(C.byref(d256)).decode() == "conv_igemm256_kernel"
    assert L.backend().nirgan_conv_kernel_name(C.byref(d128)).decode() == "conv_igemm_kernel<128>"
    L.call("nirgan_conv_igemm", C.byref(d256), None)
    L.call("nirgan_conv_igemm", C.byref(d128), None)
    torch.cuda.synchronize()
    a, r = y256.t.float(), y128.t.float()
    assert torch.isfinite(a).all(), "rows or columns left unwritten"
    tol = 2 ** -7 if out16 else 2e-5            # (two bf16 roundings of sums that differ in the last fp32 bits may land one bf16 ulp apart)
    assert _rel(a, r) <= tol, f"256-row tile vs 128-row tile: {_rel(a, r):.3e}"
    if B * H * W <= 45000:
        p = (k - 1) // 2 if k != 4 else 1
        xi = x.t16.double().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi, w.to(torch.bfloat16).double(), None if b is None else b.double()).permute(0, 2, 3, 1)
        if out16:
            ref = ref.float().to(torch.bfloat16)
        assert _rel(a, ref.float()) <= (2 ** -7 if out16 else 1e-5), f"256-row tile vs float64 conv2d on the rounded operands: {_rel(a, ref.float()):.3e}"
    else:
        # the benchmark's own shapes: float64 on a random sample of 8 192 output pixels (every channel) instead of the whole tensor
        OH, OW = a.shape[1], a.shape[2]
        g = torch.Generator().manual_seed(17)
        idx = torch.randint(0, B * OH * OW, (8192,), generator=g).to(DEV)
        bi, r = idx // (OH * OW), idx % (OH * OW)
        oh, ow = r // OW, r % OW
        xp = x.t16.double()                                              # [B][H + 2p][W + 2p][Cin]: the halo is part of the buffer
        patches = torch.stack([xp[bi, oh + kh, ow + kw, :] for kh in range(k) for kw in range(k)], 1)
        wk = w.to(torch.bfloat16).double().permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)
        ref = torch.einsum("ntc,otc->no", patches, wk) + (0.0 if b is None else b.double())
        if out16:
            ref = ref.float().to(torch.bfloat16)
        got = a[bi, oh, ow, :]
        assert _rel(got, ref.float()) <= (2 ** -7 if out16 else 1e-5), f"256-row tile vs float64 on 8 192 sampled pixels: {_rel(got, ref.float()):.3e}"


def test_tile256_repeated_launches_agree_bitwise():
    """Race screen for the counted-vmcnt / raw-barrier structure: 200 launches each at three sizes, all outputs bitwise equal to the
    first (the summation order is fixed, so any difference is a fragment read that saw stale LDS)."""
    for case in (CASES[0], CASES[2], CASES[5]):
        ctx, x, w, b, (y256, _), (d256, _) = _problem(*case, seed=3)
        L.call("nirgan_conv_igemm", C.byref(d256), None)
        torch.cuda.synchronize()
        first = y256.t.clone()
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        for it in range(200 * SOAK):
            y256.t.fill_(0)
            L.call("nirgan_conv_igemm", C.byref(d256), None)
            bad += (y256.t != first).any().to(torch.int64)          # (device-side: the launches stay back to back)
        assert int(bad.item()) == 0, f"{case}: {int(bad.item())} of {200 * SOAK} launches differ"


# ------------------------------------------------------------------ weight gradient and the fused launch on the 256-wide tiles
def _wgrad_problem(B, H, W, Cin, Cout, k, seed=0):
    """dY (zero halo k - 1) and X (halo p) as bf16 twins, the convolution's weight-gradient descriptors for the default kernels and for
    the 128-row ones, each with its own split and gradient tensor."""
    from nirgan_hip.engine import emit_wgrad
    ctx = Ctx(DEV, "bf16")
    g = torch.Generator().manual_seed(seed)
    p = (k - 1) // 2 if k != 4 else 1
    OH, OW = H + 2 * p - k + 1, W + 2 * p - k + 1
    x = Halo(ctx, B, H, W, Cin, p, twin=True)
    x.t.copy_(torch.randn(x.t.shape, generator=g).to(DEV))
    x.t16.copy_(x.t.to(torch.bfloat16))
    dy = Halo(ctx, B, OH, OW, Cout, k - 1, twin=True)
    dy.interior().copy_(torch.randn(B, OH, OW, Cout, generator=g).to(DEV))
    dy.t16.copy_(dy.t.to(torch.bfloat16))
    return ctx, x, dy, (OH, OW, p)


def _emit_wgrad(ctx, x, dy, geo, Cin, Cout, k, tile128, pair_with=None):
    from nirgan_hip.engine import emit_wgrad
    from nirgan_hip.options import OPT
    OH, OW, p = geo
    gw = ctx.zeros(Cout, Cin, k, k)
    plan = Plan(ctx)
    OPT.tile256 = not tile128
    try:
        d = emit_wgrad(plan, ctx, dy, x, G.conv_fwd_taps(k, Cin), G.conv_fwd_pack(Cout, Cin, k), gw, N=Cout, OH=OH, OW=OW,
                       p_oh=dy.pad, p_ow=dy.pad, q_oh=x.pad - p, q_ow=x.pad - p, pair_with=pair_with)
    finally:
        OPT.reset()
    assert d.pq_bf16 == 1
    return plan, d, gw


WCASES = [
    # B, H, W, Cin, Cout, k
    (16, 64, 64, 256, 256, 3),       # the benchmark's residual-block layer: 9 column tiles x the planned splits
    (4, 128, 128, 256, 256, 3),      # the 512-pixel bucket's maps (OW = 128: two K-tiles per image row)
    (8, 32, 32, 256, 256, 3),        # OW = 32: a K-tile is two image rows
    (6, 64, 64, 128, 256, 3),        # run = 128: a column tile spans two taps (and half of the ninth)  -> K = 1152, not a multiple of 256: 128-row tiles
    (3, 64, 64, 256, 512, 1),        # two row tiles, one tap
]


def _wgrad_float64(x, dy, OH, OW, k):
    """dW[co][ci][kh][kw] of a stride-1 convolution from the bf16 twins, in float64: one [Cout x M] x [M x Cin] product per tap"""
    xd = x.t16.double()
    dyd = dy.t16.double()[:, dy.pad:dy.pad + OH, dy.pad:dy.pad + OW].reshape(-1, dy.C)
    return torch.stack([torch.stack([dyd.T @ xd[:, kh:kh + OH, kw:kw + OW, :].reshape(-1, x.C) for kw in range(k)], -1) for kh in range(k)], -2)


@pytest.mark.parametrize("case", WCASES)
def test_wgrad256_against_the_128_row_tile_and_float64(case):
    B, H, W, Cin, Cout, k = case
    ctx, x, dy, geo = _wgrad_problem(*case)
    p256, d256, g256 = _emit_wgrad(ctx, x, dy, geo, Cin, Cout, k, False)
    p128, d128, g128 = _emit_wgrad(ctx, x, dy, geo, Cin, Cout, k, True)
    eligible = (9 if k == 3 else 1) * Cin % 256 == 0
    assert (d256.rows_per_split % 64 == 0) and d128.algo == L.WGRAD_TILE128 and d256.algo == 0
    p256.run()
    p128.run()
    torch.cuda.synchronize()
    assert _rel(g256, g128) <= 2e-5, f"256-wide tile vs 128-row tile: {_rel(g256, g128):.3e}"
    if eligible and case == WCASES[0]:
        assert (d256.nsplit, d256.rows_per_split) == G.pair256_plan(B * H * W, 9), "the 256-wide plan chooses its own split"
    OH, OW, p = geo
    if B * H * W <= 40000:
        xi = x.t16.double().permute(0, 3, 1, 2)
        gi = dy.t16.double()[:, dy.pad:dy.pad + OH, dy.pad:dy.pad + OW].permute(0, 3, 1, 2)
        ref = torch.nn.grad.conv2d_weight(xi, (Cout, Cin, k, k), gi)
    else:
        ref = _wgrad_float64(x, dy, OH, OW, k)                 # the benchmark's shapes too: one float64 GEMM per tap
    assert _rel(g256, ref.float()) <= 1e-5, f"256-wide tile vs float64 on the rounded operands: {_rel(g256, ref.float()):.3e}"


@pytest.mark.parametrize("case", [(16, 64, 64, 256, 256, 3, False), (16, 64, 64, 256, 256, 3, True), (32, 32, 32, 256, 256, 3, True), (5, 64, 128, 256, 256, 3, False)])
def test_pair256_equals_separate_launches(case):
    """nirgan_conv_wgrad_pair on the persistent 256-wide tiles (data-gradient tiles and weight-gradient units on disjoint sets of CUs) against
    the same two descriptors launched on the 128-row tiles: data gradient over the padded extent (fp32 or bf16 store) and the reduced
    weight gradient."""
    from nirgan_hip.options import OPT
    B, H, W, Cin, Cout, k, g16 = case
    ctx, x, dy, geo = _wgrad_problem(B, H, W, Cin, Cout, k, seed=1)
    w = (torch.randn(Cout, Cin, k, k, generator=torch.Generator().manual_seed(2)) * 0.05).to(DEV)
    hw = [(a, b) for a in range(k) for b in range(k)]
    wd = _packed16(ctx, w, G.conv_dgrad_pack(Cout, Cin, k, hw))
    res = []
    for tile128 in (False, True):
        gx = Halo(ctx, B, H, W, Cin, 1, bf16=g16)
        gx.t.fill_(float("nan"))
        OPT.tile256 = not tile128
        try:
            cd = emit_conv(None, ctx, dy, G.conv_dgrad_s1_taps(k, Cout), wd, None, gx, N=Cin, OH=gx.hp, OW=gx.wp)
        finally:
            OPT.reset()
        plan, d, gw = _emit_wgrad(ctx, x, dy, geo, Cin, Cout, k, tile128, pair_with=cd)
        assert [n for n, _ in plan.ops] == ["nirgan_conv_wgrad_pair", "nirgan_reduce_rows"]
        assert L.backend().nirgan_conv_wgrad_pair_kernel_name(C.byref(cd), C.byref(d)).decode() == ("conv_wgrad_pair_kernel" if tile128 else "conv_wgrad_pair256_kernel")
        plan.run()
        res.append((gx, gw, d))
    torch.cuda.synchronize()
    (gx256, gw256, d256), (gx128, gw128, d128) = res
    assert torch.isfinite(gx256.t.float()).all()
    assert _rel(gx256.t.float(), gx128.t.float()) <= (2 ** -7 if g16 else 2e-5)
    assert _rel(gw256, gw128) <= 2e-5
    xi = x.t16.double().permute(0, 3, 1, 2)
    OH, OW, p = geo
    if B * H * W <= 45000:
        gi = dy.t16.double()[:, dy.pad:dy.pad + OH, dy.pad:dy.pad + OW].permute(0, 3, 1, 2)
        assert _rel(gw256, torch.nn.grad.conv2d_weight(xi, (Cout, Cin, k, k), gi).float()) <= 1e-5
    else:
        assert _rel(gw256, _wgrad_float64(x, dy, OH, OW, k).float()) <= 1e-5, "fused launch's weight gradient vs float64 at the benchmark shape"


def test_pair256_repeated_launches_agree_bitwise():
    """Race screen of the persistent fused launch (two items per data-gradient workgroup, the barrier between items, the transposing
    reads issued from inline asm): 150 launches, data gradient and slabs bitwise equal to the first."""
    from nirgan_hip.options import OPT
    B, H, W, Cin, Cout, k = 16, 64, 64, 256, 256, 3
    ctx, x, dy, geo = _wgrad_problem(B, H, W, Cin, Cout, k, seed=5)
    w = (torch.randn(Cout, Cin, k, k, generator=torch.Generator().manual_seed(2)) * 0.05).to(DEV)
    wd = _packed16(ctx, w, G.conv_dgrad_pack(Cout, Cin, k, [(a, b) for a in range(k) for b in range(k)]))
    gx = Halo(ctx, B, H, W, Cin, 1)
    cd = emit_conv(None, ctx, dy, G.conv_dgrad_s1_taps(k, Cout), wd, None, gx, N=Cin, OH=gx.hp, OW=gx.wp)
    plan, d, gw = _emit_wgrad(ctx, x, dy, geo, Cin, Cout, k, False, pair_with=cd)
    plan.run()
    torch.cuda.synchronize()
    first_x, first_w = gx.t.clone(), gw.clone()
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for it in range(150 * SOAK):
        gx.t.fill_(0)
        plan.run()
        bad += (gx.t != first_x).any().to(torch.int64) + (gw != first_w).any().to(torch.int64)
    assert int(bad.item()) == 0, f"{int(bad.item())} differing results in {150 * SOAK} launches"


# ------------------------------------------------------------------ the same tile in exact fp32 (v_mfma_f32_32x32x2_f32)
def _problem_f32(B, H, W, Cin, Cout, k, stride, bias, seed=0):
    ctx = Ctx(DEV, "fp32")
    g = torch.Generator().manual_seed(seed)
    p = (k - 1) // 2 if k != 4 else 1
    x = Halo(ctx, B, H, W, Cin, p)
    x.t.copy_(torch.randn(x.t.shape, generator=g).to(DEV))
    OH, OW = G.conv_out(H, k, stride, p), G.conv_out(W, k, stride, p)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV) if bias else None
    spec = G.conv_fwd_pack(Cout, Cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    ctx.keep.append(wp)          # (the descriptors hold its ADDRESS only: without this the caching allocator hands the block to the next clone())
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    torch.cuda.synchronize()
    outs, descs = [], []
    for algo in (L.CONV_TILE256, 0):
        y = Halo(ctx, B, OH, OW, Cout, 0)
        y.t.fill_(float("nan"))
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, Cin), wp, b, y, N=Cout, OH=OH, OW=OW, in_stride=stride, allow_split=False)
        d.algo = algo
        outs.append(y)
        descs.append(d)
    return ctx, x, w, b, outs, descs


F32_CASES = [
    # B, H, W, Cin, Cout, k, stride, bias
    (16, 128, 128, 128, 256, 3, 2, True),        # the generator's second down-sampling convolution at the benchmark batch
    (12, 64, 64, 256, 256, 3, 1, False),         # a residual-block shape on the direct path (OPT.winograd = 'off')
    (11, 61, 63, 96, 256, 3, 1, True),           # ragged last M tile, run = 96: three 32-k slices per tap
    (18, 33, 35, 32, 512, 1, 1, True),           # one K-tile, two column tiles
    (32, 32, 32, 256, 512, 4, 1, False),         # 4x4: 16 taps x 8 slices = 128 K-tiles
]


@pytest.mark.parametrize("case", F32_CASES)
def test_tile256_fp32_against_the_128_row_tile_and_float64(case):
    """Exact-fp32 mode, NIRGAN_CONV_TILE256 (the default keeps the 128-row tile there): the 256-wide tile against the 128-row tile (fp32 summation order only: both are exact fp32 FMA
    chains) and against torch's conv2d in float64 (the reference's nn.Conv2d arithmetic, model/networks.py:349,405-427)."""
    B, H, W, Cin, Cout, k, stride, bias = case
    ctx, x, w, b, (y256, y128), (d256, d128) = _problem_f32(*case)
    assert L.backend().nirgan_conv_kernel_name(C.byref(d256)).decode() == "conv_igemm256_kernel<fp32>"
    assert L.backend().nirgan_conv_kernel_name(C.byref(d128)).decode() == "conv_igemm_kernel<128>"
    L.call("nirgan_conv_igemm", C.byref(d256), None)
    L.call("nirgan_conv_igemm", C.byref(d128), None)
    torch.cuda.synchronize()
    assert torch.isfinite(y256.t).all(), "rows or columns left unwritten"
    assert _rel(y256.t, y128.t) <= 1e-5, f"256-wide tile vs 128-row tile: {_rel(y256.t, y128.t):.3e}"
    if B * H * W <= 70000:
        ref = torch.nn.functional.conv2d(x.t.double().permute(0, 3, 1, 2), w.double(), None if b is None else b.double(), stride=stride).permute(0, 2, 3, 1)
        assert _rel(y256.t, ref.float()) <= 1e-5, f"256-wide tile vs float64 conv2d: {_rel(y256.t, ref.float()):.3e}"


def test_tile256_fp32_repeated_launches_agree_bitwise():
    for case in (F32_CASES[0], F32_CASES[2]):
        ctx, x, w, b, (y256, _), (d256, _) = _problem_f32(*case, seed=3)
        L.call("nirgan_conv_igemm", C.byref(d256), None)
        torch.cuda.synchronize()
        first = y256.t.clone()
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        for it in range(100 * SOAK):
            y256.t.fill_(0)
            L.call("nirgan_conv_igemm", C.byref(d256), None)
            bad += (y256.t != first).any().to(torch.int64)
        assert int(bad.item()) == 0, f"{case}: {int(bad.item())} of {100 * SOAK} launches differ"
