"""The bf16 operand mode's 256 x 256 x 64 eight-phase tile (csrc/igemm_tile256.h) on the MI355X: the default launch of an
eligible nirgan_conv_desc against (a) the 128-row tile on the same descriptor (NIRGAN_CONV_TILE128: fp32 summation order only),
(b) torch's conv2d in float64 on the bf16-rounded operands (the reference's nn.Conv2d arithmetic, model/networks.py:405-427, with
the operand rounding of BASELINE.json configs[4]), and -- for the synchronisation structure -- repeated launches that must agree
bit for bit (a fragment read that beats its LDS-DMA shows up as a rare differing tile)."""
import ctypes as C

import pytest
import torch

from nirgan_hip import geometry as G
from nirgan_hip import lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv

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
    (9, 61, 63, 128, 256, 3, True, False),       # ragged last M tile (34 587 rows), 18 K-tiles
    (8, 64, 64, 64, 256, 3, False, False),       # 9 K-tiles: the odd tail (one buffer only in the last round)
    (16, 33, 35, 64, 512, 1, True, False),       # one K-tile: prologue + four phases, two column tiles
    (32, 31, 31, 256, 512, 4, False, False),     # the PatchGAN's stride-1 4x4 layer: 16 taps, 64 K-tiles, two column tiles
    (5, 80, 52, 192, 256, 3, False, True),       # run = 192: three slices per tap
]


@pytest.mark.parametrize("case", CASES)
def test_tile256_against_the_128_row_tile_and_float64(case):
    B, H, W, Cin, Cout, k, bias, out16 = case
    ctx, x, w, b, (y256, y128), (d256, d128) = _problem(*case)
    L.call("nirgan_conv_igemm", C.byref(d256), None)
    L.call("nirgan_conv_igemm", C.byref(d128), None)
    torch.cuda.synchronize()
    a, r = y256.t.float(), y128.t.float()
    assert torch.isfinite(a).all(), "rows or columns left unwritten"
    tol = 2 ** -8 if out16 else 2e-5            # (two bf16 roundings of sums that differ in the last fp32 bits may land one bf16 ulp apart)
    assert _rel(a, r) <= tol, f"256-row tile vs 128-row tile: {_rel(a, r):.3e}"
    if B * H * W <= 40000:
        p = (k - 1) // 2 if k != 4 else 1
        xi = x.t16.double().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi, w.to(torch.bfloat16).double(), None if b is None else b.double()).permute(0, 2, 3, 1)
        if out16:
            ref = ref.float().to(torch.bfloat16)
        assert _rel(a, ref.float()) <= (2 ** -8 if out16 else 1e-5), f"256-row tile vs float64 conv2d on the rounded operands: {_rel(a, ref.float()):.3e}"


def test_tile256_repeated_launches_agree_bitwise():
    """Race screen for the counted-vmcnt / raw-barrier structure: 200 launches each at three sizes, all outputs bitwise equal to the
    first (the summation order is fixed, so any difference is a fragment read that saw stale LDS)."""
    for case in (CASES[0], CASES[2], CASES[5]):
        ctx, x, w, b, (y256, _), (d256, _) = _problem(*case, seed=3)
        L.call("nirgan_conv_igemm", C.byref(d256), None)
        torch.cuda.synchronize()
        first = y256.t.clone()
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        for it in range(200):
            y256.t.fill_(0)
            L.call("nirgan_conv_igemm", C.byref(d256), None)
            bad += (y256.t != first).any().to(torch.int64)          # (device-side: the launches stay back to back)
        assert int(bad.item()) == 0, f"{case}: {int(bad.item())} of 200 launches differ"
