"""Per-kernel parity on the MI355X: every C-ABI entry point is run on device buffers and
compared with the numpy restatement of the same descriptor (tests/emu_backend.py, test
infrastructure) on host copies.  fp32 MFMA results are exact-fp32 FMA chains in a different
order than the fp64 restatement: tolerance 1e-5 relative to the output's max-abs (1e-4 for
the long reductions of the weight gradient), far inside the 1e-3 the path is specified to.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from emu_backend import EmuBackend
from nirgan_hip import geometry as G
from nirgan_hip import lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv, emit_in_bwd, emit_in_fwd, emit_wgrad

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, tol, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what + ": non-finite"
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= tol * max(ref, 1e-20), f"{what}: err {err:.3e} ref {ref:.3e}"


class Twin:
    """The same buffers on the device and on the host; descriptors are emitted against either."""

    def __init__(self, precision="fp32"):
        self.gctx = Ctx(DEV, precision)
        self.emu = EmuBackend()
        L.set_backend(self.emu)
        try:
            self.cctx = Ctx("cpu", precision)
        finally:
            L.set_backend(None)
        self.pairs = []

    def halo(self, B, H, W, Cc, pad, gen=None, fill_halo=True):
        g = Halo(self.gctx, B, H, W, Cc, pad)
        c = Halo(self.cctx, B, H, W, Cc, pad)
        if gen is not None:
            data = torch.randn(c.t.shape, generator=gen)
            if not fill_halo and pad:
                m = torch.zeros_like(data)
                m[:, pad:pad + H, pad:pad + W] = 1
                data = data * m
            c.t.copy_(data)
            g.t.copy_(data)
        self.pairs.append((g.t, c.t))
        return g, c

    def tensor(self, *shape, gen=None, scale=1.0):
        c = torch.zeros(*shape)
        if gen is not None:
            c.copy_(torch.randn(*shape, generator=gen) * scale)
        g = c.to(DEV)
        self.pairs.append((g, c))
        return g, c

    def run(self, gplan: Plan, cplan: Plan):
        gplan.run()
        torch.cuda.synchronize()
        L.set_backend(self.emu)
        try:
            cplan.run()
        finally:
            L.set_backend(None)


from conv_cases import CONV_CASES, build_conv_case


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward_wgrad_dgrad(case, precision):
    """Device kernels against the numpy restatement of the same descriptors, in every operand precision: products of
    bf16 values are exact in fp32, so the bf16 modes differ from the restatement only by summation order, like fp32."""
    _, B, H, W, Cin, Cout, k, s, p = case
    gen = torch.Generator().manual_seed(11)
    tw = Twin(precision)
    xg, xc = tw.halo(B, H, W, Cin, p, gen)
    wg, wc = tw.tensor(Cout, Cin, k, k, gen=gen, scale=0.05)
    bg, bc = tw.tensor(Cout, gen=gen)
    (gp, gp2, gy, gdy, ggw, ggx) = build_conv_case(tw.gctx, xg, wg, bg, case)
    (cp, cp2, cy, cdy, cgw, cgx) = build_conv_case(tw.cctx, xc, wc, bc, case)
    tw.run(gp, cp)
    close(gy.t, cy.t, 1e-5, "conv fwd")
    if precision != "bf16":
        # also against torch's own convolution (the reference's nn.Conv2d arithmetic)
        ref = torch.nn.functional.conv2d(xc.t.permute(0, 3, 1, 2), wc, bc, stride=s)
        close(gy.t.permute(0, 3, 1, 2), ref, 1e-4, "conv fwd vs torch")
    cdy.interior().copy_(cy.t)
    gdy.interior().copy_(cy.t.to(DEV))
    tw.run(gp2, cp2)
    close(ggw, cgw, 1e-4, "wgrad")
    close(ggx.t, cgx.t, 1e-5, "dgrad")


def test_convT_phases_match_torch():
    gen = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout, k = 2, 7, 6, 128, 64, 3
    tw = Twin()
    xg, xc = tw.halo(B, H, W, Cin, 1, gen, fill_halo=False)
    wg, wc = tw.tensor(Cin, Cout, k, k, gen=gen, scale=0.05)
    ctx = tw.gctx
    y = Halo(ctx, B, 2 * H, 2 * W, Cout, 0)
    plan = Plan(ctx)
    for ph in G.convT_fwd_phases(H, W, k, 1):
        spec = G.convT_fwd_pack(Cin, Cout, k, ph.taps_hw)
        wp = ctx.zeros(spec.N, spec.K)
        ctx.keep.append(wp)
        plan.add("nirgan_pack_rows", wg.data_ptr(), wg.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K)
        emit_conv(plan, ctx, xg, G.Taps(ph.dh, ph.dw, Cin), wp, None, y, N=Cout, OH=ph.n_h, OW=ph.n_w,
                  in_oh=ph.in_oh, in_ow=ph.in_ow, out_stride=2, out_oh=ph.out_oh, out_ow=ph.out_ow)
    plan.run()
    ref = torch.nn.functional.conv_transpose2d(xc.interior().permute(0, 3, 1, 2), wc, None, stride=2, padding=1, output_padding=1)
    close(y.t.permute(0, 3, 1, 2), ref, 1e-5, "convT fwd")


@pytest.mark.parametrize("shape", [(16, 36, 36), (16, 32, 32), (7, 50, 38), (1, 133, 133), (2, 71, 101)])
def test_sub_pixel_phases_on_the_split_tile_spread_walk(shape):
    """ConvTranspose2d(256, 128, 3, s2, p1, op1) as ONE persistent launch of the split tile over its four phases (1 / 2 / 2 / 4 taps):
    the spread walk of igemm_x3.h (tile t of phase k at walk position (start[k] + t) mod 256; every workgroup gets its share of every
    phase) with tile counts that are / are not multiples of the grid (81, 64 and 52 tiles per phase), against torch in float64."""
    from nirgan_hip.engine import Weights, emit_conv_group
    B, H, W = shape
    gen = torch.Generator().manual_seed(41)
    tw = Twin()
    ctx = tw.gctx

    class Eng:
        pass
    eng = Eng()
    eng.ctx, eng.weights = ctx, Weights(ctx)
    Cin, Cout, k = 256, 128, 3
    xg, xc = tw.halo(B, H, W, Cin, 1, gen, fill_halo=False)
    wg, wc = tw.tensor(Cin, Cout, k, k, gen=gen, scale=0.05)
    bg, bc = tw.tensor(Cout, gen=gen, scale=1.0)
    y = Halo(ctx, B, 2 * H, 2 * W, Cout, 0)
    plan, pack = Plan(ctx), Plan(ctx)
    descs = []
    for ph in G.convT_fwd_phases(H, W, k, 1):
        w = eng.weights.packed(pack, wg, G.convT_fwd_pack(Cin, Cout, k, ph.taps_hw))
        descs.append(emit_conv(None, ctx, xg, G.Taps(ph.dh, ph.dw, Cin), w, bg, y, N=Cout, OH=ph.n_h, OW=ph.n_w,
                               in_oh=ph.in_oh, in_ow=ph.in_ow, out_stride=2, out_oh=ph.out_oh, out_ow=ph.out_ow))
    assert all(d.precision == 3 for d in descs) and sorted(d.ntaps for d in descs) == [1, 2, 2, 4]
    assert sum(-(-(d.B * d.OH * d.OW) // 256) for d in descs) >= 200
    emit_conv_group(plan, ctx, descs)
    pack.run()
    plan.run()
    ref = torch.nn.functional.conv_transpose2d(xc.interior().permute(0, 3, 1, 2).double(), wc.double(), bc.double(), stride=2, padding=1, output_padding=1)
    close(y.t.permute(0, 3, 1, 2), ref.float(), 2e-6, "four phases, spread walk")
    first = y.t.clone()
    for _ in range(20):                       # the walk is a fixed assignment: bitwise repeatable
        plan.run()
        assert torch.equal(y.t, first)


@pytest.mark.parametrize("shape", [(2, 7, 6), (3, 32, 32), (1, 40, 24), (16, 48, 48)])
def test_paired_phases_match_torch_float64(shape):
    """nirgan_conv_desc.out_span = 2: ConvTranspose2d(128, 64, 3, s2, p1, op1) + bias as the two paired problems of engine.emit_phase_pairs
    (128 columns = two adjacent output pixels, union of the two phases' taps, zero weight blocks) and the data gradient of
    Conv2d(64, 128, 3, s2, p1) the same way, against torch in float64 (ragged sizes: partial 256-row tiles, 64-column tile for the small
    problems)."""
    from nirgan_hip.engine import Weights, emit_phase_pairs, emit_conv_group, want_phase_pairs
    B, H, W = shape
    gen = torch.Generator().manual_seed(31)
    tw = Twin()
    ctx = tw.gctx

    class Eng:
        pass
    eng = Eng()
    eng.ctx, eng.weights = ctx, Weights(ctx)
    # forward of the transposed convolution
    Cin, Cout, k = 128, 64, 3
    xg, xc = tw.halo(B, H, W, Cin, 1, gen, fill_halo=False)
    wg, wc = tw.tensor(Cin, Cout, k, k, gen=gen, scale=0.05)
    bg, bc = tw.tensor(Cout, gen=gen, scale=1.0)
    y = Halo(ctx, B, 2 * H, 2 * W, Cout, 0)
    plan, pack = Plan(ctx), Plan(ctx)
    phases = G.convT_fwd_phases(H, W, k, 1)
    assert want_phase_pairs(ctx, phases, Cin, Cout, y)
    descs = emit_phase_pairs(eng, pack, ctx, xg, phases, lambda hw: G.convT_fwd_pack(Cin, Cout, k, hw), wg, bg, y, N=Cout, in_off=0, out_off=0)
    assert descs is not None and [d.out_span for d in descs] == [2, 2] and [d.ntaps for d in descs] == [2, 4]
    emit_conv_group(plan, ctx, descs)
    pack.run()
    plan.run()
    ref = torch.nn.functional.conv_transpose2d(xc.interior().permute(0, 3, 1, 2).double(), wc.double(), bc.double(), stride=2, padding=1, output_padding=1)
    close(y.t.permute(0, 3, 1, 2), ref.float(), 2e-6, "paired convT forward")
    # data gradient of the stride-2 convolution (even sizes: the four phases exist and pair up)
    if H % 2 == 0 and W % 2 == 0:
        Ci, Co = 64, 128
        dyg, dyc = tw.halo(B, H // 2, W // 2, Co, 1, gen, fill_halo=False)
        w2g, w2c = tw.tensor(Co, Ci, k, k, gen=gen, scale=0.05)
        dx = Halo(ctx, B, H, W, Ci, 0)
        plan2, pack2 = Plan(ctx), Plan(ctx)
        ph2 = G.conv_dgrad_s2_phases(H, W, k, 1)
        d2 = emit_phase_pairs(eng, pack2, ctx, dyg, ph2, lambda hw: G.conv_dgrad_pack(Co, Ci, k, hw), w2g, None, dx, N=Ci, in_off=0, out_off=0)
        assert d2 is not None
        emit_conv_group(plan2, ctx, d2)
        pack2.run()
        plan2.run()
        ref2 = torch.nn.grad.conv2d_input((B, Ci, H, W), w2c.double(), dyc.interior().permute(0, 3, 1, 2).double(), stride=2, padding=1)
        close(dx.t.permute(0, 3, 1, 2), ref2.float(), 2e-6, "paired data gradient")


def test_rowpacked_first_conv():
    """7x7, 3 -> 64 over a 4-channel NHWC buffer with one tap per kernel row (run = 28)."""
    gen = torch.Generator().manual_seed(5)
    B, H, W, k = 2, 20, 24, 7
    tw = Twin()
    xg, xc = tw.halo(B, H, W, 4, 3, gen)
    xg.t[..., 3] = 0
    xc.t[..., 3] = 0
    wg, wc = tw.tensor(64, 3, k, k, gen=gen, scale=0.05)
    ctx = tw.gctx
    spec = G.conv_rowpacked_pack(64, 3, k, 4)
    wp = ctx.zeros(spec.N, spec.K)
    y = Halo(ctx, B, H, W, 64, 0)
    plan = Plan(ctx)
    plan.add("nirgan_pack_rows", wg.data_ptr(), wg.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K)
    emit_conv(plan, ctx, xg, G.conv_rowpacked_taps(k, 4), wp, None, y, N=64, OH=H, OW=W)
    plan.run()
    ref = torch.nn.functional.conv2d(xc.t[..., :3].permute(0, 3, 1, 2), wc)
    close(y.t.permute(0, 3, 1, 2), ref, 1e-5, "rowpacked conv")


@pytest.mark.parametrize("shape", [(2, 17, 13, 64), (3, 31, 31, 512), (1, 64, 64, 256), (2, 6, 6, 8)])
def test_instnorm_forward_backward(shape):
    B, H, W, Cc = shape
    gen = torch.Generator().manual_seed(9)
    tw = Twin()
    res = []
    for border, act, use_res in ((L.BORDER_REFLECT, L.ACT_RELU, False), (L.BORDER_KEEP, L.ACT_LRELU, False),
                                 (L.BORDER_REFLECT, L.ACT_NONE, True)):
        yg, yc = tw.halo(B, H, W, Cc, 0, gen)
        yc.t.add_(0.7)      # a mean far from zero exercises the shifted-sum variance
        yg.t.copy_(yc.t)
        rg, rc = tw.halo(B, H, W, Cc, 1, gen)
        gg, gc = tw.halo(B, H, W, Cc, 1, gen)          # incoming gradient wrt the halo'd output (to fold)
        g2g, g2c = tw.halo(B, H, W, Cc, 0, gen)
        for ctx, y, r, g, g2 in ((tw.gctx, yg, rg, gg, g2g), (tw.cctx, yc, rc, gc, g2c)):
            be = tw.emu if ctx is tw.cctx else None
            n = int((be or L.backend()).nirgan_instnorm_ws_elems(B, H, W, Cc))
            ws = ctx.zeros(n)
            out = Halo(ctx, B, H, W, Cc, 1)
            stats = (ctx.zeros(B, Cc), ctx.zeros(B, Cc))
            plan = Plan(ctx)
            emit_in_fwd(plan, ctx, y, out, norm=True, act=act, residual=r if use_res else None, border=border, stats=stats, ws=ws)
            dy = Halo(ctx, B, H, W, Cc, 2)
            gsum = Halo(ctx, B, H, W, Cc, 0)
            dbias = ctx.zeros(Cc)
            emit_in_bwd(plan, ctx, g=g, g_fold=(border == L.BORDER_REFLECT), g2=g2, a=out, act=act, y=y, stats=stats,
                        norm=True, dy=dy, gsum=gsum, dbias=dbias, ws=ws, shape=(B, H, W, Cc))
            res.append((plan, out, stats, dy, gsum, dbias))
        (gp, go, gs, gdy, ggs, gdb), (cp, co, cs, cdy, cgs, cdb) = res[-2:]
        tw.run(gp, cp)
        close(go.t, co.t, 1e-5, "in fwd")
        close(gs[0], cs[0], 1e-5, "mean")
        close(gs[1], cs[1], 1e-5, "rstd")
        close(ggs.t, cgs.t, 1e-5, "gsum")
        # elements within 1e-5 of the activation's kink may take either branch (the two evaluations sum the statistics in different
        # orders): they are left out of the element-wise comparison, at most a handful of them
        zc = (yc.t - cs[0][:, None, None, :]) * cs[1][:, None, None, :]
        near = (zc.abs() < 1e-5) if act != L.ACT_NONE else torch.zeros_like(zc, dtype=torch.bool)
        assert int(near.sum()) <= 1e-4 * near.numel()
        keep = torch.nn.functional.pad(~near, (0, 0, 2, 2, 2, 2), value=True).to(gdy.t.device)
        close(torch.where(keep, gdy.t, torch.zeros_like(gdy.t)).cpu(), torch.where(keep.cpu(), cdy.t, torch.zeros_like(cdy.t)), 2e-5, "in bwd")
        # torch's instance_norm on the same data
        ref = torch.nn.functional.instance_norm(yc.t.permute(0, 3, 1, 2), eps=1e-5)
        if act == L.ACT_NONE and use_res:
            ref = ref + rc.interior().permute(0, 3, 1, 2)
            close(go.interior().permute(0, 3, 1, 2), ref, 1e-5, "in fwd vs torch")


def test_layout_tap_and_loss_kernels(golden_dir):
    import os
    gen = torch.Generator().manual_seed(21)
    emu = EmuBackend()
    be = L.backend()
    st = torch.cuda.current_stream().cuda_stream
    # nchw -> halo, composite reflect (data padding 10 then conv padding 3)
    src = torch.rand(2, 3, 24, 20, generator=gen)
    gsrc = src.to(DEV)
    P = 13
    gd = torch.zeros(2, 24 + 2 * P, 20 + 2 * P, 4, device=DEV)
    L.check(be.nirgan_nchw_to_halo(gsrc.data_ptr(), 2, 3, 24, 20, gd.data_ptr(), 4, 0, 10, 3, L.BORDER_REFLECT, st))
    ref = torch.nn.functional.pad(torch.nn.functional.pad(src, (10,) * 4, mode="reflect"), (3,) * 4, mode="reflect")
    close(gd[..., :3].permute(0, 3, 1, 2), ref, 0, "composite reflect")
    # lsgan + pixel losses against the golden known answers of the reference
    z = np.load(os.path.join(golden_dir, "f3_losses.npz"))
    pd_ = torch.from_numpy(z["pred_d"]).to(DEV)
    for real, tag in ((True, "real"), (False, "fake")):
        loss = torch.zeros(1, device=DEV)
        grad = torch.empty_like(pd_)
        L.check(be.nirgan_lsgan(pd_.data_ptr(), pd_.numel(), 1.0 if real else 0.0, 1.0, loss.data_ptr(), grad.data_ptr(), st))
        close(loss[0], torch.from_numpy(z["lsgan_" + tag]), 1e-5, "lsgan")
        close(grad, torch.from_numpy(z["lsgan_grad_" + tag]), 1e-5, "lsgan grad")
    from utils.remote_sensing_indices import RemoteSensingIndices
    rgb, nir, pred = (torch.from_numpy(z[k]).to(DEV) for k in ("rgb", "nir", "pred"))
    w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333}
    for c in ("l1", "l2"):
        p = pred.clone().requires_grad_(True)
        l = RemoteSensingIndices("loss", c).get_and_weight_losses(rgb, nir, p, loss_config=w)
        l.backward()
        close(l, torch.from_numpy(z["rs_" + c]), 1e-5, "rs " + c)
        close(p.grad, torch.from_numpy(z[f"rs_{c}_grad"]), 1e-4, "rs grad " + c)
        for k, v in RemoteSensingIndices("loss", c).get_and_weight_losses(rgb, nir, pred, mode="logging_dict").items():
            close(v, torch.from_numpy(z[f"rslog_{c}/{k}"]), 2e-5, k)
    from model.pix2pix import HipL1Loss
    p = pred.clone().requires_grad_(True)
    l = HipL1Loss()(p, nir)
    l.backward()
    close(l, torch.from_numpy(z["l1"]), 1e-5, "l1")
    close(p.grad, torch.from_numpy(z["l1_grad"]), 1e-6, "l1 grad")


@pytest.mark.parametrize("case", [(2, 7, 70, 70, 3, 52), (1, 7, 262, 262, 0, 52), (3, 4, 67, 80, 0, 16), (2, 7, 40, 40, 0, 52), (2, 4, 31, 31, 0, 16)])
def test_tap_gather_kernels(case):
    """nirgan_tap_gather: out[y][x] = act(bias + sum_t Q[y+crop+dh_t][x+crop+dw_t][t]) over the k*k tap planes of a single-output-channel
    convolution (7x7 64->1 + tanh with the data-padding crop, model/networks.py:367-368; 4x4 512->1, :579)."""
    import ctypes as C
    B, k, OH, OW, crop, qcs = case
    g = torch.Generator().manual_seed(19)
    qh, qw = OH + k - 1, OW + k - 1
    q = torch.randn(B, qh, qw, qcs, generator=g)
    bias = torch.randn(1, generator=g)
    H2, W2 = OH - 2 * crop, OW - 2 * crop
    ref = torch.zeros(B, H2, W2, dtype=torch.float64) + bias.double()
    for t in range(k * k):
        dh, dw = t // k, t % k
        ref += q[:, crop + dh:crop + dh + H2, crop + dw:crop + dw + W2, t].double()
    ref = torch.tanh(ref).float()
    qd, bd, out = q.to(DEV), bias.to(DEV), torch.zeros(B, H2, W2, device=DEV)
    d = L.TapGatherDesc()
    d.q, d.q_hp, d.q_wp, d.q_cs, d.ntaps = qd.data_ptr(), qh, qw, qcs, k * k
    for t in range(k * k):
        d.tap_dh[t], d.tap_dw[t] = t // k, t % k
    d.bias, d.act, d.B, d.OH, d.OW, d.crop, d.dst = bd.data_ptr(), L.ACT_TANH, B, OH, OW, crop, out.data_ptr()
    L.call("nirgan_tap_gather", C.byref(d), torch.cuda.current_stream().cuda_stream)
    close(out, ref, 5e-6, "tap gather")


def test_adam_kernel_matches_torch_optimizer():
    torch.manual_seed(3)
    n = 100003
    p = torch.randn(n + 1)[:n].contiguous()
    q = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([q], lr=2e-4, betas=(0.5, 0.999))
    gp, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for step in range(1, 5):
        g = torch.randn(n) * (10.0 ** (step - 3))
        q.grad = g.clone()
        opt.step()
        gg = g.to(DEV)
        L.call("nirgan_adam", gp.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), n, 2e-4, 0.5, 0.999, 1e-8, step, st)
        close(gp, q.detach(), 1e-6, f"adam step {step}")


def test_bilinear_and_inject_match_torch():
    gen = torch.Generator().manual_seed(8)
    B, S, O_ = 2, 128, 69
    src = torch.randn(B, 1, S, S, generator=gen)
    st = torch.cuda.current_stream().cuda_stream
    gs, gd = src.to(DEV), torch.zeros(B, O_, O_, device=DEV)
    L.call("nirgan_bilinear_fwd", gs.data_ptr(), B, S, S, gd.data_ptr(), O_, O_, st)
    s2 = src.clone().requires_grad_(True)
    ref = torch.nn.functional.interpolate(s2, size=(O_, O_), mode="bilinear", align_corners=False)
    close(gd, ref[:, 0], 1e-6, "bilinear fwd")
    dd = torch.randn(B, O_, O_, generator=gen)
    ref.backward(dd[:, None])
    gds = torch.zeros(B, S, S, device=DEV)
    L.call("nirgan_bilinear_bwd", dd.to(DEV).data_ptr(), B, O_, O_, gds.data_ptr(), S, S, st)
    close(gds, s2.grad[:, 0], 1e-5, "bilinear bwd")


@pytest.mark.parametrize("shape", [(2, 20, 24, 64), (3, 64, 48, 32), (1, 6, 10, 8), (2, 256, 256, 64)])
def test_channel_dgrad_matches_torch(shape):
    """dD/dpred only: data gradient of Conv2d(4, 64, 4, stride 2, padding 1) wrt input channel 3 (the output-stationary k4/s2 kernel:
    partial 8x8 tiles, borders, several images)."""
    gen = torch.Generator().manual_seed(13)
    B, H, W, Cout = shape
    OH, OW = H // 2, W // 2
    w = torch.randn(Cout, 4, 4, 4, generator=gen) * 0.1
    dy = torch.randn(B, Cout, OH, OW, generator=gen)
    x = torch.zeros(B, 4, H, W, requires_grad=True)
    torch.nn.functional.conv2d(x, w, stride=2, padding=1).backward(dy)
    ctx = Ctx(DEV)
    z = Halo(ctx, B, OH, OW, Cout, 1)
    z.interior().copy_(dy.permute(0, 2, 3, 1).to(DEV))
    wg = w.to(DEV)
    out = ctx.zeros(B, H, W)
    d = L.ChanDgradDesc()
    d.dy, d.dy_hp, d.dy_wp, d.dy_pad, d.C = z.ptr, z.hp, z.wp, 1, Cout
    d.w, d.cin, d.k, d.stride, d.pad, d.channel = wg.data_ptr(), 4, 4, 2, 1, 3
    d.B, d.H, d.W, d.out = B, H, W, out.data_ptr()
    L.call("nirgan_conv_channel_dgrad", C.byref(d), torch.cuda.current_stream().cuda_stream)
    close(out, x.grad[:, 3], 1e-5, "channel dgrad")


def test_conv_split_k_matches_plain():
    """Few output tiles + long K: the split-K path (partial tiles + fixed-order reduce) equals the plain launch."""
    gen = torch.Generator().manual_seed(17)
    B, H, W, Cin, Cout, k = 2, 12, 12, 256, 256, 4
    ctx = Ctx(DEV)
    x = Halo(ctx, B, H, W, Cin, 1)
    x.t.copy_(torch.randn(x.t.shape, generator=gen))
    w = (torch.randn(Cout, Cin, k, k, generator=gen) * 0.05).to(DEV)
    bias = torch.randn(Cout, generator=gen).to(DEV)
    spec = G.conv_fwd_pack(Cout, Cin, k)
    wp = ctx.zeros(spec.N, spec.K)
    L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    OH = G.conv_out(H, k, 1, 1)
    outs = []
    for split in (False, True):
        y = Halo(ctx, B, OH, OH, Cout, 1)
        plan = Plan(ctx)
        d = emit_conv(plan, ctx, x, G.conv_fwd_taps(k, Cin), wp, bias, y, N=Cout, OH=OH, OW=OH, out_oh=1, out_ow=1, allow_split=split)
        assert (d.ksplit > 1) == split
        plan.run()
        outs.append(y.t.clone())
    close(outs[1], outs[0], 1e-5, "split-K conv")
    ref = torch.nn.functional.conv2d(x.t.permute(0, 3, 1, 2).cpu(), w.cpu(), bias.cpu())
    close(outs[1][:, 1:-1, 1:-1].permute(0, 3, 1, 2), ref, 1e-4, "split-K conv vs torch")


@pytest.mark.parametrize("shape,window", [((2, 1, 256, 256), 5), ((3, 1, 40, 50), 5), ((1, 2, 64, 33), 11), ((1, 1, 8, 8), 5),
                                          ((1, 1, 70, 6), 11)])
def test_image_metrics_kernel(shape, window):
    """nirgan_image_metrics against the oracle's restatement of utils/calculate_metrics.py (L1, L2, SSIM mean)."""
    import nirgan_oracle as O
    from utils.calculate_metrics import calculate_metrics, image_metrics_device
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(*shape, generator=g)
    target = (pred + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    got = image_metrics_device(pred.to(DEV), target.to(DEV), window_size=window).cpu()
    d = pred.double() - target.double()
    ref = torch.tensor([d.abs().mean(), (d * d).mean(), O.ssim_map(pred.double(), target.double(), window).mean()])
    close(got, ref.float(), 2e-5, "metrics")
    again = image_metrics_device(pred.to(DEV), target.to(DEV), window_size=window).cpu()
    assert torch.equal(got, again)          # fixed-order partial sums: bitwise reproducible
    if window == 5:
        m, r = calculate_metrics(pred.to(DEV), target.to(DEV), "train"), O.calculate_metrics(pred, target, "train")
        for k in r:
            close(torch.tensor(m[k]), torch.tensor(r[k]), 1e-4, k)


@pytest.mark.parametrize("shape,window", [((2, 1, 40, 50), 11), ((16, 1, 256, 256), 11), ((3, 2, 33, 31), 5), ((1, 1, 12, 9), 11)])
def test_ssim_loss_value_and_gradient(shape, window):
    """nirgan_ssim_loss (utils/losses.py:10-30 = 1 - kornia.metrics.ssim(.., 11).mean(), the lambda_ssim term of model/pix2pix.py:233-237)
    against torch autograd of the oracle's restatement in float64: value, gradient wrt the prediction (adjoint of the reflect-padded
    Gaussian filter), the += / weight contract of the C ABI, and the autograd bridge of utils.losses.ssim_loss."""
    import ctypes as C
    import nirgan_oracle as O
    from utils.losses import ssim_loss
    g = torch.Generator().manual_seed(6)
    pred = torch.rand(*shape, generator=g)
    target = (pred + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    p64 = pred.double().requires_grad_(True)
    ref_v = 1.0 - O.ssim_map(p64, target.double(), window).mean()
    ref_g, = torch.autograd.grad(ref_v, p64)
    B, Cc, H, W = shape
    pd, td = pred.to(DEV).contiguous(), target.to(DEV).contiguous()
    ws = torch.zeros(int(L.backend().nirgan_ssim_loss_ws_elems(B * Cc, H, W, window)), device=DEV)
    loss, value = torch.full((1,), 2.0, device=DEV), torch.zeros(1, device=DEV)
    base = (1e-4 * torch.randn(*shape, generator=g)).to(DEV)        # (small: grad - base below must not drown in fp32 cancellation)
    grad = base.clone()
    d = L.SsimLossDesc()
    d.pred, d.target, d.planes, d.H, d.W = pd.data_ptr(), td.data_ptr(), B * Cc, H, W
    d.window, d.sigma, d.max_val, d.eps, d.weight = window, 1.5, 1.0, 1e-12, 0.75
    d.ws, d.ws_elems, d.loss, d.value, d.grad_pred = ws.data_ptr(), ws.numel(), loss.data_ptr(), value.data_ptr(), grad.data_ptr()
    L.call("nirgan_ssim_loss", C.byref(d), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    close(value.cpu(), ref_v.detach().float().reshape(1), 2e-5, "1 - mean SSIM")
    close(loss.cpu() - 2.0, 0.75 * ref_v.detach().float().reshape(1), 2e-5, "weighted loss added to the accumulator")
    close((grad - base).cpu(), 0.75 * ref_g.float(), 1e-4, "gradient added to grad_pred")
    pa = pd.clone().requires_grad_(True)
    (2.0 * ssim_loss(pa, td, window)).backward()
    close(pa.grad.cpu(), 2.0 * ref_g.float(), 1e-4, "autograd bridge")
    close(ssim_loss(pd, td, window).cpu(), ref_v.detach().float(), 2e-5, "value-only call")


@pytest.mark.parametrize("shape", [(2, 1, 40, 50), (16, 1, 256, 256), (3, 2, 33, 31), (1, 1, 3, 5)])
def test_emd_loss_value_and_gradient(shape):
    """nirgan_emd_loss (utils/losses.py:64-78) against the oracle's restatement: value in float32 as the reference evaluates it, gradient
    wrt the prediction against autograd in float64 (sign of the CDF difference -> suffix sums -> softmax backward)."""
    import nirgan_oracle as O
    from utils.losses import emd_loss
    g = torch.Generator().manual_seed(8)
    pred = torch.rand(*shape, generator=g)
    target = (pred + 0.2 * torch.randn(*shape, generator=g)).clamp(0, 1)
    ref_v = O.emd_loss(pred, target)
    p64 = pred.double().requires_grad_(True)
    ref_g, = torch.autograd.grad(O.emd_loss(p64, target.double()), p64)
    pd, td = pred.to(DEV), target.to(DEV)
    close(emd_loss(pd, td).cpu(), ref_v, 1e-3, "emd value")     # float32 CDFs (ulp 6e-8 near 1) against differences of ~3e-4: the reference's own float evaluation is this noisy
    pa = pd.clone().requires_grad_(True)
    (3.0 * emd_loss(pa, td)).backward()
    # a CDF difference within float rounding of 0 flips its sign term: compare in L2 (a flip moves every earlier suffix sum by 2 of up to N)
    e = (pa.grad.cpu() - 3.0 * ref_g.float()).norm() / (3.0 * ref_g.float()).norm()
    assert e < 1e-3, f"emd gradient: rel L2 {e:.3e}"
    again = emd_loss(pd, td).cpu()
    assert torch.equal(again, emd_loss(pd, td).cpu())


def test_emd_loss_against_the_references_vectors(golden_dir):
    """nirgan_emd_loss against utils/losses.py::emd_loss of the REFERENCE itself (fixture f8: values and autograd gradients)."""
    import os
    from utils.losses import emd_loss
    z = np.load(os.path.join(golden_dir, "f8_emd.npz"))
    for i in range(3):
        pred, target = torch.from_numpy(z[f"pred_{i}"]).float().to(DEV), torch.from_numpy(z[f"target_{i}"]).float().to(DEV)
        close(emd_loss(pred, target).cpu().reshape(()), torch.from_numpy(z[f"value_f64_{i}"]).float(), 1e-3, "emd value")   # (float32 CDFs: see above)
        pa = pred.clone().requires_grad_(True)
        emd_loss(pa, target).backward()
        ref = torch.from_numpy(z[f"grad_f64_{i}"]).float()
        e = (pa.grad.cpu() - ref).norm() / ref.norm()
        assert e < 1e-3, f"emd gradient {i}: rel L2 {e:.3e}"


def test_location_encoder_kernel(golden_dir, tmp_path):
    """nirgan_location_encoder (fp64) against the reference's closed-form harmonics (fixture f6) and the oracle's
    Siren restatement; a SatCLIP-sized encoder (L = 10 -> 512 -> 512 -> 256, B = 32) against the oracle."""
    import os
    import nirgan_oracle as O
    from model.satclip.location_encoder import LocationEncoder, SphericalHarmonics, get_neural_network, get_positional_encoding
    z = np.load(os.path.join(golden_dir, "f6_locenc.npz"))
    lonlat = torch.from_numpy(z["lonlat"]).to(DEV)
    for L_ in (10, 16):
        got = SphericalHarmonics(L_, "closed-form").to(DEV)(lonlat).cpu()
        assert (got - torch.from_numpy(z[f"Y{L_}"])).abs().max().item() < 1e-12
    # the generator script as its text reads, evaluated with sympy (fixture f7); the default 'analytic' = its signs with the
    # orthonormal zonal constant (f7 on m != 0, f6 on m == 0)
    z7 = np.load(os.path.join(golden_dir, "f7_sh_analytic.npz"))
    got = SphericalHarmonics(10, "analytic-generator-text").to(DEV)(lonlat).cpu()
    assert (got - torch.from_numpy(z7["Y10"])).abs().max().item() < 1e-12
    got = SphericalHarmonics(10).to(DEV)(lonlat).cpu()
    zonal = [l * l + l for l in range(10)]
    want = torch.from_numpy(z7["Y10"]).clone()
    want[:, zonal] = torch.from_numpy(z["Y10"])[:, zonal]
    assert (got - want).abs().max().item() < 1e-12
    torch.manual_seed(4)
    enc = LocationEncoder(get_positional_encoding("sphericalharmonics", 10, "analytic"), get_neural_network("siren", 100, 256, 512, 2)).double().eval().to(DEV)
    g = torch.Generator().manual_seed(9)
    ll = torch.stack((torch.rand(32, generator=g, dtype=torch.float64) * 360 - 180, torch.rand(32, generator=g, dtype=torch.float64) * 180 - 90), -1)
    got = enc(ll.to(DEV)).cpu()
    p = {"nnet." + k: v.detach().cpu() for k, v in enc.nnet.state_dict().items()}
    ref = O.location_encoder_forward(p, ll, 10, 2, "analytic")
    assert got.dtype == torch.float64 and (got - ref).abs().max().item() < 1e-11 * max(ref.abs().max().item(), 1.0)
    # the reference's own SirenNet (fixture f9: its seeded state_dict) behind the reference's closed-form harmonics: the whole
    # LocationEncoder.forward from reference parts
    z9 = np.load(os.path.join(golden_dir, "f9_siren.npz"))
    enc9 = LocationEncoder(get_positional_encoding("sphericalharmonics", 10, "closed-form"), get_neural_network("siren", 100, 32, 64, 2)).double().eval()
    enc9.nnet.load_state_dict({k[len("siren/nnet."):]: torch.from_numpy(z9[k]) for k in z9.files if k.startswith("siren/nnet.")})
    got = enc9.to(DEV)(torch.from_numpy(z9["lonlat"]).to(DEV)).cpu()
    want = torch.from_numpy(z9["y_lonlat"])
    assert (got - want).abs().max().item() < 1e-11 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("shape,ties", [((2, 1, 64, 64), False), ((1, 1, 256, 256), False), ((1, 1, 50, 70), False),
                                        ((2, 1, 64, 64), True), ((1, 1, 4, 4), True), ((1, 1, 512, 512), True)])
def test_histogram_match_kernel(shape, ties):
    """nirgan_hist_match (bitonic sort + quantile interpolation on the device) against the numpy restatement of
    skimage.exposure.match_histograms; sizes below / at / above one LDS chunk, non-powers of two, heavy ties."""
    import nirgan_oracle as O
    from nirgan_hip.inference import histogram_match
    g = torch.Generator().manual_seed(13)
    img = torch.randn(*shape, generator=g)
    ref = torch.rand(*shape, generator=g) * 2 - 0.5
    if ties:
        img, ref = (img * 4).round() / 4, (ref * 16).round() / 16
        img[0, 0, 0, 0], img[0, 0, 0, 1] = 0.0, -0.0
    got = histogram_match(img.to(DEV), ref.to(DEV)).cpu()
    want = O.histogram_match(img, ref)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1.0)
    if not ties and all(t[b].unique().numel() == t[b].numel() for t in (img, ref) for b in range(shape[0])):
        assert torch.equal(got.flatten().sort().values, ref.flatten().sort().values)      # no ties at all: a permutation of the template
    # the script's case: low-resolution reference resized on the device first.  Matching is discontinuous in the
    # template's ties (a run of equal values interpolates, values one ulp apart do not), and an upsampled image is full
    # of ties, so the comparison is teacher-forced on the device's own resize; the resize itself is within 1e-6 of torch's
    if shape[-1] >= 16:
        from nirgan_hip.inference import resize_bilinear
        small = ref[..., ::4, ::4].contiguous()
        up = resize_bilinear(small.to(DEV), shape[-2], shape[-1]).cpu()
        want_up = torch.nn.functional.interpolate(small, size=shape[-2:], mode="bilinear", align_corners=False)
        assert (up - want_up).abs().max().item() <= 1e-6 * max(want_up.abs().max().item(), 1.0)
        got = histogram_match(img.to(DEV), small.to(DEV)).cpu()
        want = O.histogram_match(img, up)
        assert (got - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_fullsize_adjoint_and_linearity_identities(precision):
    """Size-independent properties at the benchmark size (bs 16, 64x64x256 residual-block layer; no oracle needed):
    <conv(x; W), dY> = <x, dgrad(dY; W)> = <W, wgrad(x, dY)> (the three kernels are one bilinear form), and
    conv(a x1 + b x2) = a conv(x1) + b conv(x2).  Dot products are taken in float64 on the device."""
    case = ("res_full", 16, 64, 64, 256, 256, 3, 1, 1)
    _, B, H, W, Cin, Cout, k, s, p = case
    ctx = Ctx(DEV, precision)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = Halo(ctx, B, H, W, Cin, p)
    x.t.copy_(torch.randn(x.t.shape, generator=g, device=DEV))            # halo included: the ring of the data gradient counts
    w = torch.randn(Cout, Cin, k, k, generator=g, device=DEV) * 0.05
    b = torch.zeros(Cout, device=DEV)
    plan, plan2, y, dy, gw, gx = build_conv_case(ctx, x, w, b, case)
    plan.run()
    y1 = y.t.clone()
    dyv = torch.randn(y.t.shape, generator=g, device=DEV)
    dy.interior().copy_(dyv)
    plan2.run()
    torch.cuda.synchronize()
    dot = lambda a, c: (a.double() * c.double()).sum().item()              # noqa: E731
    lhs = dot(y1, dyv)
    tol = 1e-5 if precision == "fp32" else 2e-4
    assert abs(dot(x.t, gx.t) - lhs) <= tol * abs(lhs), ("data gradient is not the adjoint", dot(x.t, gx.t), lhs)
    assert abs(dot(w, gw) - lhs) <= tol * abs(lhs), ("weight gradient is not the adjoint", dot(w, gw), lhs)
    # linearity in x
    x2 = torch.randn(x.t.shape, generator=g, device=DEV)
    x1 = x.t.clone()
    x.t.copy_(x2)
    plan.run()
    y2 = y.t.clone()
    x.t.copy_(0.75 * x1 - 1.5 * x2)
    plan.run()
    err = (y.t - (0.75 * y1 - 1.5 * y2)).abs().max().item()
    assert err <= (1e-5 if precision == "fp32" else 2e-4) * y1.abs().max().item(), err


def test_fullsize_postprocess_properties():
    """512x512 planes: histogram matching is idempotent and maps onto the template's values; SSIM(x, x) = 1, symmetric."""
    from nirgan_hip.inference import histogram_match
    from utils.calculate_metrics import image_metrics_device
    g = torch.Generator(device=DEV).manual_seed(9)
    a = torch.randn(2, 1, 512, 512, generator=g, device=DEV)
    b = torch.rand(2, 1, 512, 512, generator=g, device=DEV)
    m = histogram_match(a, b)
    assert torch.equal(histogram_match(m, b), m)
    assert torch.equal(histogram_match(b, b), b)
    for i in range(2):
        if b[i].unique().numel() == b[i].numel() and a[i].unique().numel() == a[i].numel():
            assert torch.equal(m[i].flatten().sort().values, b[i].flatten().sort().values)
    order_a = a.flatten(1).argsort(dim=1, stable=True)
    assert bool(m.flatten(1).gather(1, order_a).diff(dim=1).ge(0).all()), "matching must be monotone in the source value"
    s_xx = image_metrics_device(b, b)
    assert s_xx[0].item() == 0.0 and s_xx[1].item() == 0.0 and abs(s_xx[2].item() - 1.0) < 1e-6
    c = (b + 0.05 * torch.randn(b.shape, generator=g, device=DEV)).clamp(0, 1)
    assert abs(image_metrics_device(b, c)[2].item() - image_metrics_device(c, b)[2].item()) < 1e-6


def test_conv_random_shapes_fuzz():
    """40 random layer shapes (channels 4..160 in multiples of 4, odd sizes, stride 1/2, every operand
    precision; k in {1,3,4}) through forward, weight gradient and data gradient: device vs the numpy restatement of the descriptors."""
    import random
    rnd = random.Random(1234)
    for it in range(40):
        k = rnd.choice([1, 3, 3, 4])          # 7x7 layers run row-packed / as tap planes (their own tests): <= 16 taps here
        s = rnd.choice([1, 1, 2]) if k in (3, 4) else 1
        p = rnd.choice([0, 1]) if k == 1 else (k - 1) // 2 if k != 4 else 1
        p = 0 if k == 1 else p
        Cin, Cout = 4 * rnd.randint(1, 40), 4 * rnd.randint(1, 40)
        B = rnd.randint(1, 3)
        H, W = rnd.randint(max(k, 5), 23), rnd.randint(max(k, 5), 23)
        if s == 2 and k == 3:
            H, W = H & ~1, W & ~1          # the engines use 3x3 stride 2 on even sizes only (down-sampling of 4k tiles)
        if s == 2 and k == 4:
            H, W = H & ~1, W & ~1
        H, W = max(H, 6), max(W, 6)
        precision = rnd.choice(["fp32", "bf16", "bf16x3"])
        case = (f"fuzz{it}", B, H, W, Cin, Cout, k, s, p)
        gen = torch.Generator().manual_seed(100 + it)
        tw = Twin(precision)
        xg, xc = tw.halo(B, H, W, Cin, p, gen)
        wg, wc = tw.tensor(Cout, Cin, k, k, gen=gen, scale=0.1)
        bg, bc = tw.tensor(Cout, gen=gen)
        gp, gp2, gy, gdy, ggw, ggx = build_conv_case(tw.gctx, xg, wg, bg, case)
        cp, cp2, cy, cdy, cgw, cgx = build_conv_case(tw.cctx, xc, wc, bc, case)
        tw.run(gp, cp)
        close(gy.t, cy.t, 2e-5, f"{case} {precision} fwd")
        cdy.interior().copy_(cy.t)
        gdy.interior().copy_(cy.t.to(DEV))
        tw.run(gp2, cp2)
        close(ggw, cgw, 1e-4, f"{case} {precision} wgrad")
        close(ggx.t, cgx.t, 2e-5, f"{case} {precision} dgrad")


def test_bf16_twins_and_twin_fed_convolution():
    """bf16 mode: the instance-norm kernels mirror every store into a bf16 twin (forward `out`, backward `dy`), bit for bit
    the round-to-nearest-even of the fp32 buffer, halo included; a convolution fed from the twin (in_bf16) with bf16-stored
    weights equals the numpy restatement and equals the same convolution fed from the fp32 buffer (same operand values)."""
    from conv_cases import _packed
    B, H, W, Cc, Cout = 2, 12, 10, 64, 128
    gen = torch.Generator().manual_seed(5)
    tw = Twin("bf16")
    res = []
    for ctx in (tw.gctx, tw.cctx):
        dev = ctx.device
        y = Halo(ctx, B, H, W, Cc, 0)
        y.t.copy_(torch.randn(y.t.shape, generator=torch.Generator().manual_seed(6)).to(dev))
        out = Halo(ctx, B, H, W, Cc, 1, twin=True)
        assert out.t16 is not None
        stats = (ctx.zeros(B, Cc), ctx.zeros(B, Cc))
        ws = ctx.zeros(int(L.backend().nirgan_instnorm_ws_elems(B, H, W, Cc)) if ctx is tw.gctx else 1 << 16)
        plan = Plan(ctx)
        emit_in_fwd(plan, ctx, y, out, norm=True, act=L.ACT_RELU, border=L.BORDER_REFLECT, stats=stats, ws=ws)
        w = (torch.randn(Cout, Cc, 3, 3, generator=torch.Generator().manual_seed(7)) * 0.05).to(dev)
        spec = G.conv_fwd_pack(Cout, Cc, 3)
        wp = _packed(ctx, plan, w, spec)
        assert wp.dtype == torch.bfloat16
        o = Halo(ctx, B, H, W, Cout, 0)
        d = emit_conv(plan, ctx, out, G.conv_fwd_taps(3, Cc), wp, None, o, N=Cout, OH=H, OW=W)
        assert d.in_bf16 == 1 and d.w_bf16 == 1
        # the same convolution from the fp32 buffer (twin detached): identical operand values after rounding
        o2 = Halo(ctx, B, H, W, Cout, 0)
        t16, out.t16 = out.t16, None
        d2 = emit_conv(plan, ctx, out, G.conv_fwd_taps(3, Cc), wp, None, o2, N=Cout, OH=H, OW=W)
        out.t16 = t16
        assert d2.in_bf16 == 0
        # backward twin
        g = Halo(ctx, B, H, W, Cc, 0)
        g.t.copy_(torch.randn(g.t.shape, generator=torch.Generator().manual_seed(8)).to(dev))
        dy = Halo(ctx, B, H, W, Cc, 2, twin=True)
        emit_in_bwd(plan, ctx, g=g, act=L.ACT_RELU, y=y, stats=stats, norm=True, dy=dy, ws=ws, shape=(B, H, W, Cc))
        # weight gradient of that convolution from the twins (dY of the conv output through a norm-free backward pass so that
        # its twin is written), and the same from the fp32 buffers
        gc = Halo(ctx, B, H, W, Cout, 0)
        gc.t.copy_(torch.randn(gc.t.shape, generator=torch.Generator().manual_seed(9)).to(dev))
        dyc = Halo(ctx, B, H, W, Cout, 2, twin=True)
        emit_in_bwd(plan, ctx, g=gc, act=L.ACT_NONE, norm=False, dy=dyc, shape=(B, H, W, Cout))
        gw16, gw32 = ctx.zeros(Cout, Cc, 3, 3), ctx.zeros(Cout, Cc, 3, 3)
        dw = emit_wgrad(plan, ctx, dyc, out, G.conv_fwd_taps(3, Cc), spec, gw16, N=Cout, OH=H, OW=W, p_oh=2, p_ow=2)
        assert dw.pq_bf16 == 1
        ta, tb = dyc.t16, out.t16
        dyc.t16 = out.t16 = None
        dw2 = emit_wgrad(plan, ctx, dyc, out, G.conv_fwd_taps(3, Cc), spec, gw32, N=Cout, OH=H, OW=W, p_oh=2, p_ow=2)
        dyc.t16, out.t16 = ta, tb
        assert dw2.pq_bf16 == 0
        res.append((plan, out, o, o2, dy, gw16, gw32))
    (gp, gout, go, go2, gdy, ggw16, ggw32), (cp, cout_, co, co2, cdy, cgw16, cgw32) = res
    tw.run(gp, cp)
    close(ggw16, cgw16, 1e-4, "twin-fed weight gradient vs restatement")
    close(ggw16, ggw32, 1e-5, "twin-fed weight gradient vs fp32-fed")
    assert torch.equal(gout.t16.cpu(), gout.t.cpu().to(torch.bfloat16)), "forward twin is not RNE(out)"
    assert torch.equal(gdy.t16.cpu(), gdy.t.cpu().to(torch.bfloat16)), "backward twin is not RNE(dy)"
    assert float(gout.t[:, 0].abs().max()) > 0 and float(gdy.t[:, :2].abs().max()) == 0     # reflect halo filled, zero halo kept
    close(go.t, co.t, 1e-5, "twin-fed conv vs restatement")
    close(go.t, go2.t, 1e-6, "twin-fed conv vs fp32-fed conv")


@pytest.mark.parametrize("shape", [(2, 8, 12, 64, 128, 3), (1, 64, 64, 256, 256, 3), (3, 6, 5, 32, 128, 3), (2, 9, 7, 64, 192, 3), (1, 69, 69, 256, 256, 3),
                                   (16, 64, 64, 256, 256, 3), (2, 8, 12, 64, 128, 4), (2, 31, 31, 256, 512, 4), (3, 7, 5, 32, 128, 4), (32, 31, 31, 256, 512, 4),
                                   (2, 8, 12, 64, 128, 6), (1, 64, 64, 256, 256, 6), (3, 6, 5, 32, 128, 6), (2, 9, 7, 64, 192, 6), (1, 69, 69, 256, 256, 6),
                                   (16, 64, 64, 256, 256, 6), (16, 66, 66, 256, 256, 6),
                                   # the persistent plane GEMM (C = 256) with partly filled N tiles, few tiles per workgroup, ragged M tiles
                                   (2, 20, 14, 256, 192, 6), (2, 20, 14, 256, 320, 3), (1, 7, 9, 256, 128, 6), (5, 33, 31, 256, 256, 6),
                                   # C = 512: the plain plane GEMM runs as persistent workgroups too (16 K-steps of 32 per tile)
                                   (2, 20, 14, 512, 192, 6), (16, 34, 34, 512, 256, 4), (3, 9, 13, 512, 128, 3)])
def test_wino6_conv3x3_matches_direct(shape):
    """nirgan_wino6_weights_r + input / (r+3)^2 plane GEMMs / output transform against torch's conv2d in float64 (the reference's
    nn.Conv2d arithmetic) and the numpy restatement: fp32 rounding only.  F(4x4,3x3) and F(4x4,4x4); extents that are multiples of 4,
    odd, smaller than a tile; K = 192 leaves a half-used N tile; the benchmark's residual-block layer and its PatchGAN 4x4 layer."""
    import ctypes as C
    B, H, W, Cc, K, v = shape                                            # v: the descriptors' variant code, 6 = F(6x6,3x3)
    r, mo = (3, 6) if v == 6 else (v, 4)
    NP = (mo + r - 1) ** 2
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, H + r - 1, W + r - 1, Cc, generator=g)            # halo included: any values (reflect or zero in the nets)
    w = torch.randn(K, Cc, r, r, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double()).permute(0, 2, 3, 1).float()
    T = B * (-(-H // mo)) * (-(-W // mo))
    assert int(L.backend().nirgan_wino6_tiles_r(B, H, W, v)) == T
    outs = []
    emu = EmuBackend()
    for dev, be in ((DEV, None), ("cpu", emu)):
        if be is not None and B * H * W > 20000:
            continue
        xt, wt, bt = x.to(dev).contiguous(), w.to(dev).contiguous(), b.to(dev)
        U = torch.zeros(NP * K * Cc, device=dev)
        V = torch.zeros(NP * T * Cc, device=dev)
        M = torch.full((NP * T * K,), float("nan"), device=dev)
        y = torch.full((B, H, W, K), float("nan"), device=dev)
        zero = torch.zeros(64, device=dev)
        d = L.Wino6Desc()
        d.r = v
        d.x, d.x_hp, d.x_wp, d.B, d.H, d.W, d.C, d.K = xt.data_ptr(), H + r - 1, W + r - 1, B, H, W, Cc, K
        d.U, d.bias, d.V, d.V_elems, d.M, d.M_elems = U.data_ptr(), bt.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel()
        d.y, d.zero_page = y.data_ptr(), zero.data_ptr()
        if be is None:
            st = torch.cuda.current_stream().cuda_stream
            L.call("nirgan_wino6_weights_r", wt.data_ptr(), K, Cc, v, 0, U.data_ptr(), st)
            L.call("nirgan_wino6_conv3x3", C.byref(d), st)
            torch.cuda.synchronize()
        else:
            assert be.nirgan_wino6_weights_r(wt.data_ptr(), K, Cc, v, 0, U.data_ptr()) == 0
            assert be.nirgan_wino6_conv3x3(d) == 0
        outs.append(y.cpu())
    tol = 3e-5 if v == 3 else 8e-5          # F(4x4,4x4) / F(6x6,3x3): seven / eight points; measured 1e-5 .. 3e-5
    close(outs[0], ref, tol, "device vs torch")
    if len(outs) > 1:
        close(outs[1], ref, 1e-5, "restatement vs torch")
        close(outs[0], outs[1], tol, "device vs restatement")


@pytest.mark.parametrize("hw", [(12, 16, 3, 3), (9, 11, 3, 3), (64, 64, 3, 3), (12, 16, 4, 4), (9, 11, 4, 4), (31, 31, 4, 4), (12, 16, 3, 6), (9, 11, 3, 6),
                                (64, 64, 3, 6), (13, 6, 3, 6), (64, 64, 3, 6, 256), (21, 17, 3, 6, 256), (22, 18, 3, 3, 256), (40, 40, 3, 6, 256, 384),
                                (64, 64, 3, 6, 256, 256, "wgrad_one_unit"), (64, 64, 3, 6, 256, 256, "gemm_one_tile"),
                                (64, 64, 3, 6, 256, 256, "no_pair"), (64, 64, 3, 6, 256, 256, "no_pair_wgrad_one_unit")])
def test_wino6_backward_matches_autograd(hw, monkeypatch):
    """The exact-fp32 backward of a ResnetBlock convolution as the engines emit it with F(4x4,3x3): data gradient over the padded
    extent (dY transformed once for both uses), transform-domain weight gradient (36 planes in one weight-gradient launch, then
    G^T dU G) against torch autograd of conv2d in float64; device and numpy restatement.  Even and odd extents; V re-derived from the
    forward input here (the nets keep the forward's V: covered by the net-level tests)."""
    from nirgan_hip.engine import emit_wino6, emit_wino6_backward, SlabPool, _FullExtent
    H, W, r, v = hw[:4]                               # H x W = the layer's OUTPUT extent; its input is (H + r - 3) x (W + r - 3) + halo 1
    if len(hw) > 6:                                   # the A/B fallbacks of the persistent pair launch (descriptor field `algo`, set when the plan is built)
        from nirgan_hip.options import OPT
        if "wgrad_one_unit" in hw[6]:
            monkeypatch.setattr(OPT, "wgrad_algo", L.WGRAD_ONE_UNIT)
        if hw[6] == "gemm_one_tile":
            monkeypatch.setattr(OPT, "w6_gemm_algo", L.W6_ONE_TILE)
        if "no_pair" in hw[6]:                        # two separate launches
            monkeypatch.setattr(OPT, "w6_pair", False)
    if v == 3:
        from nirgan_hip.options import OPT as _OPT
        monkeypatch.setattr(_OPT, "winograd", "f4")   # 3x3 filters: F(4x4,3x3) instead of the default F(6x6,3x3)
    from nirgan_hip.engine import wino6_variant
    assert wino6_variant(r) == v
    # 256 input channels of the data-gradient GEMM (= the layer's output channels) take the persistent pair launch: the transform-domain
    # weight gradient and the plane GEMMs walked by the same 512 workgroups (csrc/wino6.hip::wino6_pair16p_kernel)
    B, Cin, Cout = 2, (hw[5] if len(hw) > 5 else hw[4] if len(hw) > 4 else 128), (hw[4] if len(hw) > 4 else 128)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, H + r - 1, W + r - 1, Cin, generator=g)
    w = torch.randn(Cout, Cin, r, r, generator=g) * 0.05
    dyv = torch.randn(B, H, W, Cout, generator=g)
    xp = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xp, wt).backward(dyv.permute(0, 3, 1, 2).double())
    ref_gx, ref_gw = xp.grad.permute(0, 2, 3, 1).float(), wt.grad.float()
    tw = Twin("fp32")
    res = []
    for ctx in (tw.gctx, tw.cctx):
        dev = ctx.device
        inp = Halo(ctx, B, H + r - 3, W + r - 3, Cin, 1)
        inp.t.copy_(x.to(dev))
        dy = Halo(ctx, B, H, W, Cout, r - 1)
        dy.interior().copy_(dyv.to(dev))
        gx = Halo(ctx, B, H + r - 3, W + r - 3, Cin, 1)
        gx.t.fill_(float("nan"))
        gw = ctx.zeros(Cout, Cin, r, r)
        wd = w.to(dev).contiguous()
        ctx.keep.append(wd)
        plan, pack = Plan(ctx), Plan(ctx)
        wdesc = emit_wino6(None, pack, ctx, dy, wd, None, _FullExtent(gx), H=gx.hp, W=gx.wp, cin=Cout, cout=Cin, flip=True, r=r)
        emit_wino6_backward(plan, ctx, dy, inp, gw, OH=H, OW=W, cin=Cin, cout=Cout, slabs_pool=SlabPool(ctx), dgrad=wdesc, r=r)
        res.append((pack, plan, gx, gw))
    (gpk, gpl, ggx, ggw), (cpk, cpl, cgx, cgw) = res
    tw.run(gpk, cpk)
    tw.run(gpl, cpl)
    close(cgx.t, ref_gx, 1e-5, "restatement: data gradient")
    close(cgw, ref_gw, 1e-5, "restatement: weight gradient")
    close(ggx.t, ref_gx, 3e-5 if v == 3 else 8e-5, "device: data gradient")
    close(ggw, ref_gw, 1e-4, "device: weight gradient")


@pytest.mark.parametrize("hw", [(12, 16, 3), (9, 11, 3), (64, 64, 3), (12, 16, 6), (9, 11, 6), (64, 64, 6)])
def test_wino6_input_transform_with_the_instance_norm_folded_in(hw):
    """nirgan_instnorm_fwd(out = NULL: statistics only) + nirgan_wino6_input_norm(y, mean, rstd, ReLU) = the V that the full
    instance-norm pass (ReLU, reflect halo 1) followed by nirgan_wino6_input produces -- bitwise, same fp32 arithmetic."""
    import ctypes as C
    from nirgan_hip.engine import emit_in_fwd
    H, W, v = hw
    B, Cc = 2, 128
    g = torch.Generator().manual_seed(31)
    y = (torch.randn(B, H, W, Cc, generator=g) * 1.7 + 0.3).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    ctx = Ctx(DEV, "fp32")
    yh = Halo(ctx, B, H, W, Cc, 0)
    yh.t.copy_(y)
    out = Halo(ctx, B, H, W, Cc, 1)
    stats = (ctx.zeros(B, Cc), ctx.zeros(B, Cc))
    ws = ctx.zeros(int(L.backend().nirgan_instnorm_ws_elems(B, H, W, Cc)))
    full = Plan(ctx)
    emit_in_fwd(full, ctx, yh, out, norm=True, act=L.ACT_RELU, border=L.BORDER_REFLECT, stats=stats, ws=ws)
    full.run()
    T = int(L.backend().nirgan_wino6_tiles_r(B, H, W, v))
    NP = 64 if v == 6 else 36
    V1, V2 = torch.zeros(NP * T * Cc, device=DEV), torch.full((NP * T * Cc,), float("nan"), device=DEV)
    d = L.Wino6Desc()
    d.r = v
    d.x, d.x_hp, d.x_wp, d.B, d.H, d.W, d.C, d.K = out.ptr, H + 2, W + 2, B, H, W, Cc, 128
    d.V, d.V_elems = V1.data_ptr(), V1.numel()
    L.call("nirgan_wino6_input", C.byref(d), st)
    d.V = V2.data_ptr()
    L.call("nirgan_wino6_input_norm", C.byref(d), yh.ptr, stats[0].data_ptr(), stats[1].data_ptr(), L.ACT_RELU, 0.2, st)
    torch.cuda.synchronize()
    assert torch.equal(V1, V2), f"max diff {(V1 - V2).abs().max().item():.3e}"


@pytest.mark.parametrize("shape", [(2, 9, 10, 1), (1, 70, 67, 0), (16, 256, 256, 0), (2, 276, 276, 10), (3, 5, 130, 2)])
def test_direct_last_layer_kernels(shape):
    """Conv2d(64, 1, 7) + bias + tanh + crop and its three gradients as direct kernels (csrc/endconv.hip) against torch's conv2d +
    autograd in fp64: full configs[1] size, the padding-10 geometry, ragged row / column groups."""
    from endconv_case import run_endconv
    B, OH, OW, crop = shape
    run_endconv(DEV, B, OH, OW, crop)
    run_endconv(DEV, 1, 8, 8, 0, act=L.ACT_NONE) if B == 2 and OH == 9 else None


@pytest.mark.parametrize("shape", [(2, 12, 16, 64, 128, 6), (16, 64, 64, 256, 256, 6), (3, 9, 7, 32, 128, 3), (2, 31, 31, 64, 128, 4)])
def test_wino6_output_leaves_the_instance_norm_partial_sums(shape):
    """nirgan_wino6_output with stats_ws + nirgan_instnorm_fwd(stats_chunks, stats_shift = bias) gives the mean / rstd of the
    separate statistics pass over y (model/networks.py:30: InstanceNorm2d, biased variance): same sums in another order and about
    another shift (the bias instead of the first pixel), so equal to fp32 rounding; ragged tiles count only their stored outputs."""
    import ctypes as C
    B, H, W, Cc, K, v = shape
    r, mo = (3, 6) if v == 6 else (v, 4)
    NP = (mo + r - 1) ** 2
    g = torch.Generator().manual_seed(41)
    be = L.backend()
    T = int(be.nirgan_wino6_tiles_r(B, H, W, v))
    x = torch.randn(B, H + r - 1, W + r - 1, Cc, generator=g).to(DEV)
    w = (torch.randn(K, Cc, r, r, generator=g) * 0.05).to(DEV)
    bias = (torch.randn(K, generator=g) * 2.0).to(DEV)                   # a large bias: the shift matters
    U, V, M = torch.zeros(NP * K * Cc, device=DEV), torch.zeros(NP * T * Cc, device=DEV), torch.zeros(NP * T * K, device=DEV)
    y = torch.zeros(B, H, W, K, device=DEV)
    zero = torch.zeros(64, device=DEV)
    sws = torch.full((T * 4 * K,), float("nan"), device=DEV)
    d = L.Wino6Desc()
    d.r = v
    d.x, d.x_hp, d.x_wp, d.B, d.H, d.W, d.C, d.K = x.data_ptr(), H + r - 1, W + r - 1, B, H, W, Cc, K
    d.U, d.bias, d.V, d.V_elems, d.M, d.M_elems = U.data_ptr(), bias.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel()
    d.y, d.zero_page, d.stats_ws, d.stats_ws_elems = y.data_ptr(), zero.data_ptr(), sws.data_ptr(), sws.numel()
    st = torch.cuda.current_stream().cuda_stream
    L.call("nirgan_wino6_weights_r", w.data_ptr(), K, Cc, v, 0, U.data_ptr(), st)
    L.call("nirgan_wino6_conv3x3", C.byref(d), st)
    res = []
    for pre in (True, False):
        mean, rstd = torch.zeros(B, K, device=DEV), torch.zeros(B, K, device=DEV)
        ws = sws if pre else torch.zeros(int(be.nirgan_instnorm_ws_elems(B, H, W, K)), device=DEV)
        n = L.InFwdDesc()
        n.y, n.B, n.H, n.W, n.C, n.norm, n.eps = y.data_ptr(), B, H, W, K, 1, 1e-5
        n.mean, n.rstd, n.ws, n.ws_elems = mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), ws.numel()
        if pre:
            n.stats_chunks, n.stats_shift = T // B, bias.data_ptr()
        L.call("nirgan_instnorm_fwd", C.byref(n), st)
        torch.cuda.synchronize()
        res.append((mean.cpu(), rstd.cpu()))
    y64 = y.double().cpu().reshape(B, H * W, K)
    close(res[0][0], y64.mean(1), 1e-6, "mean from the output transform's partial sums")
    close(res[1][0], y64.mean(1), 1e-6, "mean from the statistics pass")
    ref_rstd = 1.0 / torch.sqrt(y64.var(1, unbiased=False) + 1e-5)
    # (3e-6: fp32 summation noise of 4 096 squares against float64 of the same y -- round 5 measured 2.02e-6 on the 16 x 64 x 64 x 256 shape
    # after an unrelated recompile of the weight transform moved U by an ulp; this launch sets no U3, its GEMMs are exact fp32)
    close(res[0][1], ref_rstd, 3e-6, "rstd from the output transform's partial sums")
    close(res[1][1], ref_rstd, 3e-6, "rstd from the statistics pass")


@pytest.mark.parametrize("case", [("conv", 2, 32, 32, 32, 64, 3, 1), ("conv", 3, 32, 32, 64, 128, 3, 2), ("convT", 2, 16, 16, 128, 64, 3, 2), ("conv", 16, 128, 128, 128, 256, 3, 2),
                                  ("conv", 2, 32, 32, 16, 8, 3, 1), ("conv", 2, 64, 64, 8, 16, 3, 2), ("convT", 2, 16, 16, 32, 16, 3, 2), ("conv", 2, 16, 16, 32, 32, 3, 1), ("rowpacked", 2, 64, 64, 4, 8, 7, 1), ("rowpacked", 16, 256, 256, 4, 64, 7, 1),
                                  ("conv", 2, 32, 32, 4, 8, 4, 2), ("conv", 2, 16, 16, 32, 32, 3, 1, "residual")])
def test_conv_epilogue_leaves_the_instance_norm_partial_sums(case):
    """A ConvIN layer with the direct tiles: the convolution's epilogue leaves per-(64 pixels) partial sums of its output without the
    bias, nirgan_instnorm_fwd merges them (stats_chunks, stats_shift = bias) -- the output of the layer (normalised, ReLU, halo) equals
    the one produced with the separate statistics pass (OPT.epilogue_stats = False) to fp32 rounding; device against the numpy restatement
    too.  Stride-1, stride-2, and the four sub-pixel phases of a transposed convolution numbering their chunks into one workspace."""
    import os
    from nirgan_hip.engine import ConvIN
    kind, B, H, W, Cin, Cout, k, s_ = case[:8]

    class Eng:
        pass

    def build(ctx, stats):
        from nirgan_hip.engine import Weights, _Scratch, SlabPool
        from nirgan_hip.options import OPT
        OPT.epilogue_min_pixels = 0          # small layers too (the engines keep the separate pass below 16 K pixels)
        OPT.epilogue_stats = bool(stats)
        try:
            g = torch.Generator().manual_seed(5)
            eng = Eng()
            eng.ctx, eng.weights, eng.scratch, eng.slabs, eng.need_backward = ctx, Weights(ctx), _Scratch(ctx), SlabPool(ctx), False
            pad = k // 2 if kind == "rowpacked" else 1
            inp = Halo(ctx, B, H, W, Cin, pad)
            inp.t.copy_(torch.randn(inp.t.shape, generator=g).to(ctx.device))
            cin_w = 3 if kind == "rowpacked" else Cin
            wshape = (Cin, Cout, k, k) if kind == "convT" else (Cout, cin_w, k, k)
            w = (torch.randn(wshape, generator=g) * 0.05).to(ctx.device)
            bias = (torch.randn(Cout, generator=g) * 3.0).to(ctx.device)
            res = None
            if len(case) > 8:
                res = Halo(ctx, B, H, W, Cout, 1)
                res.t.copy_(torch.randn(res.t.shape, generator=g).to(ctx.device))
            layer = ConvIN(eng, "t", kind, inp, w, bias, k=k, s=s_, p=pad, cout=Cout, norm=True, act=(L.ACT_NONE if res is not None else L.ACT_RELU), out_pad=1,
                           out_border=L.BORDER_REFLECT, residual=res, cin_real=(3 if kind == "rowpacked" else None))
            plan, pack = Plan(ctx), Plan(ctx)
            layer.emit_fwd(plan, pack)
            pack.run()
            plan.run()
            return layer.out.t.detach().cpu().clone(), layer.stats[0].cpu().clone(), layer.stats[1].cpu().clone(), plan
        finally:
            OPT.reset()

    gctx = Ctx(DEV, "fp32")
    o1, m1, r1, plan1 = build(gctx, True)
    o0, m0, r0, plan0 = build(Ctx(DEV, "fp32"), False)
    torch.cuda.synchronize()
    d1 = [a[0]._obj for n, a in plan1.ops if n == "nirgan_instnorm_fwd"][0]
    d0 = [a[0]._obj for n, a in plan0.ops if n == "nirgan_instnorm_fwd"][0]
    assert d1.stats_chunks > 0 and d0.stats_chunks == 0
    close(m1, m0, 1e-6, "mean")
    close(r1, r0, 3e-6, "rstd")
    close(o1, o0, 1e-5, "layer output")


@pytest.mark.parametrize("case", [("conv", 2, 64, 64, 32, 64, 3, 1), ("conv", 2, 128, 128, 64, 128, 3, 2), ("convT", 2, 32, 32, 128, 64, 3, 2), ("conv", 16, 64, 64, 256, 256, 3, 1)])
def test_bf16_mode_stores_the_convolution_output_as_bf16(case):
    """bf16 operand mode, nirgan_conv_desc.out_bf16 + nirgan_in_fwd_desc / nirgan_in_bwd_desc.y_bf16: the convolution stores its output
    rounded to nearest-even bf16 -- bitwise the rounding of what the same launch stores as fp32 (OPT.bf16_y = False) -- while mean / rstd
    come from the fp32 accumulators (bitwise the same in both forms); the norm's apply and the two passes of its backward then use the
    rounded tensor: checked against torch on the widened values with the device's own statistics."""
    from nirgan_hip.engine import ConvIN, SlabPool, Weights, _Scratch
    from nirgan_hip.options import OPT
    kind, B, H, W, Cin, Cout, k, s_ = case

    class Eng:
        pass

    def build(y16):
        OPT.bf16_y = y16
        OPT.bf16_store_min_tiles = 0         # small launches too (the engines keep fp32 below 400 output tiles: split-K territory)
        try:
            ctx = Ctx(DEV, "bf16")
            g = torch.Generator().manual_seed(5)
            eng = Eng()
            eng.ctx, eng.weights, eng.scratch, eng.slabs, eng.need_backward = ctx, Weights(ctx), _Scratch(ctx), SlabPool(ctx), True
            inp_layer_out = Halo(ctx, B, H, W, Cin, 1, twin=True)
            x = torch.randn(inp_layer_out.t.shape, generator=g).to(DEV)
            inp_layer_out.t.copy_(x)
            inp_layer_out.t16.copy_(x.to(torch.bfloat16))
            wshape = (Cin, Cout, k, k) if kind == "convT" else (Cout, Cin, k, k)
            w = (torch.randn(wshape, generator=g) * 0.05).to(DEV)
            bias = (torch.randn(Cout, generator=g) * 3.0).to(DEV)
            layer = ConvIN(eng, "t", kind, inp_layer_out, w, bias, k=k, s=s_, p=1, cout=Cout, norm=True, act=L.ACT_RELU, out_pad=1, out_border=L.BORDER_KEEP)
            plan, pack = Plan(ctx), Plan(ctx)
            layer.emit_fwd(plan, pack)
            layer.alloc_bwd(need_dgrad=False)
            gw = ctx.zeros(*wshape)
            gh = Halo(ctx, B, layer.OH, layer.OW, Cout, 0, bf16=y16)           # the gradient arrives as bf16 too (nirgan_in_bwd_desc.g_bf16)
            gh.t.copy_(torch.randn(gh.t.shape, generator=g).to(DEV))
            bplan = Plan(ctx)
            layer.emit_bwd(bplan, pack, g=gh, gw=gw, gb=None, dgrad_out=None)
            pack.run()
            plan.run()
            bplan.run()
            torch.cuda.synchronize()
            layer.bplan = bplan
            return layer, gh
        finally:
            OPT.reset()

    l16, gh = build(True)
    l32, _ = build(False)
    assert l16.y.t.dtype == torch.bfloat16 and l32.y.t.dtype == torch.float32 and gh.t.dtype == torch.bfloat16
    assert [a[0]._obj.g_bf16 for n, a in l16.bplan.ops if n == "nirgan_instnorm_bwd"] == [1]
    assert torch.equal(l16.y.t, l32.y.t.to(torch.bfloat16)), "the bf16 store is not the rounding of the fp32 store"
    assert torch.equal(l16.stats[0], l32.stats[0]) and torch.equal(l16.stats[1], l32.stats[1]), "statistics must come from the fp32 accumulators"
    yq = l16.y.t.double()
    mean, rstd = l16.stats[0].double()[:, None, None, :], l16.stats[1].double()[:, None, None, :]
    z = (yq - mean) * rstd
    out = l16.out.t16 if l16.out.t16 is not None else l16.out.t
    close(out[:, 1:-1, 1:-1].double(), torch.relu(z).to(torch.bfloat16).double(), 2.0 ** -7, "apply on the rounded tensor")
    gz = torch.where(z > 0, gh.t.double(), torch.zeros((), dtype=torch.float64, device=DEV))
    dy = rstd * (gz - gz.mean((1, 2), keepdim=True) - z * (gz * z).mean((1, 2), keepdim=True))
    P = l16.dy.pad
    got = (l16.dy.t16 if l16.dy.t16 is not None else l16.dy.t)
    got = got[:, P:got.shape[1] - P, P:got.shape[2] - P].double()
    err = ((got - dy).norm() / dy.norm()).item()
    assert err < 4e-3, f"instance-norm backward on the rounded tensor: {err:.2e}"          # bf16 twin of dy: 2^-9 per element


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 8, "plain", L.ACT_RELU), (2, 32, 16, 64, 64, "plain", L.ACT_LRELU), (3, 16, 16, 32, 128, "plain", L.ACT_RELU),
                                  (2, 16, 8, 64, 256, "plain", L.ACT_NONE), (2, 32, 32, 16, 64, "phases", L.ACT_RELU), (2, 32, 32, 32, 128, "phases", L.ACT_LRELU)])
def test_conv_epilogue_takes_the_instance_norm_backward_first_pass(case):
    """nirgan_conv_desc.fuse_*: a launch that writes the gradient wrt a ConvIN layer's output leaves, per 128-pixel tile, the sums of
    g_z = g act'(z) and g_z z (csrc/igemm_tiles.h, conv_tile epilogue) -- against numpy on the stored g, for the 64- and 128-wide tiles,
    two N tiles, and the four sub-pixel phases of a stride-2 data gradient numbering their chunks into one workspace; then
    nirgan_instnorm_bwd with sums_chunks against its own first pass."""
    B, H, W, Cin, N, mode, act = case
    gen = torch.Generator().manual_seed(31)
    ctx = Ctx(DEV)
    k = 3
    y = (torch.randn(B, H, W, N, generator=gen) * 1.5 + 0.3).to(DEV)                  # the layer's pre-normalisation output
    mean = y.mean((1, 2)).contiguous()
    rstd = (1.0 / torch.sqrt(y.var((1, 2), unbiased=False) + 1e-5)).contiguous()
    g = Halo(ctx, B, H, W, N, 1)                                                     # the gradient buffer the launches write (zero halo)
    chunks = H * W // 128
    part = ctx.zeros(B * chunks * 2 * N + B * 2 * N)
    descs = []
    if mode == "plain":
        x = Halo(ctx, B, H, W, Cin, 1)
        x.t.copy_(torch.randn(x.t.shape, generator=gen))
        spec = G.conv_fwd_pack(N, Cin, k)
        w = (torch.randn(N, Cin, k, k, generator=gen) * 0.05).to(DEV)
        wp = ctx.zeros(spec.N, spec.K)
        L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
        descs.append(emit_conv(None, ctx, x, G.conv_fwd_taps(k, Cin), wp, None, g, N=N, OH=H, OW=W, out_oh=1, out_ow=1, allow_split=False))
    else:
        # the data gradient of a 3x3 stride-2 convolution N -> Cin: dy is (H/2, W/2, Cin), the four phases interleave into g
        dy = Halo(ctx, B, H // 2, W // 2, Cin, 1)
        dy.t[:, 1:-1, 1:-1].copy_(torch.randn(B, H // 2, W // 2, Cin, generator=gen))
        w = (torch.randn(Cin, N, k, k, generator=gen) * 0.05).to(DEV)
        for ph in G.conv_dgrad_s2_phases(H, W, k, 1):
            spec = G.conv_dgrad_pack(Cin, N, k, ph.taps_hw)
            wp = ctx.zeros(spec.N, spec.K)
            ctx.keep.append(wp)
            L.call("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
            descs.append(emit_conv(None, ctx, dy, G.Taps(ph.dh, ph.dw, Cin), wp, None, g, N=N, OH=ph.n_h, OW=ph.n_w, in_oh=ph.in_oh, in_ow=ph.in_ow,
                                   out_stride=2, out_oh=ph.out_oh + 1, out_ow=ph.out_ow + 1))
    first = 0
    for d in descs:
        d.fuse_y, d.fuse_mean, d.fuse_rstd = y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        d.fuse_h, d.fuse_w, d.fuse_oh, d.fuse_ow = H, W, d.out_oh - 1, d.out_ow - 1
        d.fuse_act, d.fuse_slope = act, 0.2
        d.fuse_part, d.fuse_part_elems, d.fuse_chunk0, d.fuse_chunks = part.data_ptr(), part.numel(), first, chunks
        first += d.OH * d.OW // 128
    assert first == chunks
    st = torch.cuda.current_stream().cuda_stream
    if mode == "plain":
        L.call("nirgan_conv_igemm", C.byref(descs[0]), st)
    else:
        arr = (C.POINTER(L.ConvDesc) * 4)(*[C.pointer(d) for d in descs])
        L.call("nirgan_conv_igemm_group", arr, 4, st)
    torch.cuda.synchronize()
    gi = g.t[:, 1:-1, 1:-1].double().cpu()
    z = ((y - mean[:, None, None]) * rstd[:, None, None]).cpu()
    neg = {L.ACT_NONE: 1.0, L.ACT_RELU: 0.0, L.ACT_LRELU: 0.2}[act]
    gz = torch.where(z > 0, gi, gi * neg)
    got = part[:B * chunks * 2 * N].view(B, chunks, 2, N).double().cpu().sum(1)
    scale = gz.abs().sum((1, 2)).max().item()
    assert (got[:, 0] - gz.sum((1, 2))).abs().max().item() < 2e-6 * scale
    assert (got[:, 1] - (gz * z.double()).sum((1, 2))).abs().max().item() < 4e-6 * scale
    # the consumer: nirgan_instnorm_bwd with the sums in place against its own first pass
    outs = []
    for pre in (chunks, 0):
        dyo = Halo(ctx, B, H, W, N, 1)
        ws = part if pre else ctx.zeros(L.backend().nirgan_instnorm_ws_elems(B, H, W, N))
        plan = Plan(ctx)
        yh = Halo(ctx, B, H, W, N, 0, tensor=y)
        emit_in_bwd(plan, ctx, g=g, g_fold=False, act=act, y=yh, stats=(mean, rstd), norm=True, dy=dyo, ws=ws, shape=(B, H, W, N), pre_sums=pre)
        plan.run()
        outs.append(dyo.t.clone())
    torch.cuda.synchronize()
    close(outs[0], outs[1], 2e-5, "dy with the sums from the epilogue")


@pytest.mark.parametrize("case", [(2, 32, 16, 64, 64), (2, 32, 32, 64, 128), (3, 16, 16, 128, 64)])
def test_conv_epilogue_first_pass_with_bf16_output(case):
    """The same fused first pass when the launch stores its output as bf16 and reads a bf16 y (bf16 operand mode: nirgan_conv_desc.out_bf16,
    fuse_y_bf16): the 128-row tile's epilogue then hands a lane EIGHT channels (16-byte stores and y loads).  Against the fp32-output form
    of the same launch (four channels per lane): the stored gradient is bitwise the rounding of the fp32 store, the partial sums -- taken
    from the fp32 accumulators either way, by different threads in a different row order -- agree to fp32 rounding."""
    B, H, W, Cin, N = case
    gen = torch.Generator().manual_seed(37)
    ctx = Ctx(DEV, "bf16")
    k = 3
    y = (torch.randn(B, H, W, N, generator=gen) * 1.5 + 0.3).to(DEV).to(torch.bfloat16)
    yf = y.float()
    mean = yf.mean((1, 2)).contiguous()
    rstd = (1.0 / torch.sqrt(yf.var((1, 2), unbiased=False) + 1e-5)).contiguous()
    x = Halo(ctx, B, H, W, Cin, 1, twin=True)
    xv = torch.randn(x.t.shape, generator=gen).to(DEV)
    x.t.copy_(xv)
    x.t16.copy_(xv.to(torch.bfloat16))
    spec = G.conv_fwd_pack(N, Cin, k)
    w = (torch.randn(N, Cin, k, k, generator=gen) * 0.05).to(DEV)
    wp = torch.zeros(spec.N, spec.K, device=DEV, dtype=torch.bfloat16)
    L.call("nirgan_pack_rows_bf16", w.data_ptr(), w.numel(), spec.row_stride, ctx.i32(spec.index_map).data_ptr(), wp.data_ptr(), spec.N, spec.K, None)
    chunks = H * W // 128
    res = []
    for out16 in (True, False):
        g = Halo(ctx, B, H, W, N, 1, bf16=out16)
        part = ctx.zeros(B * chunks * 2 * N)
        d = emit_conv(None, ctx, x, G.conv_fwd_taps(k, Cin), wp, None, g, N=N, OH=H, OW=W, out_oh=1, out_ow=1, allow_split=False)
        assert d.in_bf16 == 1 and d.w_bf16 == 1 and d.out_bf16 == (1 if out16 else 0)
        d.algo = L.CONV_TILE128
        d.fuse_y, d.fuse_y_bf16, d.fuse_mean, d.fuse_rstd = y.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr()
        d.fuse_h, d.fuse_w, d.fuse_oh, d.fuse_ow = H, W, 0, 0
        d.fuse_act, d.fuse_slope = L.ACT_RELU, 0.2
        d.fuse_part, d.fuse_part_elems, d.fuse_chunk0, d.fuse_chunks = part.data_ptr(), part.numel(), 0, chunks
        L.call("nirgan_conv_igemm", C.byref(d), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res.append((g.t[:, 1:-1, 1:-1].clone(), part.view(B, chunks, 2, N).clone()))
    (g16, p16), (g32, p32) = res
    assert g16.dtype == torch.bfloat16 and torch.equal(g16, g32.to(torch.bfloat16)), "the bf16 store is not the rounding of the fp32 store"
    scale = p32.abs().max().item()
    assert (p16 - p32).abs().max().item() < 1e-5 * scale, f"partial sums differ by {(p16 - p32).abs().max().item():.3e} at scale {scale:.3e}"
    # and against torch on the fp32 gradient
    z = (yf - mean[:, None, None]) * rstd[:, None, None]
    gz = torch.where(z > 0, g32.double(), torch.zeros((), dtype=torch.float64, device=DEV))
    tot = p16.double().sum(1)
    ref_scale = gz.abs().sum((1, 2)).max().item()
    assert (tot[:, 0] - gz.sum((1, 2))).abs().max().item() < 4e-6 * ref_scale
    assert (tot[:, 1] - (gz * z.double()).sum((1, 2))).abs().max().item() < 8e-6 * ref_scale


def test_wino6_plane_gemm_stage_depths_agree(monkeypatch):
    """The plane GEMM kernels a descriptor can select (nirgan_wino6_desc.algo): persistent workgroups on 32-k stages (default) and on
    16-k stages, one tile per workgroup, the direct convolution tile -- the same products in the same k order: equal to fp32 rounding
    on the benchmark's residual-block shape and on a ragged one; the name query reports the kernel each one launches."""
    import ctypes as C
    for (B, H, W, Cc, K) in ((16, 64, 64, 256, 256), (3, 21, 17, 256, 192)):
        g = torch.Generator().manual_seed(9)
        T = B * (-(-H // 6)) * (-(-W // 6))
        V = torch.randn(64 * T * Cc, generator=g).to(DEV)
        U = (torch.randn(64 * K * Cc, generator=g) * 0.05).to(DEV)
        zero = torch.zeros(64, device=DEV)
        outs = []
        names = []
        for algo in (0, L.W6_PERSIST16, L.W6_ONE_TILE, L.W6_DIRECT_TILE, L.W6_TILE256):
            M = torch.full((64 * T * K,), float("nan"), device=DEV)
            d = L.Wino6Desc()
            d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, K
            d.U, d.V, d.V_elems, d.M, d.M_elems, d.zero_page = U.data_ptr(), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), zero.data_ptr()
            d.algo = algo
            names.append(L.backend().nirgan_wino6_gemm_kernel_name(C.byref(d)).decode())
            L.call("nirgan_wino6_gemm", C.byref(d), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs.append(M.clone())
        assert names == ["wino6_gemm32p_kernel", "wino6_gemm16p_kernel", "wino6_gemm16_kernel", "wino6_gemm_kernel",
                         "wino6_gemm256_kernel" if K % 256 == 0 else "wino6_gemm32p_kernel"], names
        ref = torch.bmm(V.view(64, T, Cc).double(), U.view(64, K, Cc).double().transpose(1, 2)).float().reshape(-1)
        close(outs[0], ref, 2e-5, "32-k stages vs fp64")
        for o, n in zip(outs[1:], names[1:]):
            close(o, outs[0], 2e-6, n + " vs 32-k persistent")


@pytest.mark.parametrize("case", [(6, 3, 2, 256, 256), (4, 2, 1, 128, 64), (3, 16, 2, 32, 32)])
def test_wino6_weight_gradient_finish_batch(case):
    """nirgan_wino6_wgrad_finish_batch (n layers of one geometry in one grid) against n single nirgan_wino6_wgrad_finish_r calls: bitwise,
    with and without accumulation; and the numpy restatement."""
    import ctypes as C
    v, n, nsplit, K, Cc = case
    r = 3 if v == 6 else v
    NP = 64 if v == 6 else (v + 3) ** 2
    g = torch.Generator().manual_seed(4)
    slabs = [torch.randn(NP * nsplit * K * Cc, generator=g).to(DEV) for _ in range(n)]
    st = torch.cuda.current_stream().cuda_stream
    for acc in (0, 1):
        base = [torch.randn(K * Cc * r * r, generator=g).to(DEV) for _ in range(n)]
        one = [b.clone() for b in base]
        many = [b.clone() for b in base]
        for s_, o in zip(slabs, one):
            L.call("nirgan_wino6_wgrad_finish_r", s_.data_ptr(), nsplit, K, Cc, v, o.data_ptr(), acc, st)
        sp = (C.c_void_p * n)(*[s_.data_ptr() for s_ in slabs])
        gp = (C.c_void_p * n)(*[m.data_ptr() for m in many])
        L.call("nirgan_wino6_wgrad_finish_batch", sp, gp, n, nsplit, K, Cc, v, acc, st)
        torch.cuda.synchronize()
        for o, m in zip(one, many):
            assert torch.equal(o, m)
    if K * Cc <= 8192:
        emu = EmuBackend()
        sc = [s_.cpu() for s_ in slabs]
        oc = [torch.zeros(K * Cc * r * r) for _ in range(n)]
        sp = (C.c_void_p * n)(*[s_.data_ptr() for s_ in sc])
        gp = (C.c_void_p * n)(*[o.data_ptr() for o in oc])
        assert emu.nirgan_wino6_wgrad_finish_batch(sp, gp, n, nsplit, K, Cc, v, 0) == 0
        fresh = [torch.zeros(K * Cc * r * r, device=DEV) for _ in range(n)]
        gp2 = (C.c_void_p * n)(*[f.data_ptr() for f in fresh])
        sp2 = (C.c_void_p * n)(*[s_.data_ptr() for s_ in slabs])
        L.call("nirgan_wino6_wgrad_finish_batch", sp2, gp2, n, nsplit, K, Cc, v, 0, st)
        torch.cuda.synchronize()
        for o, f in zip(oc, fresh):
            close(f, o, 1e-5, "device vs restatement")


@pytest.mark.parametrize("case", [(2, 3, 70, 90, 32, 8, 5), (1, 3, 256, 256, 128, 16, 3), (1, 1, 33, 47, 16, 2, 64), (3, 4, 40, 24, 24, 0, 4)])
def test_tile_gather_and_scatter_match_pad_slice_stack(case):
    """nirgan_tile_gather / nirgan_tile_scatter (tiled inference) against torch: F.pad(mode='reflect') of the scene, slices
    of overlapping tiles, and the tiles' cores written back -- bit for bit (pure data movement), in batches that do not divide the tile
    count; a scene too small for its reflected border is refused."""
    B, Cc, H, W, tile, margin, batch = case
    g = torch.Generator().manual_seed(17)
    scene = torch.randn(B, Cc, H, W, generator=g)
    core = tile - 2 * margin
    ph, pw = (-H) % core, (-W) % core
    xp = torch.nn.functional.pad(scene, (margin, margin + pw, margin, margin + ph), mode="reflect")
    coords = [(b, i, j) for b in range(B) for i in range(0, H + ph, core) for j in range(0, W + pw, core)]
    want = torch.stack([xp[b, :, i:i + tile, j:j + tile] for b, i, j in coords])
    total = int(L.backend().nirgan_tile_count(B, H, W, tile, margin))
    assert total == len(coords)
    sd = scene.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    got = torch.full((total, Cc, tile, tile), float("nan"), device=DEV)
    back = torch.full((B, Cc, H, W), float("nan"), device=DEV)
    for first in range(0, total, batch):
        n = min(batch, total - first)
        L.call("nirgan_tile_gather", sd.data_ptr(), B, Cc, H, W, tile, margin, first, n, got[first:].data_ptr(), st)
        L.call("nirgan_tile_scatter", got[first:].data_ptr(), B, Cc, H, W, tile, margin, first, n, back.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want), "gathered tiles differ from pad + slice"
    assert torch.equal(back.cpu(), scene), "scattering the tiles' cores does not reproduce the scene"
    with pytest.raises(RuntimeError):
        L.call("nirgan_tile_gather", sd.data_ptr(), B, Cc, H, W, 4 * max(H, W), max(H, W), 0, 1, got.data_ptr(), st)


@pytest.mark.parametrize("case", [("fold", 12, 16, 64), ("skip", 12, 16, 64), ("fold", 64, 64, 256), ("skip", 64, 64, 256), ("skip", 9, 11, 32),
                                  ("pre", 64, 64, 256)])
def test_wino6_dy_transform_with_the_instance_norm_backward_folded_in(case):
    """nirgan_instnorm_bwd(dy = NULL: the two reductions only) + nirgan_wino6_input_dy_norm = the V / Yt that the full instance-norm
    backward (which stores dY) followed by nirgan_wino6_input_dy produces -- bitwise: the second pass's arithmetic is evaluated inside
    the lane-spread transform and dY never exists in memory.  Block kinds of the residual chain: first convolution (halo'd gradient
    folded through the reflect padding, ReLU mask), second convolution (dense skip-path sum from pass 1, no activation), and 'pre' -- the
    folded gradient dense in gsum_out with the first pass's partial sums already in ws, as the fused output transform leaves them."""
    from nirgan_hip.engine import emit_in_bwd, emit_in_fwd
    kind, H, W, Cc = case
    B = 2
    g = torch.Generator().manual_seed(33)
    ctx = Ctx(DEV, "fp32")
    st = torch.cuda.current_stream().cuda_stream
    yh = Halo(ctx, B, H, W, Cc, 0)
    yh.t.copy_((torch.randn(B, H, W, Cc, generator=g) * 1.3 + 0.2).to(DEV))
    out = Halo(ctx, B, H, W, Cc, 1)
    stats = (ctx.zeros(B, Cc), ctx.zeros(B, Cc))
    ws = ctx.zeros(int(L.backend().nirgan_instnorm_ws_elems(B, H, W, Cc)) + 64 * 2 * B * Cc)
    f = Plan(ctx)
    emit_in_fwd(f, ctx, yh, out, norm=True, act=L.ACT_RELU, border=L.BORDER_REFLECT, stats=stats, ws=ws)
    f.run()
    gh = Halo(ctx, B, H, W, Cc, 1)
    gh.t.copy_(torch.randn(B, H + 2, W + 2, Cc, generator=g).to(DEV))
    g2 = Halo(ctx, B, H, W, Cc, 0)
    g2.t.copy_(torch.randn(B, H, W, Cc, generator=g).to(DEV))
    gsum = Halo(ctx, B, H, W, Cc, 0)
    dy = Halo(ctx, B, H, W, Cc, 2)
    kw = dict(g=gh, g_fold=True, y=yh, stats=stats, norm=True, dy=dy, ws=ws, shape=(B, H, W, Cc))
    if kind == "skip":
        kw.update(g2=g2, gsum=gsum, act=L.ACT_NONE)
    elif kind == "pre":
        chunks = 16
        gsum.t.copy_(torch.randn(B, H, W, Cc, generator=g).to(DEV))
        ws[:B * chunks * 2 * Cc].copy_(torch.randn(B * chunks * 2 * Cc, generator=g).to(DEV))
        kw.update(gsum=gsum, act=L.ACT_RELU, pre_sums=chunks)
    else:
        kw.update(act=L.ACT_RELU)
    full, sums = Plan(ctx), Plan(ctx)
    emit_in_bwd(full, ctx, **kw)
    nd = emit_in_bwd(sums, ctx, sums_only=True, **kw)
    Td, Ty = B * (-(-(H + 2) // 6)) * (-(-(W + 2) // 6)), B * (-(-H // 6)) * (-(-W // 6))
    res = []
    for fused in (False, True):
        V = torch.full((64 * Td * Cc,), float("nan"), device=DEV)
        Yt = torch.full((64 * Ty * Cc,), float("nan"), device=DEV)
        d = L.Wino6Desc()
        d.r, d.x, d.x_hp, d.x_wp, d.B, d.H, d.W, d.C, d.K = 6, dy.ptr, H + 4, W + 4, B, H + 2, W + 2, Cc, Cc
        d.V, d.V_elems = V.data_ptr(), V.numel()
        yd = L.WinoDyDesc()
        yd.dy, yd.dy_hp, yd.dy_wp, yd.dy_pad, yd.B, yd.H, yd.W, yd.K = dy.ptr, H + 4, W + 4, 2, B, H, W, Cc
        yd.Yt, yd.Yt_elems, yd.r = Yt.data_ptr(), Yt.numel(), 6
        if fused:
            dy.t.fill_(float("nan"))          # the fused pass must not read the buffer
            sums.run()
            L.call("nirgan_wino6_input_dy_norm", C.byref(d), C.byref(yd), C.byref(nd), st)
        else:
            full.run()
            L.call("nirgan_wino6_input_dy", C.byref(d), C.byref(yd), st)
        torch.cuda.synchronize()
        res.append((V.cpu(), Yt.cpu()))
    assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0]), "V differs"
    assert torch.isfinite(res[0][1]).all() and torch.equal(res[0][1], res[1][1]), "Yt differs"


@pytest.mark.parametrize("shape", [(2, 12, 16, 64), (16, 64, 64, 256), (3, 9, 11, 32), (1, 21, 17, 96)])
def test_wino6_input_transform_forms_agree_bitwise(shape):
    """F(6x6,3x3) input transforms: the patch-per-thread kernel and the lane-spread kernel (one wave per patch x 32 channels, transpose
    through LDS) run the same arithmetic in the same order -- V, and Yt of the dY pass, bit for bit; so does the normalising variant."""
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(41)
    st = torch.cuda.current_stream().cuda_stream
    T = B * (-(-(H + 2) // 6)) * (-(-(W + 2) // 6))            # data-gradient extent (H + 2) x (W + 2) over dY with a zero halo of 2
    dy = torch.zeros(B, H + 4, W + 4, Cc)
    dy[:, 2:-2, 2:-2] = torch.randn(B, H, W, Cc, generator=g)
    dy = dy.to(DEV)
    yT = B * (-(-H // 6)) * (-(-W // 6))
    outs = []
    for algo in (0, L.W6_PATCH_PER_THREAD):
        V = torch.full((64 * T * Cc,), float("nan"), device=DEV)
        Yt = torch.full((64 * yT * Cc,), float("nan"), device=DEV)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H + 2, W + 2, Cc, Cc
        d.x, d.x_hp, d.x_wp, d.V, d.V_elems, d.algo = dy.data_ptr(), H + 4, W + 4, V.data_ptr(), V.numel(), algo
        y = L.WinoDyDesc()
        y.dy, y.dy_hp, y.dy_wp, y.dy_pad, y.B, y.H, y.W, y.K = dy.data_ptr(), H + 4, W + 4, 2, B, H, W, Cc
        y.Yt, y.Yt_elems, y.r = Yt.data_ptr(), Yt.numel(), 6
        L.call("nirgan_wino6_input_dy", C.byref(d), C.byref(y), st)
        torch.cuda.synchronize()
        outs.append((V.cpu(), Yt.cpu()))
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # normalising variant: y dense [B][H][W][C], reflect halo of 1, ReLU
    yv = (torch.randn(B, H, W, Cc, generator=g) * 1.3 + 0.2).to(DEV)
    mean, rstd = (torch.randn(B, Cc, generator=g) * 0.1).to(DEV), (torch.rand(B, Cc, generator=g) + 0.5).to(DEV)
    Tn = B * (-(-H // 6)) * (-(-W // 6))
    res = []
    for algo in (0, L.W6_PATCH_PER_LANES):
        V = torch.full((64 * Tn * Cc,), float("nan"), device=DEV)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, Cc, Cc
        d.x_hp, d.x_wp, d.V, d.V_elems, d.algo = H + 2, W + 2, V.data_ptr(), V.numel(), algo
        L.call("nirgan_wino6_input_norm", C.byref(d), yv.data_ptr(), mean.data_ptr(), rstd.data_ptr(), L.ACT_RELU, 0.2, st)
        torch.cuda.synchronize()
        res.append(V.cpu())
    assert torch.isfinite(res[0]).all() and torch.equal(res[0], res[1])


@pytest.mark.parametrize("kind", ["conv", "wino6"])
def test_producer_partial_sums_do_not_cancel_with_a_large_channel_mean(kind):
    """The instance-norm statistics a producer leaves (convolution epilogue / Winograd output transform: per chunk {k, sum (v - k),
    sum (v - k)^2, count} about a value of the chunk itself) for channels whose mean is ~1e3 times their spread: rstd from the merged
    chunks within 1e-4 of float64 (sums about the bias alone lose the variance entirely there: relative error ~1e-7 (mean / std)^2)."""
    from nirgan_hip.engine import ConvIN, Weights, _Scratch, SlabPool
    from nirgan_hip.options import OPT
    B, H, W, Cin, Cout = (16 if kind == "conv" else 2), 64, 64, 128, 128        # (the direct tile splits K below ~400 tiles: no epilogue sums then)
    g = torch.Generator().manual_seed(77)

    class Eng:
        pass
    OPT.epilogue_min_pixels = 0
    OPT.winograd = "f6" if kind == "wino6" else "off"
    try:
        ctx = Ctx(DEV, "fp32")
        eng = Eng()
        eng.ctx, eng.weights, eng.scratch, eng.slabs, eng.need_backward = ctx, Weights(ctx), _Scratch(ctx), SlabPool(ctx), False
        inp = Halo(ctx, B, H, W, Cin, 1)
        # a constant per input channel plus a small texture: every output channel gets a large mean (sum of w * const) and a small spread
        x = 5.0 + 0.01 * torch.randn(inp.t.shape, generator=g)
        inp.t.copy_(x.to(DEV))
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05 + 0.02).to(DEV)
        bias = torch.zeros(Cout, device=DEV)
        layer = ConvIN(eng, "t", "conv", inp, w, bias, k=3, s=1, p=1, cout=Cout, norm=True, act=L.ACT_RELU, out_pad=1, out_border=L.BORDER_REFLECT)
        plan, pack = Plan(ctx), Plan(ctx)
        layer.emit_fwd(plan, pack)
        pack.run()
        plan.run()
        torch.cuda.synchronize()
        d = [a[0]._obj for n, a in plan.ops if n == "nirgan_instnorm_fwd"][0]
        assert d.stats_chunks > 0, "the producer's partial sums were not used"
        y = layer.y.t.double().cpu().reshape(B, H * W, Cout)
        ratio = (y.mean(1).abs() / y.std(1)).median().item()
        assert ratio > 300, ratio
        ref = 1.0 / torch.sqrt(y.var(1, unbiased=False) + 1e-5)
        err = ((layer.stats[1].double().cpu() - ref).abs() / ref).max().item()
        assert err < 1e-4, f"{kind}: rstd off by {err:.2e} at mean/std {ratio:.0f}"
        close(layer.stats[0], y.mean(1), 1e-6, "mean")
    finally:
        OPT.reset()


@pytest.mark.parametrize("shape", [(2, 10, 10, 64, "relu", True), (16, 64, 64, 256, "none", True), (3, 22, 16, 96, "lrelu", False), (1, 64, 34, 32, "relu", True)])
def test_wino6_fused_output_transform_forms_and_restatement(shape):
    """nirgan_wino6_output in its fused mode (output transform of a data gradient over the padded extent + reflect fold + skip gradient +
    the partial sums of the consumer's first instance-norm backward pass): the tile-per-thread kernel, the lane-spread kernel and the
    numpy restatement.  The folded gradient g_a agrees bit for bit between the two device forms; the tile sums to fp32 rounding."""
    B, Hi, Wi, K, actn, skip = shape
    act = {"relu": L.ACT_RELU, "lrelu": L.ACT_LRELU, "none": L.ACT_NONE}[actn]
    H, W = Hi + 2, Wi + 2                                   # padded extent of the data gradient
    assert (H - 3) // 6 == (H - 1) // 6 and (W - 3) // 6 == (W - 1) // 6
    TH, TW = -(-H // 6), -(-W // 6)
    T = B * TH * TW
    g = torch.Generator().manual_seed(19)
    M = torch.randn(64 * T * K, generator=g)
    y = torch.randn(B, Hi, Wi, K, generator=g)
    g2 = torch.randn(B, Hi, Wi, K, generator=g) if skip else None
    mean, rstd = torch.randn(B, K, generator=g) * 0.2, torch.rand(B, K, generator=g) + 0.5
    outs = []
    emu = EmuBackend()
    for mode in ("lanes", "thread", "emu"):
        dev = "cpu" if mode == "emu" else DEV
        bufs = [t_.to(dev) if t_ is not None else None for t_ in (M, y, g2, mean, rstd)]
        gz = torch.full((B, Hi, Wi, K), float("nan"), device=dev)
        part = torch.full((T * 2 * K,), float("nan"), device=dev)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = 6, B, H, W, K, K
        d.M, d.M_elems = bufs[0].data_ptr(), bufs[0].numel()
        d.fuse_y, d.fuse_mean, d.fuse_rstd = bufs[1].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr()
        d.fuse_g2 = bufs[2].data_ptr() if skip else None
        d.fuse_gz, d.fuse_part, d.fuse_part_elems, d.fuse_act, d.fuse_slope = gz.data_ptr(), part.data_ptr(), part.numel(), act, 0.2
        d.algo = L.W6_PATCH_PER_THREAD if mode == "thread" else 0
        if mode == "emu":
            assert emu.nirgan_wino6_output(d) == 0
        else:
            L.call("nirgan_wino6_output", C.byref(d), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        outs.append((gz.cpu(), part.cpu().view(T, 2, K), bufs))
    (gl, pl, _), (gt, pt, _), (ge, pe, _) = outs
    assert torch.isfinite(gl).all() and torch.isfinite(pl).all()
    assert torch.equal(gl, gt), "folded gradient differs between the two device forms"
    close(pl, pt, 2e-6, "tile sums, lanes vs thread")
    close(gl, ge, 2e-5, "folded gradient vs restatement")
    close(pl, pe, 2e-5, "tile sums vs restatement")


@pytest.mark.parametrize("case", [(6, 2, 12, 16, 64, 128), (6, 16, 64, 64, 256, 256), (6, 3, 9, 11, 32, 96), (4, 2, 9, 13, 64, 128), (4, 32, 31, 31, 256, 512), (4, 1, 6, 5, 32, 32)])
def test_wino6_input_transform_forms_agree_for_both_patch_sizes(case):
    """The lane-spread form of the input transform / dY pass against the patch-per-thread kernels (nirgan_wino6_desc.algo) for the 8 x 8
    patches of F(6x6,3x3) and the 7 x 7 patches of F(4x4,4x4) (the eighth lane row idles): V and Yt bit for bit."""
    v, B, H, W, Cc, K = case
    mo, r = (6, 3) if v == 6 else (4, 4)
    n = mo + r - 1
    g = torch.Generator().manual_seed(23)
    st = torch.cuda.current_stream().cuda_stream
    # ---- input + dY pass: dY [B][H][W][K] with a zero halo of r - 1; the data gradient covers (H + r - 1) x (W + r - 1)
    He, We = H + r - 1, W + r - 1
    T = B * (-(-He // mo)) * (-(-We // mo))
    yT = B * (-(-H // mo)) * (-(-W // mo))
    dy = torch.zeros(B, H + 2 * (r - 1), W + 2 * (r - 1), K)
    dy[:, r - 1:r - 1 + H, r - 1:r - 1 + W] = torch.randn(B, H, W, K, generator=g)
    dy = dy.to(DEV)
    res = []
    for algo in (0, L.W6_PATCH_PER_THREAD):
        V = torch.full((n * n * T * K,), float("nan"), device=DEV)
        Yt = torch.full((n * n * yT * K,), float("nan"), device=DEV)
        d = L.Wino6Desc()
        d.r, d.B, d.H, d.W, d.C, d.K = v, B, He, We, K, Cc
        d.x, d.x_hp, d.x_wp, d.V, d.V_elems, d.algo = dy.data_ptr(), dy.shape[1], dy.shape[2], V.data_ptr(), V.numel(), algo
        y = L.WinoDyDesc()
        y.dy, y.dy_hp, y.dy_wp, y.dy_pad, y.B, y.H, y.W, y.K = dy.data_ptr(), dy.shape[1], dy.shape[2], r - 1, B, H, W, K
        y.Yt, y.Yt_elems, y.r = Yt.data_ptr(), Yt.numel(), v
        L.call("nirgan_wino6_input_dy", C.byref(d), C.byref(y), st)
        torch.cuda.synchronize()
        res.append((V.cpu(), Yt.cpu()))
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][1]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_reduce_rows_part_folds_the_bands_of_a_slab():
    """nirgan_reduce_rows_part: a band of the slabs' rows summed over the splits into dst rows 0.. -- bitwise the same sums as
    nirgan_reduce_rows over a slab that holds only that band; the two pixel-parity bands of the first convolution's weight gradient
    (geometry.conv_rowpacked_pair_pack: column (kh, px, c) of parity q is weight element (c, kh, px - q)) folded into one tensor equal the
    float64 fold."""
    g = torch.Generator().manual_seed(19)
    cout, cin, k, cs, nsplit = 64, 3, 7, 4, 11
    specs = [G.conv_rowpacked_pair_pack(cout, cin, k, cs, q) for q in (0, 1)]
    K = specs[0].K
    slabs = torch.randn(nsplit, 2 * cout, K, generator=g).to(DEV)
    out = torch.full((cout * cin * k * k,), float("nan"), device=DEV)
    for q, spec in enumerate(specs):
        imap = torch.from_numpy(spec.index_map).to(DEV)
        L.call("nirgan_reduce_rows_part", slabs.data_ptr(), nsplit, 2 * cout, q * cout, cout, K, imap.data_ptr(), out.data_ptr(), out.numel(), spec.row_stride, q, None)
        band = slabs[:, q * cout:(q + 1) * cout].contiguous()
        alone = torch.zeros_like(out)
        L.call("nirgan_reduce_rows", band.data_ptr(), nsplit, cout, K, imap.data_ptr(), alone.data_ptr(), alone.numel(), spec.row_stride, 0, None)
        only = torch.zeros_like(out)
        L.call("nirgan_reduce_rows_part", slabs.data_ptr(), nsplit, 2 * cout, q * cout, cout, K, imap.data_ptr(), only.data_ptr(), only.numel(), spec.row_stride, 0, None)
        torch.cuda.synchronize()
        assert torch.equal(only, alone), f"band {q}"
    ref = np.zeros((cout, cin * k * k))
    s64 = slabs.double().sum(0).cpu().numpy()
    for q, spec in enumerate(specs):
        ok = spec.index_map >= 0
        np.add.at(ref, (np.arange(cout)[:, None], spec.index_map[ok][None, :]), s64[q * cout:(q + 1) * cout][:, ok])
    close(out.reshape(cout, -1), torch.from_numpy(ref).float(), 1e-6, "two bands folded into the weight gradient")
    assert L.backend().nirgan_reduce_rows_part(slabs.data_ptr(), nsplit, 2 * cout, 100, cout, K, imap.data_ptr(), out.data_ptr(), out.numel(), spec.row_stride, 0, None) != 0


def test_reduce_rows_batch_equals_the_single_launches():
    """nirgan_reduce_rows_batch: several weight gradients' slab sums in one launch -- the general form (any index map) and the Conv2d-layout
    form (taps > 0: whole runs of the [N][Cin][kh][kw] gradient stored contiguously) -- bitwise equal to nirgan_reduce_rows job by job
    (same split order), with and without accumulation, odd split counts."""
    g = torch.Generator().manual_seed(13)
    jobs, keep, refs, outs = [], [], [], []
    first = 0
    for (N, Cin, k, nsplit, acc, conv_layout) in ((256, 256, 3, 13, 0, True), (128, 64, 3, 5, 1, True), (64, 128, 4, 7, 0, True), (96, 36, 3, 4, 0, False), (256, 128, 3, 26, 1, False)):
        spec = G.conv_fwd_pack(N, Cin, k) if conv_layout else G.convT_dgrad_pack(N, Cin, k)
        K = spec.K
        slabs = torch.randn(nsplit, N, K, generator=g).to(DEV)
        imap = torch.from_numpy(spec.index_map).to(DEV)
        base = torch.randn(N * spec.row_stride, generator=g).to(DEV)
        ref, out = base.clone(), base.clone()
        L.call("nirgan_reduce_rows", slabs.data_ptr(), nsplit, N, K, imap.data_ptr(), ref.data_ptr(), ref.numel(), spec.row_stride, acc, None)
        T = k * k if conv_layout else 0
        jobs.append([slabs.data_ptr(), out.data_ptr(), imap.data_ptr(), nsplit, N, K, out.numel(), spec.row_stride | (acc << 32), first, T])
        first += N * ((Cin // 64) if T else ((K + 255) // 256))
        keep += [slabs, imap]
        refs.append(ref)
        outs.append(out)
    table = torch.tensor(jobs, dtype=torch.int64).to(DEV)
    L.call("nirgan_reduce_rows_batch", table.data_ptr(), len(jobs), first, None)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(outs, refs)):
        assert torch.equal(a, b), f"job {i}: batched slab sum differs from nirgan_reduce_rows (max {float((a - b).abs().max()):.3e})"
