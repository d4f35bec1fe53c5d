"""Shared by the device kernel tests and the CPU host-logic tests: one convolution layer as the engines emit it
(packed weights, forward, weight gradient with split slabs, data gradient as full correlation / sub-pixel phases)."""
from nirgan_hip import geometry as G
from nirgan_hip.engine import Halo, Plan, emit_conv, emit_wgrad

CONV_CASES = [
    # name, B, H, W, Cin, Cout, k, s, p
    ("res3x3_256", 2, 9, 11, 256, 256, 3, 1, 1),
    ("down3x3_s2", 2, 16, 12, 64, 128, 3, 2, 1),
    ("d4x4_s2", 1, 18, 18, 128, 256, 4, 2, 1),
    ("d4x4_s1_512", 1, 8, 9, 256, 512, 4, 1, 1),
    ("small_c8_n16", 3, 10, 10, 8, 16, 3, 1, 1),
    ("n64_tail", 1, 13, 7, 32, 64, 3, 1, 1),
]


def _packed(ctx, plan, w, spec):
    """Packed copy of w as the engines keep it: bf16-stored in the bf16 operand mode when the rows stay 16-byte aligned."""
    import torch
    imap = ctx.i32(spec.index_map)
    if ctx.precision == 1 and spec.run % 8 == 0 and spec.K % 8 == 0:
        buf = torch.zeros(spec.N, spec.K, dtype=torch.bfloat16, device=ctx.device)
        plan.add("nirgan_pack_rows_bf16", w.data_ptr(), w.numel(), spec.row_stride, imap.data_ptr(), buf.data_ptr(), spec.N, spec.K)
    else:
        buf = ctx.zeros(spec.N, spec.K)
        plan.add("nirgan_pack_rows", w.data_ptr(), w.numel(), spec.row_stride, imap.data_ptr(), buf.data_ptr(), spec.N, spec.K)
    ctx.keep.append(buf)
    return buf


def build_conv_case(ctx, x, w, b, case):
    """x: input Halo (pad p), w: torch-layout weight, b: bias.  Returns (forward plan, backward plan, y, dy, gw, gx);
    the caller fills dy's interior between the two plans."""
    _, B, H, W, Cin, Cout, k, s, p = case
    OH, OW = G.conv_out(H, k, s, p), G.conv_out(W, k, s, p)
    spec = G.conv_fwd_pack(Cout, Cin, k)
    taps = G.conv_fwd_taps(k, Cin)
    zpad = k - 1 if s == 1 else 1
    plan = Plan(ctx)
    wp = _packed(ctx, plan, w, spec)
    y = Halo(ctx, B, OH, OW, Cout, 0)
    emit_conv(plan, ctx, x, taps, wp, b, y, N=Cout, OH=OH, OW=OW, in_stride=s, in_oh=0, in_ow=0)
    # weight gradient with dY in a zero-halo buffer
    dy = Halo(ctx, B, OH, OW, Cout, zpad)
    gw = ctx.zeros(Cout, Cin, k, k)
    ctx.keep.append(gw)
    plan2 = Plan(ctx)
    emit_wgrad(plan2, ctx, dy, x, taps, spec, gw, N=Cout, OH=OH, OW=OW, p_oh=zpad, p_ow=zpad, q_stride=s)
    # data gradient
    if s == 1:
        gx = Halo(ctx, B, H, W, Cin, p)
        hw = [(kh, kw) for kh in range(k) for kw in range(k)]
        dspec = G.conv_dgrad_pack(Cout, Cin, k, hw)
        wd = _packed(ctx, plan2, w, dspec)
        emit_conv(plan2, ctx, dy, G.conv_dgrad_s1_taps(k, Cout), wd, None, gx, N=Cin, OH=gx.hp, OW=gx.wp)
    else:
        gx = Halo(ctx, B, H, W, Cin, 0)
        for ph in G.conv_dgrad_s2_phases(H, W, k, p):
            dspec = G.conv_dgrad_pack(Cout, Cin, k, ph.taps_hw)
            wd = _packed(ctx, plan2, w, dspec)
            emit_conv(plan2, ctx, dy, G.Taps(ph.dh, ph.dw, Cout), wd, None, gx, N=Cin, OH=ph.n_h, OW=ph.n_w,
                      in_oh=ph.in_oh, in_ow=ph.in_ow, out_stride=2, out_oh=ph.out_oh, out_ow=ph.out_ow)
    return plan, plan2, y, dy, gw, gx
