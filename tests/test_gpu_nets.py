"""Whole-path parity on the MI355X through the C ABI.

1. the committed golden vectors of the reference (small-width nets) through the fused HIP trainer;
2. full-size nets (ngf = ndf = 64, 256x256 tiles) against the CPU oracle on the same seeded tiles;
3. size-independent properties at the benchmark size (bs 16): per-sample independence of the
   batch (InstanceNorm has no cross-sample coupling, so data parallelism is exact), halo
   invariants after many steps, determinism.

Tolerances.  Forward outputs and losses: 1e-3 relative to the reference value (the bar of
BASELINE.json; measured errors are ~1e-6..1e-5).  Gradients of the small golden nets: relative
L2 error <= 1e-3 per tensor.  Gradients of the full-size nets (ngf 64) are compared with the
oracle evaluated in FP64 (the truth both fp32 implementations approximate), per tensor:
rel-L2 <= 6e-3 (generator), 2e-3 (PatchGAN) for a given smooth output gradient in the engine
tests, and <= 3e-4 for EVERY gradient tensor of the whole fused step once the oracle is made to
take the device's branch decisions (oracle.forced_kinks: test_fullsize_fused_step_against_oracle).
Why the unforced comparisons cannot be at 1e-3: the gradients are only piecewise smooth (26 M ReLU / LeakyReLU masks per tile); ONE
activation within fp32 rounding of zero flips between two correct evaluations and moves a
gradient tensor by |g_i| / ||g|| ~ 1 / sqrt(4 M) = 5e-4 of its L2 norm, and instance-norm's
backward subtracts near-equal means (cancellation).  Measured on the MI355X against fp64
(scripts/diag_grad_error2.py -> profiles/r02_grad_error_kinkfree_vs_fp64.txt): the reference's own
fp32 CPU arithmetic sits at 1.3e-3..3.2e-3 (engine) and 1e-2 (fused step) on these tensors, the HIP
path at 1.2e-3..3.4e-3 and 2e-3 -- the bounds are ~1.5x the measured HIP values (kernels and seeds
are deterministic), ten times tighter than round 1's 5e-2 and enough to expose a 1 % error in
any one layer.  The 1e-5 gradient checks are the per-kernel tests (tests/test_gpu_kernels.py) and
the 1e-3 ones the golden small nets.  Biases feeding an InstanceNorm are excluded (mathematically dead, gradient
= rounding noise in the reference too).
"""
import os
import types

import numpy as np
import pytest
import torch

import nirgan_oracle as O
from nirgan_hip.options import OPT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def sub(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in z.items() if k.startswith(prefix)}


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what + ": non-finite"
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= tol * max(ref, 1e-20), f"{what}: err {err:.3e} ref {ref:.3e}"


def grad_close(a, b, what, l2=1e-3, mx=1e-2):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape and torch.isfinite(a).all(), what
    nrm = b.norm().item()
    e2 = (a - b).norm().item()
    em = (a - b).abs().max().item()
    assert e2 <= l2 * max(nrm, 1e-20), f"{what}: rel L2 {e2 / max(nrm, 1e-20):.3e}"
    assert em <= mx * max(b.abs().max().item(), 1e-20), f"{what}: max err {em:.3e} of {b.abs().max().item():.3e}"


def grad_close64(a, b64, what, l2, mx=5e-2):
    """Against an fp64 reference: relative L2 and max error."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b64).detach().double().cpu()
    assert a.shape == b.shape and torch.isfinite(a).all(), what
    e2, em = ((a - b).norm() / b.norm().clamp_min(1e-30)).item(), ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert e2 <= l2, f"{what}: rel L2 vs fp64 {e2:.3e} > {l2:.1e}"
    assert em <= mx, f"{what}: max err vs fp64 {em:.3e}"


def leaf64(sd):
    return {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}


def make_nets(z, n_blocks, ngf=8):
    from model import networks
    netG = networks.define_G(3, 1, ngf, f"resnet_{n_blocks}blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, ngf, "basic", 3, "instance", "normal", 0.02)
    netG.load_state_dict(sub(z, "G0/"))
    netD.load_state_dict(sub(z, "D0/"))
    return netG.to(DEV), netD.to(DEV)


RS_W = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333, "lambda_savi": 0.0, "lambda_msavi": 0.0,
        "lambda_gndvi": 0.0}


@pytest.mark.parametrize("name", ["f1_g6_d.npz", "f1_g9_rs_pad.npz"])
def test_golden_small_nets_fused_step(golden_dir, name):
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, name)
    nb, pad, lam_rs = int(z["n_blocks"]), int(z["padding"]), float(z["lambda_rs"])
    netG, netD = make_nets(z, nb)
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, lambda_rs=lam_rs, rs_weights=RS_W, padding=pad)
    rgb, nir = torch.from_numpy(z["rgb"]).to(DEV), torch.from_numpy(z["nir"]).to(DEV)
    out = tr.step(rgb, nir).as_dict()
    close(tr.G.pred, z["pred"], 1e-3, "pred")
    for k in ("loss_D", "loss_G", "loss_G_gan", "loss_G_l1"):
        close(out[k], z[k], 1e-3, k)
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    for k, v in sub(z, "gD/").items():
        if k not in O.shadowed_bias_keys("D"):
            grad_close(gD[k], v, "gD " + k)
    shadow = O.shadowed_bias_keys("G", nb)
    for k, v in sub(z, "gG/").items():
        if k not in shadow:
            grad_close(gG[k], v, "gG " + k)
    # Adam: exact for our own gradient
    pD = dict(netD.named_parameters())
    for k in ("model.0.weight", "model.8.weight", "model.11.bias"):
        p0 = torch.from_numpy(z["D0/" + k].copy())
        m, v = torch.zeros_like(p0), torch.zeros_like(p0)
        O.adam_step(p0, gD[k].cpu(), m, v, 1, lr=2e-4, b1=0.5)
        close(pD[k], p0, 1e-6, "adam " + k)


@pytest.mark.parametrize("variant", ["F(6x6,3x3)", "F(4x4,3x3)", "direct"])
def test_medium_width_nets_take_the_winograd_paths(monkeypatch, variant):
    """ngf = ndf = 32 on 64x64 tiles: every Winograd variant at small tile counts in one fused step against the oracle -- residual blocks
    (128 channels at 16x16: the instance-norm apply folded into the second convolution's input transform, fused dY transforms,
    transform-domain weight gradient) and the PatchGAN's 4x4 layer (128 -> 256 channels at 8x8 -> 7x7: odd extent): F(6x6,3x3) / F(4x4,4x4) by
    default, F(4x4,3x3) with OPT.winograd = "f4", the direct tiles everywhere with OPT.winograd = "off"."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    from nirgan_hip.options import OPT
    monkeypatch.setattr(OPT, "winograd", {"F(6x6,3x3)": "f6", "F(4x4,3x3)": "f4", "direct": "off"}[variant])
    torch.manual_seed(7)
    netG = networks.define_G(3, 1, 32, "resnet_6blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, 32, "basic", 3, "instance", "normal", 0.02)
    G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()}
    g = torch.Generator().manual_seed(8)
    rgb, nir = torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 1, 64, 64, generator=g)
    tr = Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=6, lr=0.0)
    out = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
    names = [n for pl in (tr.G.fwd, tr.G.bwd, tr.D2.fwd, tr.D2.bwd, tr.D1.bwd_pred) for n, _ in pl.ops]
    if variant != "direct":      # residual blocks as F(6x6,3x3) / F(4x4,3x3) AND the PatchGAN's 4x4 layer as F(4x4,4x4)
        rcodes = {a[0]._obj.r for pl in (tr.G.fwd, tr.G.bwd) for n, a in pl.ops if n == "nirgan_wino6_gemm"}
        assert rcodes == ({6} if variant == "F(6x6,3x3)" else {3}), rcodes
        # (with the three-term split tiles -- the default -- a layer's data-gradient GEMMs and its transform-domain weight gradient are two
        # launches, nirgan_wino6_gemm + nirgan_wgrad_igemm with planes; the exact-fp32 tiles share one grid: nirgan_wino6_gemm_wgrad_pair)
        for want in ("nirgan_wino6_input_norm", "nirgan_wino6_input_dy_norm" if variant == "F(6x6,3x3)" else "nirgan_wino6_input_dy", "nirgan_wino6_gemm",
                     "nirgan_wgrad_igemm" if (OPT.split3 and OPT.split3_wino) else "nirgan_wino6_gemm_wgrad_pair", "nirgan_wino6_output",
                     "nirgan_wino6_wgrad_finish_r"):
            assert want in names, want
        assert sum(1 for pl in (tr.D2.fwd, tr.D2.bwd, tr.D1.fwd, tr.D1.bwd_pred) for n, a in pl.ops if n.startswith("nirgan_wino6_gemm")) == 4
    else:
        assert "nirgan_conv_wgrad_pair" in names and not any(n.startswith("nirgan_wino6") for n in names)
    ref = O.OracleTrainer(G0, D0, 6, lr=0.0)
    o = ref.step(rgb, nir)
    close(tr.G.pred, ref.last["pred"], 1e-3, "pred")
    for k in ("loss_D", "loss_G", "loss_G_l1"):
        close(out[k], o[k], 1e-3, k)
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    # unforced fp32-vs-fp32 comparison on a 2 x 64 x 64 batch: a handful of activations within rounding of zero flip; F(6x6,3x3) carries
    # 5x the rounding noise of F(4x4,3x3) (1.7e-5 against 3.4e-6 of a layer's output range), so more of them do.  Which ones is a lottery
    # of the last ulp: over three seeds and three arithmetic settings (exact fp32 tiles; the three-term split tiles for the direct layers;
    # for the Winograd layers too) the worst tensor lands anywhere between 1.5e-5 and 8.2e-3 with the prediction at 2-3e-5 every time,
    # the exact-fp32 setting itself at 5.1e-3 / 6.3e-3 on two of the seeds (scripts/diag_x3_medium.py, profiles/r05_medium_width_kink_noise.txt);
    # the bound that does not depend on the flips is the forced-kink one of test_fullsize_fused_step_against_oracle (3e-4, unchanged)
    # The same holds for F(4x4,3x3) (profiles/r05_medium_width_kink_noise_f4.txt: seed 17 in exact fp32 3.0e-3, seed 7 with the split tiles
    # 5.4e-3, seed 27 6e-6 in every setting), so the three variants share the bound; "direct" has no Winograd noise and keeps 1e-3.
    l2 = 1e-3 if variant == "direct" else 1e-2
    mx = 1e-2 if variant == "direct" else 1e-1          # (one flipped activation is a local error of a few per cent of the tensor's maximum: 4.5e-2 seen)
    for k, v in ref.last["grads_D"].items():
        if k not in O.shadowed_bias_keys("D"):
            grad_close(gD[k], v, "gD " + k, l2=l2, mx=mx)
    for k, v in ref.last["grads_G"].items():
        if v is not None and k not in O.shadowed_bias_keys("G", 6):
            grad_close(gG[k], v, "gG " + k, l2=l2, mx=mx)


@pytest.mark.parametrize("variant", ["F(6x6,3x3)", "F(4x4,3x3)", "direct"])
def test_medium_width_nets_with_the_kinks_forced(monkeypatch, variant):
    """The same medium-width step (ngf = ndf = 32, 2 x 64 x 64: the Winograd variants at small tile counts) held as tightly as the
    full-size one: against the oracle in FP64 taking the device's own branch decisions (oracle.forced_kinks), every gradient tensor at
    3e-4 in relative L2 -- the bound that does not depend on which activation sits within rounding of zero (the unforced comparison
    above is a lottery of the last ulp and only bounds the damage at 1e-2)."""
    from nirgan_hip.options import OPT
    monkeypatch.setattr(OPT, "winograd", {"F(6x6,3x3)": "f6", "F(4x4,3x3)": "f4", "direct": "off"}[variant])
    _fused_step_against_oracle(2, 64, 64, 6, 8, ngf=32)


@pytest.mark.parametrize("micro", [1, 2])
def test_fused_step_with_the_ssim_term(golden_dir, micro):
    """lambda_ssim > 0 (model/pix2pix.py:233-237, utils/losses.py:10-30): the fused HIP step with the SSIM term against the oracle's
    trainer carrying the same term (lr 0: both generator passes see the same discriminator)."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g9_rs_pad.npz")
    nb, pad = int(z["n_blocks"]), int(z["padding"])
    netG, netD = make_nets(z, nb)
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, padding=pad, lr=0.0, lambda_ssim=40.0, micro_batches=micro)
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    out = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
    ref = O.OracleTrainer(sub(z, "G0/"), sub(z, "D0/"), nb, padding=pad, lr=0.0, lambda_ssim=40.0)
    o = ref.step(rgb, nir)
    for k in ("loss_G", "loss_G_ssim", "loss_G_l1", "loss_D"):
        close(out[k], o[k], 1e-3, k)
    gG = tr.flatG.grad_views()
    shadow = O.shadowed_bias_keys("G", nb)
    for k, v in ref.last["grads_G"].items():
        if v is not None and k not in shadow:
            grad_close(gG[k], v, "gG " + k)


@pytest.mark.parametrize("name,pc", [("f1_inject.npz", False), ("f1_inject_pc.npz", True)])
def test_golden_inject_generator(golden_dir, capsys, name, pc):
    """(f1_inject_pc: the reference's post_correction=True branch, model/generator_inject.py:97-100,133-134 -- the prediction times a
    learnable scalar initialised to 0.8: nirgan_param_scale_fwd / _bwd)"""
    from model import networks
    from model.generator_inject import define_G_inject
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, name)
    ns = types.SimpleNamespace
    cfg = ns(base_configs=ns(input_nc=3, output_nc=1, ngf=8, netG="resnet_9blocks", norm="instance", no_dropout=True,
                             init_type="normal", init_gain=0.02),
             satclip=ns(satclip_inject_style="multiply", post_correction=pc, post_correction_init=0.8,
                        scaling_param=True, scaling_param_init=0.01))
    netG = define_G_inject(cfg)
    sd = sub(z, "G0/")
    g = torch.Generator().manual_seed(int(z["fc_seed"]))
    sd["fc.weight"] = torch.randn(16384, 256, generator=g) * 0.02
    sd["fc.bias"] = torch.randn(16384, generator=g) * 0.02
    netG.load_state_dict(sd)
    netD = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    netD.load_state_dict(sub(z, "D0/"))
    netG, netD = netG.to(DEV), netD.to(DEV)
    rgb, nir, emb = (torch.from_numpy(z[k]).to(DEV) for k in ("rgb", "nir", "embeds"))
    tr = Pix2PixTrainer(netG, netD, n_blocks=9, inject={"style": "multiply", "use_scale": True, "post_correction": pc})
    out = tr.step(rgb, nir, emb).as_dict()
    close(tr.G.pred, z["pred"], 1e-3, "pred")
    close(out["loss_G"], z["loss_G"], 1e-3, "loss_G")
    gr = tr.flatG.grad_views()
    grad_close(gr["scale_param"].reshape(1), torch.from_numpy(z["g_scale_param"]).reshape(1), "dscale")
    if pc:
        grad_close(gr["post_correction_param"].reshape(1), torch.from_numpy(z["gG/post_correction_param"]).reshape(1), "dpost_correction_param")
        # ... and one Adam step of it (the reference's torch.optim.Adam result)
        close(tr.flatG.param_views()["post_correction_param"].reshape(1), torch.from_numpy(z["G1/post_correction_param"]).reshape(1), 1e-5, "post_correction_param after Adam")
    grad_close(gr["fc.bias"], z["g_fc_bias"], "dfc.bias")
    grad_close(gr["fc.weight"][:8], z["g_fc_weight_rows0_8"], "dfc.weight")
    for k, v in sub(z, "gG/").items():
        if k not in O.shadowed_bias_keys("G", 9):
            grad_close(gr[k], v, "gG " + k)


def synth(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return 0.02 + 0.58 * torch.rand(B, 3, H, W, generator=g), 0.05 + 0.75 * torch.rand(B, 1, H, W, generator=g)


@pytest.mark.parametrize("nb,pad,size", [(6, 0, 256), (9, 10, 256)])
def test_fullsize_generator_engine_against_oracle(golden_dir, nb, pad, size):
    """ngf = 64: forward and backward of the generator for a GIVEN smooth output gradient, through the autograd
    bridge, against the CPU oracle's autograd (and the reference's own output samples)."""
    from model import networks
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, f"resnet_{nb}blocks", "instance", False, "normal", 0.02)
    sd = {k: v.clone() for k, v in netG.state_dict().items()}
    rgb, _ = synth(1, size, size, 1234)
    dout = torch.randn(1, 1, size, size, generator=torch.Generator().manual_seed(2))
    netG = netG.to(DEV)
    netG.data_pad = pad
    pred = netG(rgb.to(DEV))
    pred.backward(dout.to(DEV))
    if pad == 0 and size == 256:   # the reference's own output on this tile (committed by oracle/make_golden.py)
        z5 = load(golden_dir, "f5_fullsize.npz")
        close(pred.detach().cpu().flatten()[torch.from_numpy(z5[f"g{nb}_idx"])], z5[f"g{nb}_samples"], 1e-3, "pred vs reference samples")
    # the oracle takes the ReLU branches the device took (see test_fullsize_fused_step_against_oracle): both evaluate the same smooth
    # function and every gradient tensor is held to 3e-4; unforced, the activations within fp32 rounding of zero that flip move the
    # tensors by 4e-3 .. 7e-3 and single elements by 7e-2 (the kink noise of DESIGN 4, not kernel error)
    kinks = generator_kinks(netG._pool().free[(1, size, size, pad, True)][-1])
    p64 = leaf64(sd)
    with O.forced_kinks(kinks):
        ref = O.px_forward(p64, rgb.double(), nb, pad)
        ref.backward(dout.double())
    close(pred, ref, 1e-3, "pred")
    shadow = O.shadowed_bias_keys("G", nb)
    worst = 0.0
    for k, p in netG.named_parameters():
        if k not in shadow:
            worst = max(worst, ((p.grad.double().cpu() - p64[k].grad).norm() / p64[k].grad.norm()).item())
            grad_close64(p.grad, p64[k].grad, "gG " + k, l2=3e-4, mx=3e-3)
    print(f"generator {nb} blocks pad {pad}, kinks forced: worst rel-L2 over the gradient tensors {worst:.2e}")


def test_fullsize_discriminator_engine_against_oracle():
    """ndf = 64 PatchGAN on 2 x 4 x 256 x 256: output, parameter gradients and input gradient for a given dout."""
    from model import networks
    torch.manual_seed(0)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    pD = leaf64(netD.state_dict())
    rgb, nir = synth(2, 256, 256, 99)
    x = torch.cat((rgb, nir), 1)
    dout = torch.randn(2, 1, 30, 30, generator=torch.Generator().manual_seed(4))
    netD = netD.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = netD(xg)
    y.backward(dout.to(DEV))
    xr = x.double().requires_grad_(True)
    yr = O.discriminator_forward(pD, xr)
    yr.backward(dout.double())
    close(y, yr, 1e-3, "D out")
    for k, p in netD.named_parameters():
        if k not in O.shadowed_bias_keys("D"):
            grad_close64(p.grad, pD[k].grad, "gD " + k, l2=2e-3)
    grad_close64(xg.grad, xr.grad, "dD/dx", l2=2e-3)


def nchw(h):
    return (h.interior() > 0).permute(0, 3, 1, 2).cpu()


def generator_kinks(G):
    """ReLU masks of the generator engine's last forward, in the oracle's call order."""
    g = [nchw(G.L1.out), nchw(G.L2.out), nchw(G.L3.out)]
    for _, c1, _c2 in G.blocks:
        if getattr(c1, "defer_apply", False):        # the activated tensor is never written: z = (y - mean) * rstd > 0  <=>  y > mean
            g.append((c1.y.t > c1.stats[0][:, None, None, :]).permute(0, 3, 1, 2).cpu())
        else:
            g.append(nchw(c1.out))
    return g + [nchw(G.U1.out), nchw(G.U2.out)]


def device_kinks(tr, nir, rgb=None):
    """Branch decisions (ReLU / LeakyReLU masks, sign of pred - nir) of the fused step the trainer just ran, in the call order of
    oracle.OracleTrainer.step: G forward, D(fake), D(real), G forward again, D(fake) against the updated D, L1."""
    G, D2, D1 = tr.G, tr.D2, tr.D1
    B = G.B
    g = generator_kinks(G)
    d2 = [nchw(c.out) for c in (D2.C1, D2.C2, D2.C3, D2.C4)]
    d1 = [nchw(c.out) for c in (D1.C1, D1.C2, D1.C3, D1.C4)]
    masks = g + [m[:B] for m in d2] + [m[B:] for m in d2] + g + d1 + [(tr.G.pred.cpu() - nir) > 0]
    if tr.lambda_rs > 0.0 and rgb is not None:       # criterion l1 on the weighted spectral indices: sign of idx(nir) - idx(pred), dict order
        idx = O.rs_index_pairs(rgb, nir, tr.G.pred.cpu(), "loss")
        masks += [(idx[k][0] - idx[k][1]) > 0 for k in O.RS_ORDER if tr.rs_weights.get("lambda_" + k, 0.0) > 0.0]
    return masks


def test_fullsize_fused_step_against_oracle():
    """The whole two-optimizer step at reference size (ngf = ndf = 64, 256 x 256) against the oracle in FP64 with the kinks
    teacher-forced: the oracle takes every ReLU / LeakyReLU / sign(pred - nir) branch the DEVICE took (oracle.forced_kinks), so that
    both evaluate the same smooth function and every gradient tensor is held to 3e-4 in relative L2 (measured: worst 5.9e-5).
    Without the forcing a few hundred of the 26 M activations sit within fp32 rounding of zero and flip between two correct
    evaluations, each moving a gradient tensor by ~1e-3 of its norm: the unforced fp64 comparison sits at 2e-3..2e-2 for the HIP
    path and at 1e-2 for the reference's own fp32 CPU arithmetic (profiles/r02_grad_error_kinkfree_vs_fp64.txt); that the device's
    branch decisions are legitimate is checked separately: they differ from the fp64 evaluation's own in < 1e-4 of the elements, all
    of them within 1e-4 of the kink."""
    _fused_step_against_oracle(1, 256, 256, 6, 1234)


RS_W3 = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333, "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}


def _fused_step_against_oracle(B, H, W, nb, seed, lambda_rs=0.0, out_bias=None, padding=0, inject=False, ngf=64):
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    torch.manual_seed(0)
    if inject:                        # configs/config_px2px_SatCLIP.yaml: define_G_inject (model/generator_inject.py:105-135), multiply style
        from model.generator_inject import define_G_inject
        ns = types.SimpleNamespace
        netG = define_G_inject(ns(base_configs=ns(input_nc=3, output_nc=1, ngf=ngf, netG=f"resnet_{nb}blocks", norm="instance", no_dropout=True,
                                                  init_type="normal", init_gain=0.02),
                                  satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0,
                                             scaling_param=True, scaling_param_init=0.5)))
    else:
        netG = networks.define_G(3, 1, ngf, f"resnet_{nb}blocks", "instance", False, "normal", 0.02)
    torch.manual_seed(0)
    netD = networks.define_D(4, ngf, "basic", 3, "instance", "normal", 0.02)
    if out_bias is not None:          # as oracle/make_golden.py::f1: pred in (0.5, 1) keeps the index denominators pred + band away from 0
        with torch.no_grad():
            list(netG.parameters())[-1].fill_(out_bias)
    pG, pD = {k: v.clone() for k, v in netG.state_dict().items()}, {k: v.clone() for k, v in netD.state_dict().items()}
    rgb, nir = synth(B, H, W, seed)
    kw = dict(lambda_rs=lambda_rs, rs_weights=RS_W3) if lambda_rs > 0.0 else {}
    okw = dict(kw)
    emb = None
    if padding:
        kw["padding"], okw["padding"] = padding, padding
    if inject:
        emb = torch.randn(B, 256, generator=torch.Generator().manual_seed(seed + 1))
        kw["inject"] = {"style": "multiply", "use_scale": True}
        okw["inject_cfg"] = {"style": "multiply", "use_scale": True}
    tr = Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=nb, **kw)
    out = tr.step(rgb.to(DEV), nir.to(DEV), None if emb is None else emb.to(DEV)).as_dict()
    kinks = device_kinks(tr, nir, rgb)
    p64G, p64D = {k: v.double() for k, v in pG.items()}, {k: v.double() for k, v in pD.items()}
    e64 = None if emb is None else emb.double()
    kw = okw
    # (1) the device's decisions against the fp64 evaluation's own: only elements at the kink may differ
    with O.record_kinks() as own:
        free = O.OracleTrainer(p64G, p64D, nb, **kw)
        free.step(rgb.double(), nir.double(), e64)
    assert len(own) == len(kinks)
    flips = sum(int((a != b).sum()) for a, b in zip(own, kinks))
    total = sum(a.numel() for a in own)
    assert flips <= 1e-4 * total, (flips, total)
    # (2) same branches, smooth comparison
    with O.forced_kinks(kinks):
        ref = O.OracleTrainer(p64G, p64D, nb, **kw)
        o = ref.step(rgb.double(), nir.double(), e64)
    close(tr.G.pred, ref.last["pred"], 1e-3, "pred")
    for k in ("loss_D", "loss_G", "loss_G_gan", "loss_G_l1") + (("loss_G_rs",) if lambda_rs > 0.0 else ()):
        close(out[k], o[k], 1e-3, k)
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    worst = 0.0
    for name, gdev, gref, shadow in (("gD", gD, ref.last["grads_D"], O.shadowed_bias_keys("D")), ("gG", gG, ref.last["grads_G"], O.shadowed_bias_keys("G", nb))):
        for k, v in gref.items():
            if k not in shadow:
                if v.numel() == 1:          # a scalar gradient (the PatchGAN's last bias, scale_param) is one sum with heavy cancellation: 2e-3 of it
                    grad_close64(gdev[k], v, f"{name} {k}", l2=2e-3, mx=2e-3)
                    continue
                worst = max(worst, ((gdev[k].double().cpu() - v).norm() / v.norm()).item())
                grad_close64(gdev[k], v, f"{name} {k}", l2=3e-4, mx=3e-3)
    if inject:
        assert {"fc.weight", "fc.bias", "scale_param"} <= set(ref.last["grads_G"]), "the oracle did not differentiate the injection"
    print(f"fused step {B}x{H}x{W}" + (f" +pad {padding}" if padding else "") + (" inject" if inject else "") + f", kinks forced: worst rel-L2 over all gradient tensors {worst:.2e}; {flips} of {total} branch decisions differ from fp64's own")


def test_configs2_fullsize_fused_step_with_the_spectral_loss_against_oracle():
    """BASELINE.json configs[2] at full width: 9-block generator (configs/config_px2px.yaml:13), ngf 64, lambda_rs_losses 1 with the
    YAML's NDVI / NDWI / EVI weights (:31-39; utils/remote_sensing_indices.py:23-71), the output bias raised as in fixture
    f1_g9_rs_pad (3.0 here: at ngf 64 the last layer's pre-activation has a standard deviation of ~0.7, and one pixel with pred + band ~ 0 dominates the mean of a singular index), kinks teacher-forced (the |idx(nir) - idx(pred)| of the l1 criterion included): every gradient tensor of the
    fused step within 3e-4 of fp64.  The bs-32 shape of the config runs in test_configs2_batch32_properties."""
    _fused_step_against_oracle(1, 256, 256, 9, 77, lambda_rs=1.0, out_bias=3.0)


@pytest.mark.parametrize("shape", [(2, 256), (1, 512)])
def test_configs3_fullsize_fused_inject_step_against_oracle(shape):
    """BASELINE.json configs[3] as the FUSED two-optimizer step at full width (round 3 had it only through the autograd bridge at B = 1):
    configs/config_px2px_SatCLIP.yaml -- define_G_inject, 9 blocks, ngf 64, multiply style with the learned scale, reflect pad 10 --
    Pix2PixTrainer(inject=..., padding=10) on two 256 x 256 tiles and on one 512 x 512 tile (the config's tile size; its per-GPU batch of
    8 runs in bench.py --inject) against the fp64 oracle with the kinks teacher-forced: prediction, the four losses, every gradient
    tensor incl. fc.weight / fc.bias within 3e-4 (2e-3 for the scalars scale_param and the PatchGAN's last bias).
    (model/generator_inject.py:105-135, model/pix2pix.py:91-108,165-257.)"""
    B, size = shape
    _fused_step_against_oracle(B, size, size, 9, 41 + size, padding=10, inject=True)


def test_configs2_batch32_properties():
    """configs[2] as BASELINE.json states it -- bs 32, 256 x 256, 9 blocks, spectral loss: finite losses, the spectral term present,
    per-sample independence of the prediction, two identical runs bitwise equal."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    rgb, nir = synth(32, 256, 256, 5)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        netG = networks.define_G(3, 1, 64, "resnet_9blocks", "instance", False, "normal", 0.02).to(DEV)
        netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
        with torch.no_grad():
            list(netG.parameters())[-1].fill_(1.5)
        p4 = None
        if not runs:                                     # the same tiles as a batch of 4, before anything is updated
            netG.eval()
            with torch.no_grad():
                p4 = netG(rgb[8:12].contiguous())
            netG.train()
        tr = Pix2PixTrainer(netG, netD, n_blocks=9, lambda_rs=1.0, rs_weights=RS_W3)
        out = tr.step(rgb, nir).as_dict()
        assert all(np.isfinite(v) for v in out.values()) and out["loss_G_rs"] > 0.0, out
        if p4 is not None:
            close(tr.pred[8:12], p4, 3e-4, "sample independence at bs 32")
        runs.append((out, tr.pred.clone(), tr.flatG.grad.clone(), tr.flatG.flat.clone()))
        del tr
    (o0, p0, g0, w0), (o1, p1, g1, w1) = runs
    assert o0 == o1 and torch.equal(p0, p1) and torch.equal(g0, g1) and torch.equal(w0, w1), "two identical bs-32 steps differ"


def test_configs3_batch8_at_512_properties():
    """configs[3] at its per-GPU batch (configs/config_px2px_SatCLIP.yaml:100: batch 8; model/generator_inject.py:105-135): define_G_inject,
    9 blocks, ngf 64, multiply style with the learned scale, reflect pad 10, eight 512 x 512 tiles -- finite losses, the prediction of
    tiles 2..3 equal to the same tiles run as a batch of two (per-sample independence: InstanceNorm and the injection have no cross-sample
    term), and two identical steps bitwise equal.  The arithmetic itself is pinned at B = 1 / 2 by the forced-kink tests above."""
    from model import networks
    from model.generator_inject import define_G_inject
    from nirgan_hip.trainer import Pix2PixTrainer
    ns = types.SimpleNamespace
    rgb, nir = synth(8, 512, 512, 9)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    emb = torch.randn(8, 256, generator=torch.Generator().manual_seed(10)).to(DEV)
    inj = {"style": "multiply", "use_scale": True}

    def nets():
        torch.manual_seed(0)
        netG = define_G_inject(ns(base_configs=ns(input_nc=3, output_nc=1, ngf=64, netG="resnet_9blocks", norm="instance", no_dropout=True,
                                                  init_type="normal", init_gain=0.02),
                                  satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0,
                                             scaling_param=True, scaling_param_init=0.5))).to(DEV)
        torch.manual_seed(0)
        return netG, networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    netG, netD = nets()
    small = Pix2PixTrainer(netG, netD, n_blocks=9, inject=inj, padding=10, lr=0.0)
    small.step(rgb[2:4].contiguous(), nir[2:4].contiguous(), emb[2:4].contiguous())
    p2 = small.pred.clone()
    del small
    runs = []
    for _ in range(2):
        netG, netD = nets()
        tr = Pix2PixTrainer(netG, netD, n_blocks=9, inject=inj, padding=10)
        out = tr.step(rgb, nir, emb).as_dict()
        assert all(np.isfinite(v) for v in out.values()), out
        assert tr.pred.shape == (8, 1, 512, 512)
        if not runs:
            close(tr.pred[2:4], p2, 3e-4, "sample independence at bs 8, 512 x 512 + pad 10")
        runs.append((out, tr.pred.clone(), tr.flatG.grad.clone(), tr.flatD.grad.clone(), tr.flatG.flat.clone()))
        del tr
        torch.cuda.empty_cache()
    (o0, p0, g0, d0, w0), (o1, p1, g1, d1, w1) = runs
    assert o0 == o1 and torch.equal(p0, p1) and torch.equal(g0, g1) and torch.equal(d0, d1) and torch.equal(w0, w1), "two identical configs[3] steps differ"


def test_fused_step_is_bitwise_reproducible():
    """bs 16, the benchmark configuration: two runs from the same weights on the same tiles give bitwise-equal losses, gradients,
    Adam moments and parameters after two steps -- every reduction (weight-gradient slabs, instance-norm partials, loss sums, live
    bias gradients) is summed in a fixed order, nothing accumulates with float atomics across workgroups."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    rgb, nir = synth(16, 256, 256, 3)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
        netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6, lambda_rs=1.0, rs_weights=RS_W3)
        outs = [tr.step(rgb, nir).as_dict() for _ in range(2)]
        runs.append((outs, tr.flatD.grad.clone(), tr.flatG.grad.clone(), tr.flatD.flat.clone(), tr.flatG.flat.clone(),
                     tr.flatG.m.clone(), tr.flatG.v.clone()))
        del tr
    a, b = runs
    assert a[0] == b[0], (a[0], b[0])
    for x, y, what in zip(a[1:], b[1:], ("grad D", "grad G", "params D", "params G", "exp_avg G", "exp_avg_sq G")):
        assert torch.equal(x, y), f"{what}: {(x - y).abs().max().item():.3e} apart"


def test_bf16_storage_rules_at_the_benchmark_scale():
    """bs 16 at 256 x 256, ngf 64, bf16 operand mode: the step with the storage rules of DESIGN 3.3 (convolution outputs and data
    gradients stored as bf16, no fp32 store where every reader takes the twin) against the same step with every tensor kept in fp32
    (OPT.bf16_y / bf16_g / bf16_twin_only = False), and both against the exact-fp32 step.  The rules round tensors that are rounded
    again one kernel later (or add one rounding of 2^-9 in front of a normalisation).  Rounding is discontinuous: two bf16 evaluations
    that differ at all drift apart, layer by layer, up to the bf16 noise level itself (measured here: the rules move the prediction by
    1.08 x the distance of the rule-free bf16 step from the fp32 step) -- so the bound is that level, tensor by tensor: within 2 x of what
    the bf16 operand mode itself changes.  A wrong stride, a missed reader of a dropped fp32 tensor or a bf16 buffer read as fp32 shows
    as O(1) = 20-50 x the noise."""
    from model import networks
    from nirgan_hip.options import OPT
    from nirgan_hip.trainer import Pix2PixTrainer
    rgb, nir = synth(16, 256, 256, 3)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    runs = []
    for prec, rules in (("bf16", True), ("bf16", False), ("fp32", False)):
        OPT.bf16_y = OPT.bf16_g = OPT.bf16_twin_only = rules
        try:
            torch.manual_seed(0)
            netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
            netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
            tr = Pix2PixTrainer(netG, netD, n_blocks=6, lr=0.0, precision=prec)
            out = tr.step(rgb, nir).as_dict()
            ys = sum(1 for l in [tr.G.L1, tr.G.L2, tr.G.L3, tr.G.U1, tr.G.U2] + [c for _, a, b in tr.G.blocks for c in (a, b)] if l.y.is16)
            dead = sum(1 for h in tr.G.twinned + tr.D2.twinned if h.fp32_dead)
            g16 = sum(1 for n_, a_ in tr.G.bwd.ops if n_ == "nirgan_instnorm_bwd" and a_[0]._obj.g_bf16)
            runs.append((out, tr.G.pred.clone(), {k: v.clone() for k, v in tr.flatG.grad_views().items()},
                         {k: v.clone() for k, v in tr.flatD.grad_views().items()}, ys, dead, g16))
            del tr
        finally:
            OPT.reset()
    A, Bq, Cf = runs
    assert A[4] == 17 and A[5] >= 30 and A[6] >= 14 and Bq[4:] == (0, 0, 0) and Cf[4:] == (0, 0, 0), (A[4:], Bq[4:], Cf[4:])
    noise = (Bq[1] - Cf[1]).norm().item()
    assert noise > 0 and (A[1] - Bq[1]).norm().item() < 2 * noise, ((A[1] - Bq[1]).norm().item(), noise)
    assert (A[1] - Cf[1]).norm().item() < 2 * noise and noise < 0.05 * Cf[1].norm().item()
    for k in ("loss_D", "loss_G", "loss_G_l1"):
        close(A[0][k], Bq[0][k], 5e-3, k)
    worst = 0.0
    for i, shadow in ((2, O.shadowed_bias_keys("G", 6)), (3, O.shadowed_bias_keys("D"))):
        for k, v in Bq[i].items():
            if k not in shadow:
                worst = max(worst, ((A[i][k] - v).norm() / ((v - Cf[i][k]).norm() + 1e-20)).item())
    assert worst < 2.0, worst        # no gradient tensor moves by more than twice what the bf16 mode itself moves it


@pytest.mark.parametrize("shape", [(3, 72, 104), (2, 100, 60), (1, 36, 40)])
def test_ragged_sizes_fused_step_against_oracle(shape):
    """The same comparison on tiles that are neither square nor powers of two (multiples of 4, as the reference's down/up path needs):
    ragged Winograd tiles (trunk maps 18 x 26, 25 x 15, 9 x 10), partial M tiles everywhere, odd PatchGAN maps."""
    _fused_step_against_oracle(*shape, 6, 91)


def test_reference_full_discriminator_output(golden_dir):
    from model import networks
    z5 = load(golden_dir, "f5_fullsize.npz")
    torch.manual_seed(0)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    rgb, nir = synth(1, 256, 256, 1234)
    with torch.no_grad():
        y = netD(torch.cat((rgb, nir), 1).to(DEV))
    close(y, z5["d_out"], 1e-3, "D vs reference output")


def test_batch16_properties():
    """bs=16 (the benchmark configuration): sample independence, determinism, halo invariants."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    rgb, nir = synth(16, 256, 256, 7)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    netG.eval()
    with torch.no_grad():
        p16 = netG(rgb)
        p16b = netG(rgb)
        p4 = netG(rgb[4:8].contiguous())
    assert torch.equal(p16, p16b), "forward is not deterministic"
    # same tiles, other batch: only the reduction order of the instance-norm statistics differs (chunking depends on B), a ~1e-7
    # difference in mean / rstd that the 14 normalised layers and the F(6x6,3x3) transforms carry to 1e-4 of the output's range
    close(p16[4:8], p4, 3e-4, "sample independence")
    netG.train()
    tr = Pix2PixTrainer(netG, netD, n_blocks=6)
    for _ in range(3):
        out = tr.step(rgb, nir)
    d = out.as_dict()
    assert all(np.isfinite(v) for v in d.values()), d
    # zero halos of every buffer that relies on them are still zero after 3 steps
    for layer in (tr.G.L1, tr.G.L2, tr.G.U1, tr.D2.C1, tr.D2.C2, tr.D2.C3, tr.D2.C4):
        t, p = layer.out.t, layer.out.pad
        assert float(t[:, :p].abs().max()) == 0 and float(t[:, :, :p].abs().max()) == 0, layer.name
        zt, zp = layer.dy.t, layer.dy.pad
        if zp:
            assert float(zt[:, :zp].abs().max()) == 0 and float(zt[:, -zp:].abs().max()) == 0, layer.name + " dy"
    # gradient of the batch mean = mean of per-half gradients (what the RCCL all-reduce relies on)
    g_full = tr.flatG.grad.clone()
    assert torch.isfinite(g_full).all()


def test_data_parallel_equivalence_on_device():
    """What the RCCL all-reduce relies on: mean of the per-shard gradients == gradient of the whole batch
    (InstanceNorm is per sample, losses are means).  Two shards of 2 tiles vs one batch of 4, 6-block ngf 64, 128x128."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    rgb, nir = synth(4, 128, 128, 11)
    grads = []
    for shard in (slice(0, 4), slice(0, 2), slice(2, 4)):
        torch.manual_seed(0)
        netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
        netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6, lr=0.0)        # lr 0: D stays put, so the G gradients are comparable too
        tr.step(rgb[shard].to(DEV), nir[shard].to(DEV))
        grads.append((tr.flatD.grad.clone(), tr.flatG.grad.clone()))
    (gD, gG), (gD0, gG0), (gD1, gG1) = grads
    for full, a, b, name in ((gD, gD0, gD1, "D"), (gG, gG0, gG1, "G")):
        mean = 0.5 * (a + b)
        err = (mean - full).norm().item() / full.norm().item()
        assert err < 1e-4, (name, err)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_reduced_operand_modes_stay_close_to_fp32_at_full_width(precision):
    """ngf 64 (the width at which the generator's last layer runs as the direct fp32 kernels in every mode, csrc/endconv.hip, fed with
    fp32-packed weights): prediction and generator gradient of one step in the bf16 operand modes against the fp32 step -- bf16 operands
    cost ~2e-2, the split mode is fp32-like.  (Caught: bf16-stored packed weights handed to the fp32 kernels.)"""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    rgb, nir = synth(2, 128, 128, 5)

    def run(prec):
        torch.manual_seed(0)
        netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
        netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision=prec, lr=0.0)
        tr.step(rgb.to(DEV), nir.to(DEV))
        torch.cuda.synchronize()
        assert any(n == "nirgan_endconv_fwd" for n, _ in tr.G.fwd.ops)
        return tr.G.pred.clone(), tr.flatG.grad.clone()

    p0, g0 = run("fp32")
    p1, g1 = run(precision)
    ep = ((p1 - p0).norm() / p0.norm()).item()
    eg = ((g1 - g0).norm() / g0.norm()).item()
    bound = (5e-2, 8e-2) if precision == "bf16" else (1e-4, 5e-3)
    assert ep < bound[0] and eg < bound[1], (ep, eg)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_training_is_stable_over_many_steps(precision):
    """41 optimizer steps on a fixed batch: finite everywhere and the L1 term drops (in every operand precision: the bf16
    modes keep fp32 master weights, accumulation, normalisation and Adam)."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision=precision)
    rgb, nir = synth(2, 128, 128, 5)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    first = tr.step(rgb, nir).as_dict()
    for _ in range(40):
        last = tr.step(rgb, nir)
    last = last.as_dict()
    assert all(np.isfinite(v) for v in last.values()), last
    assert last["loss_G_l1"] < 0.7 * first["loss_G_l1"], (first, last)      # overfits a fixed batch
    assert all(torch.isfinite(p).all() for p in list(netG.parameters()) + list(netD.parameters()))


def test_inference_engines_and_tiling():
    """no_grad forward uses forward-only engines (create_synthetic_dataset.py:106-107) and equals the training-mode output."""
    from model import networks
    from nirgan_hip.inference import predict_tiled
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_9blocks", "instance", False, "normal", 0.02).to(DEV)
    rgb, _ = synth(2, 256, 256, 3)
    rgb = rgb.to(DEV)
    netG.data_pad = 10
    with torch.no_grad():
        p0 = netG(rgb)
    p1 = netG(rgb)
    assert p1.requires_grad and not p0.requires_grad
    assert torch.equal(p0, p1.detach())
    engines = [e for lst in netG._pool().free.values() for e in lst]
    assert any(len(e.bwd.ops) == 0 for e in engines)
    netG.data_pad = 0
    big = torch.cat([rgb, rgb.flip(-1)], -1)[:, :, :200, :300]
    out = predict_tiled(netG, big, tile=128, margin=16)
    assert out.shape == (2, 1, 200, 300) and torch.isfinite(out).all()


# ---------------------------------------------------------------------------------- operand precision modes
def test_bf16x3_split_mode_keeps_fp32_parity(golden_dir):
    """precision='bf16x3' on the device against the fp32 golden vectors of the reference: same 1e-3 bar as fp32."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision="bf16x3")
    out = tr.step(torch.from_numpy(z["rgb"]).to(DEV), torch.from_numpy(z["nir"]).to(DEV)).as_dict()
    close(tr.G.pred, z["pred"], 1e-3, "pred")
    for k in ("loss_D", "loss_G", "loss_G_gan", "loss_G_l1"):
        close(out[k], z[k], 1e-3, k)
    gD = tr.flatD.grad_views()
    for k, v in sub(z, "gD/").items():
        if k not in O.shadowed_bias_keys("D"):
            grad_close(gD[k], v, "gD " + k, l2=2e-3, mx=2e-2)


def test_bf16_mode_small_net_against_bf16_restatement(golden_dir):
    """precision='bf16' end to end on the device: inside the bf16 noise band around the oracle's bf16 restatement (the
    per-contraction rule is checked exactly in test_gpu_kernels / test_bf16_contraction_backward_rule)."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision="bf16")
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    out = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
    with O.operand_precision("bf16"):
        ref = O.OracleTrainer(sub(z, "G0/"), sub(z, "D0/"), 6)
        o = ref.step(rgb, nir)
    noise = (ref.last["pred"] - torch.from_numpy(z["pred"])).abs().max().item()
    assert noise > 1e-3
    assert (tr.G.pred.cpu().reshape(-1) - ref.last["pred"].reshape(-1)).abs().max().item() < 0.5 * noise
    close(out["loss_D"], o["loss_D"], 1e-2, "loss_D")
    close(out["loss_G"], o["loss_G"], 1e-2, "loss_G")


def test_precision_modes_fullsize_forward():
    """ngf = 64, 256x256: bf16x3 reproduces the fp32 engine to 5e-4 (measured 1.1e-4); bf16 sits within 5e-2 of it (and is not identical)."""
    from model import networks
    torch.manual_seed(0)
    net = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    rgb, _ = synth(2, 256, 256, 7)
    outs = {}
    for prec in ("fp32", "bf16x3", "bf16"):
        net.precision = prec
        with torch.no_grad():
            outs[prec] = net(rgb.to(DEV)).cpu()
    ref = outs["fp32"]
    assert (outs["bf16x3"] - ref).abs().max().item() <= 5e-4 * ref.abs().max().item()   # spec: 1e-3; measured 1.1e-4
    # bf16 operands: rounding noise of 17 layers at random init; bounded in L2 (the max over 131 k pixels is a tail statistic)
    d = ((outs["bf16"] - ref).norm() / ref.norm()).item()
    assert 1e-5 < d <= 3e-2, d


def test_mixed_resolution_buckets_on_device():
    """configs[4]: one trainer, three resolution buckets, engines kept per bucket; alternating buckets gives the same
    losses as dedicated trainers on the same sequence."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer

    def nets():
        torch.manual_seed(0)
        return (networks.define_G(3, 1, 16, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV),
                networks.define_D(4, 16, "basic", 3, "instance", "normal", 0.02).to(DEV))
    buckets = [(8, 64), (2, 128), (1, 256)]
    data = [tuple(t.to(DEV) for t in synth(b, s, s, 40 + i)) for i, (b, s) in enumerate(buckets)]
    g, d = nets()
    tr = Pix2PixTrainer(g, d, n_blocks=6)
    seq = [0, 1, 2, 0, 1, 2]
    got = [tr.step(*data[i]).as_dict() for i in seq]
    assert len(tr._states) == 3
    # oracle on the same sequence (small nets: a second or two)
    g, d = nets()
    ref = O.OracleTrainer({k: v.cpu() for k, v in g.state_dict().items()}, {k: v.cpu() for k, v in d.state_dict().items()}, 6)
    for n, i in enumerate(seq):
        o = ref.step(data[i][0].cpu(), data[i][1].cpu())
        tol = 1e-3 if n < 3 else 5e-3
        close(got[n]["loss_D"], o["loss_D"], tol, f"loss_D step {n}")
        close(got[n]["loss_G"], o["loss_G"], tol, f"loss_G step {n}")


def test_configs4_bf16_mixed_resolution_with_the_spectral_loss():
    """BASELINE.json configs[4] as written: mixed-resolution buckets x bf16 MFMA x full G + D + spectral-index loss (9 blocks,
    lambda_rs 1, NDVI/NDWI/EVI), one trainer.  Per bucket the fused step is compared with the oracle's bf16 restatement
    (operands of every contraction rounded to bf16, fp32 accumulate: oracle.operand_precision) from the same weights (lr 0
    keeps them in place): prediction inside the bf16 noise band, losses to 1e-2; the fp32 mode of the same trainer set to 1e-3
    of the fp32 oracle.  Then the bucket sequence with lr > 0 stays finite with one engine set per bucket.  The output bias is
    raised (as in fixture f1_g9_rs_pad) and the last conv's weights scaled down so that pred stays in (0.5, 1): the indices are singular where pred + band ~ 0."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    nb = 9

    def nets():
        torch.manual_seed(0)
        g = networks.define_G(3, 1, 16, "resnet_9blocks", "instance", False, "normal", 0.02)
        d = networks.define_D(4, 16, "basic", 3, "instance", "normal", 0.02)
        with torch.no_grad():
            list(g.parameters())[-1].fill_(1.5)
            list(g.parameters())[-2].mul_(0.2)       # last conv: pre-tanh = 1.5 +- a narrow spread at every resolution
        return g, d
    buckets = [(8, 64), (2, 128), (1, 256)]
    data = [synth(b, s_, s_, 60 + i) for i, (b, s_) in enumerate(buckets)]
    g, d = nets()
    sdG, sdD = {k: v.clone() for k, v in g.state_dict().items()}, {k: v.clone() for k, v in d.state_dict().items()}
    for prec in ("fp32", "bf16"):
        g, d = nets()
        tr = Pix2PixTrainer(g.to(DEV), d.to(DEV), n_blocks=nb, lr=0.0, lambda_rs=1.0, rs_weights=RS_W, precision=prec)
        for i, (rgb, nir) in enumerate(data):
            out = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
            ref32 = O.OracleTrainer(sdG, sdD, nb, lr=0.0, lambda_rs=1.0, rs_weights=RS_W)
            o32 = ref32.step(rgb, nir)
            if prec == "fp32":
                close(tr.pred, ref32.last["pred"], 1e-3, f"fp32 pred bucket {i}")
                for k in ("loss_D", "loss_G", "loss_G_rs"):
                    close(out[k], o32[k], 1e-3, f"fp32 {k} bucket {i}")
                continue
            with O.operand_precision("bf16", y_bf16_min_pixels=OPT.epilogue_min_pixels_bf16, y_bf16_min_tiles=OPT.bf16_store_min_tiles, g_bf16_min_tiles=OPT.bf16_store_min_tiles):      # the build's storage rule for convolution outputs
                ref = O.OracleTrainer(sdG, sdD, nb, lr=0.0, lambda_rs=1.0, rs_weights=RS_W)
                o = ref.step(rgb, nir)
            assert float(ref.last["pred"].min()) > 0.3          # denominators pred + band stay away from 0
            noise = (ref.last["pred"] - ref32.last["pred"]).abs().max().item()
            assert noise > 1e-4
            err = (tr.pred.cpu() - ref.last["pred"]).abs().max().item()
            assert err < 0.75 * noise, f"bucket {i}: {err:.3e} vs bf16 noise {noise:.3e}"
            for k in ("loss_D", "loss_G", "loss_G_rs", "loss_G_l1"):
                close(out[k], o[k], 1e-2, f"bf16 {k} bucket {i}")
        assert len(tr._states) == 3
    g, d = nets()
    tr = Pix2PixTrainer(g.to(DEV), d.to(DEV), n_blocks=nb, lambda_rs=1.0, rs_weights=RS_W, precision="bf16")
    for n in range(6):
        rgb, nir = data[n % 3]
        out = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
        assert all(np.isfinite(v) for v in out.values()), (n, out)
    assert len(tr._states) == 3 and tr.flatG.step_count == 6


def test_configs4_bf16_with_the_spectral_loss_at_full_width():
    """BASELINE.json configs[4]'s arithmetic at the reference's width (ngf = ndf = 64, 9 blocks, NDVI / NDWI / EVI loss, bf16 operands on
    the matrix pipe): all three resolution buckets (2 @128, 1 @256, 1 @512) against the oracle's bf16 restatement (every contraction's operands
    rounded to bf16 once, fp32 accumulate: oracle.operand_precision) from the same weights.  Rounding is discontinuous, so two correct
    bf16 evaluations drift apart up to the bf16 noise level: the device has to sit INSIDE the band the restatement spans against the fp32
    evaluation -- prediction within 0.75 of it, losses to 1e-2, every gradient tensor closer to the bf16 restatement than 1.5 x the
    distance between the bf16 and the fp32 restatements.  Parity unpinned against the reference (it has no bf16 path)."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    nb = 9

    def nets():
        torch.manual_seed(0)
        g = networks.define_G(3, 1, 64, "resnet_9blocks", "instance", False, "normal", 0.02)
        d = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
        with torch.no_grad():
            list(g.parameters())[-1].fill_(1.5)
            list(g.parameters())[-2].mul_(0.2)       # last conv: pre-tanh = 1.5 +- a narrow spread (pred + band stays away from 0)
        return g, d
    g, d = nets()
    sdG, sdD = {k: v.clone() for k, v in g.state_dict().items()}, {k: v.clone() for k, v in d.state_dict().items()}
    tr = Pix2PixTrainer(g.to(DEV), d.to(DEV), n_blocks=nb, lr=0.0, lambda_rs=1.0, rs_weights=RS_W, precision="bf16")
    shadowG, shadowD = O.shadowed_bias_keys("G", nb), O.shadowed_bias_keys("D")
    for i, (b, s_) in enumerate([(2, 128), (1, 256), (1, 512)]):       # (the 512 bucket: round 3's review, missing #3)
        rgb, nir = synth(b, s_, s_, 80 + i)
        out = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
        ref32 = O.OracleTrainer(sdG, sdD, nb, lr=0.0, lambda_rs=1.0, rs_weights=RS_W)
        ref32.step(rgb, nir)
        with O.operand_precision("bf16", y_bf16_min_pixels=OPT.epilogue_min_pixels_bf16, y_bf16_min_tiles=OPT.bf16_store_min_tiles, g_bf16_min_tiles=OPT.bf16_store_min_tiles):      # the build's storage rule for convolution outputs
            ref = O.OracleTrainer(sdG, sdD, nb, lr=0.0, lambda_rs=1.0, rs_weights=RS_W)
            o = ref.step(rgb, nir)
        assert float(ref.last["pred"].min()) > 0.3
        noise = (ref.last["pred"] - ref32.last["pred"]).abs().max().item()
        err = (tr.pred.cpu() - ref.last["pred"]).abs().max().item()
        assert noise > 1e-4 and err < 0.75 * noise, f"bucket {i}: pred {err:.3e} vs bf16 noise {noise:.3e}"
        for k in ("loss_D", "loss_G", "loss_G_rs", "loss_G_l1"):
            close(out[k], o[k], 1e-2, f"bf16 {k} bucket {i}")
        worst = 0.0
        for name, gdev, gref, g32, shadow in (("gD", tr.flatD.grad_views(), ref.last["grads_D"], ref32.last["grads_D"], shadowD),
                                              ("gG", tr.flatG.grad_views(), ref.last["grads_G"], ref32.last["grads_G"], shadowG)):
            for k, v in gref.items():
                if k in shadow or v is None or v.numel() == 1:
                    continue
                band = (v - g32[k]).norm().item()
                e = (gdev[k].cpu() - v).norm().item()
                worst = max(worst, e / max(band, 1e-30))
                assert e <= 1.5 * band + 1e-6 * v.norm().item(), f"bucket {i} {name} {k}: {e:.3e} from the bf16 restatement, band {band:.3e}"
        print(f"configs[4] full width, bucket {b}@{s_}: pred {err:.2e} of noise {noise:.2e}; worst gradient distance / band {worst:.2f}")


def _layer_restatement_check(layer, gw_dev, tag):
    """One ConvIN layer of an engine that has just run forward + backward in bf16 operand mode, teacher-forced on the DEVICE's own
    tensors: y against float64 convolution of the bf16-rounded input the device read and the bf16-rounded weights (+ bias), the weight
    gradient against the float64 correlation of the layer's own bf16 input and bf16 dY.  No drift to hide in: a wrong contraction in one
    layer fails here at 1e-4, whatever the end-to-end band is."""
    F = torch.nn.functional
    inp, k, s_, p = layer.inp, layer.k, layer.s, layer.p
    src = inp.t16 if inp.t16 is not None else inp.t                       # what the launch read (the twin, or fp32 rounded at the fragment read)
    x = src.to(torch.bfloat16).double()
    cin = getattr(layer, "cin", None) or inp.C
    w = layer.weight.detach().to(torch.bfloat16).double()
    bias = None if layer.bias is None else layer.bias.detach().double()
    if layer.kind == "convT":
        o = inp.pad
        xi = x[:, o:o + inp.H, o:o + inp.W, :].permute(0, 3, 1, 2)
        ref = F.conv_transpose2d(xi, w, bias, stride=2, padding=p, output_padding=1)
    else:
        o = inp.pad - p
        xi = x[:, o:o + inp.H + 2 * p, o:o + inp.W + 2 * p, :cin].permute(0, 3, 1, 2)
        ref = F.conv2d(xi, w, bias, stride=s_)
    ref = ref.permute(0, 2, 3, 1)
    y = layer.y.t.double()
    assert y.shape == ref.shape, (tag, y.shape, ref.shape)
    scale = ref.abs().max().item()
    # a bf16-stored y carries its own rounding (half an ulp = 2^-9 relative); fp32 accumulation of up to 2 304 products: 1e-4 of the scale
    bound = (2.0 ** -8) * ref.abs() + 1e-4 * scale if layer.y.is16 else 1e-4 * scale
    bad = ((y - ref).abs() > bound)
    assert not bool(bad.any()), f"{tag} forward: {int(bad.sum())} of {bad.numel()} outputs off, worst {(y - ref).abs().max().item():.3e} at scale {scale:.3e}"
    if gw_dev is None or getattr(layer, "dy", None) is None:
        return 0.0
    dy = layer.dy
    dsrc = dy.t16 if dy.t16 is not None else dy.t
    gy = dsrc.to(torch.bfloat16).double()[:, dy.pad:dy.pad + layer.OH, dy.pad:dy.pad + layer.OW, :].permute(0, 3, 1, 2)
    if layer.kind == "convT":
        gref = torch.nn.grad.conv_transpose2d_weight(xi, tuple(w.shape), gy, stride=2, padding=p, output_padding=1) if hasattr(torch.nn.grad, "conv_transpose2d_weight") else None
        if gref is None:
            xg = xi.clone().requires_grad_(False)
            wg = w.clone().requires_grad_(True)
            (F.conv_transpose2d(xg, wg, None, stride=2, padding=p, output_padding=1) * gy).sum().backward()
            gref = wg.grad
    else:
        gref = torch.nn.grad.conv2d_weight(xi, tuple(w.shape), gy, stride=s_)
    e = ((gw_dev.double() - gref).norm() / gref.norm().clamp_min(1e-30)).item()
    assert e <= 1e-4, f"{tag} weight gradient: rel L2 {e:.3e}"
    return e


def test_bf16_every_contraction_against_the_restatement_on_the_devices_own_inputs():
    """Round 3's review, weak #1: the end-to-end bf16 comparisons are band tests (a second bf16 evaluation drifts up to the bf16 noise
    level), which a 20 % error in one layer's path could hide in.  Here every convolution and transposed convolution of the full-width
    (ngf = ndf = 64, 9 blocks) bf16 step -- 23 generator layers, 4 PatchGAN layers -- is checked in isolation on the tensors the device
    itself read: forward output and weight gradient against float64 on the bf16-rounded operands, 1e-4 (plus the store rounding where y
    is kept as bf16).  BASELINE.json configs[4]'s arithmetic rule (operands rounded to bf16 once, fp32 accumulate), unpinned against the
    reference, which has no bf16 path."""
    from model import networks
    from nirgan_hip import lib as L
    from nirgan_hip.trainer import Pix2PixTrainer
    nb = 9
    torch.manual_seed(0)
    g = networks.define_G(3, 1, 64, "resnet_9blocks", "instance", False, "normal", 0.02)
    d = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    tr = Pix2PixTrainer(g.to(DEV), d.to(DEV), n_blocks=nb, lr=0.0, precision="bf16")
    rgb, nir = synth(12, 256, 256, 91)           # 12 tiles: the trunk launches run on the 256-wide tiles (192 of them), as the benchmark's do
    tr.step(rgb.to(DEV), nir.to(DEV))
    torch.cuda.synchronize()
    G, D2 = tr.G, tr.D2

    def grad_of(flat, layer):
        views = flat.grad_views()
        for name in flat.names:
            o, n_, shp = flat.slices[name]
            if flat.flat[o:o + n_].data_ptr() == layer.weight.data_ptr():
                return views[name]
        raise AssertionError("no parameter matches the layer's weight")
    worst, n = 0.0, 0
    layers = [("G.first", G.L1), ("G.down0", G.L2), ("G.down1", G.L3)] + [(f"G.block{j}.conv{c}", l) for j, (_, c1, c2) in enumerate(G.blocks) for c, l in ((1, c1), (2, c2))] + [("G.up0", G.U1), ("G.up1", G.U2)]
    for tag, layer in layers:
        worst = max(worst, _layer_restatement_check(layer, grad_of(tr.flatG, layer), tag))
        n += 1
    for tag, layer in (("D.c1", D2.C1), ("D.c2", D2.C2), ("D.c3", D2.C3), ("D.c4", D2.C4)):
        worst = max(worst, _layer_restatement_check(layer, grad_of(tr.flatD, layer), tag))
        n += 1
    names256 = [L.backend().nirgan_conv_kernel_name(a[0]).decode() for nme, a in G.fwd.ops if nme == "nirgan_conv_igemm"]
    assert "conv_igemm256_kernel" in names256, names256
    print(f"bf16 per-layer check: {n} layers, worst weight-gradient rel-L2 {worst:.2e}")


def test_micro_batches_on_two_streams_match_the_single_stream_step():
    """ngf = 64, bs 4 @128: the two-part step on two HIP streams gives the single-part step's losses and gradients
    (fp32 summation order of the weight gradients differs: 1e-4), over two consecutive steps (stream joins before each
    Adam step, re-packed weights after)."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer

    def run(micro):
        torch.manual_seed(0)
        g = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
        d = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
        tr = Pix2PixTrainer(g, d, n_blocks=6, micro_batches=micro)   # RS indices are singular at random init (tanh output): not here
        rgb, nir = synth(4, 128, 128, 21)
        outs = []
        for _ in range(2):
            o = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
            torch.cuda.synchronize()
            outs.append((o, tr.flatG.grad.clone(), tr.flatD.grad.clone(), tr.pred.clone()))
        return tr, outs
    tr1, a = run(1)
    tr2, b = run(2)
    assert tr2._state.n == 2 and tr2._state.streams[1] is not None
    for (o1, gG1, gD1, p1), (o2, gG2, gD2, p2) in zip(a[:1], b[:1]):
        dp = (p2 - p1).abs()
        if dp.max().item() > 1e-5 * p1.abs().max().item():
            bad = (dp > 0.25 * dp.max()).nonzero()
            print(f"pred differs: max {dp.max().item():.3e}, {int((dp > 0).sum())} px differ at all; worst quarter in tiles {sorted(set(bad[:, 0].tolist()))} "
                  f"rows {bad[:, 2].min().item()}..{bad[:, 2].max().item()} cols {bad[:, 3].min().item()}..{bad[:, 3].max().item()} ({len(bad)} px)")
        close(p2, p1, 1e-5, "pred")
        for k in o1:
            close(torch.tensor(o2[k]), torch.tensor(o1[k]), 1e-4, k)
        assert ((gD2 - gD1).norm() / gD1.norm()).item() < 1e-4
        assert ((gG2 - gG1).norm() / gG1.norm()).item() < 1e-3
    # second step: trajectories have separated by Adam's amplification of rounding noise only
    for k in a[1][0]:
        close(torch.tensor(b[1][0][k]), torch.tensor(a[1][0][k]), 5e-3, "step 2 " + k)


def test_two_stream_steps_are_bitwise_reproducible_under_contention():
    """The two-part step 500 times at lr = 0 (weights re-packed every step, every launch of one part overlapping the other part's on
    the second HIP stream): every step's prediction and both flat gradients bitwise those of the first step.  A fragment read of the
    split tile's K loop that no wait covered passed every single-stream test and failed 0.6 % of such steps (a stale low term in one
    16 x 16 MFMA tile, 1e-6 at that layer, 4e-5 in the prediction): scripts/diag_micro_stress.py names the tensors that differ."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    torch.manual_seed(0)
    g = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    d = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    tr = Pix2PixTrainer(g, d, n_blocks=6, lr=0.0, micro_batches=2)
    rgb, nir = synth(4, 128, 128, 21)
    rgb, nir = rgb.to(DEV), nir.to(DEV)
    tr.step(rgb, nir)
    torch.cuda.synchronize()
    assert tr._state.n == 2 and tr._state.streams[1] is not None
    ref = (tr.pred.clone(), tr.flatG.grad.clone(), tr.flatD.grad.clone())
    bad = []
    for i in range(1, 500):
        tr.step(rgb, nir)
        torch.cuda.synchronize()
        now = (tr.pred, tr.flatG.grad, tr.flatD.grad)
        if not all(torch.equal(a, b) for a, b in zip(now, ref)):
            bad.append((i, [(a - b).abs().max().item() for a, b in zip(now, ref)]))
    assert not bad, f"{len(bad)} of 499 steps differ from the first: {bad[:5]}"


def test_paired_sub_pixel_phases_on_the_device():
    """nirgan_conv_desc.out_span = 2 (ConvTranspose2d(128, 64, 3, s2) forward and the data gradient of Conv2d(64, 128, 3, s2) as two paired
    problems on the split tile, with the instance-norm partial sums and the fused first backward pass in the epilogue) against the four
    phases on the exact fp32 tile (OPT.pair_phases = False), ngf 64, bs 2 @128: prediction and losses to fp32 rounding, gradients to the
    bound the other exact-against-split comparisons use (unforced ReLU kinks)."""
    from model import networks
    from nirgan_hip.options import OPT
    from nirgan_hip.trainer import Pix2PixTrainer

    def run(pair):
        OPT.reset()
        OPT.pair_phases = pair
        try:
            torch.manual_seed(0)
            g = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
            d = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
            tr = Pix2PixTrainer(g, d, n_blocks=6, lr=0.0)          # (no Adam step between the passes: it amplifies rounding noise to its step size)
            rgb, nir = synth(2, 128, 128, 33)
            o = tr.step(rgb.to(DEV), nir.to(DEV)).as_dict()
            torch.cuda.synchronize()
            spans = [[a[0][i].contents.out_span for i in range(a[1])] for pl in (tr.G.fwd, tr.G.bwd) for n_, a in pl.ops if n_ == "nirgan_conv_igemm_group"]
            return o, tr.flatG.grad.clone(), tr.flatD.grad.clone(), tr.pred.clone(), spans
        finally:
            OPT.reset()
    o1, gG1, gD1, p1, spans1 = run(True)
    o0, gG0, gD0, p0, spans0 = run(False)
    assert [2, 2] in spans1 and sum(s == [2, 2] for s in spans1) == 2, spans1
    assert not any(2 in s for s in spans0), spans0
    close(p1, p0, 2e-5, "prediction")
    for k in o0:
        close(torch.tensor(o1[k]), torch.tensor(o0[k]), 1e-4, k)
    # (profiles/r05_medium_width_kink_noise.txt: two exact-fp32 tilings of these networks differ by 5e-3 in the gradients at random init)
    assert ((gD1 - gD0).norm() / gD0.norm()).item() < 5e-3
    assert ((gG1 - gG0).norm() / gG0.norm()).item() < 1e-2


def test_fullsize_inject_generator_forward_against_oracle():
    """configs[3] geometry at one tile: ngf 64, 9 blocks, 256x256 with the YAML's reflect pad 10 -> the SatCLIP map (128x128
    from the 256 -> 16384 fc) is resized to the 138x138 feature map and multiplies it (generator_inject.py:110-127); the
    forward against the CPU oracle at 1e-3, scale 0.5 so that the modulation is not a rounding-level effect."""
    from model.generator_inject import define_G_inject
    ns = types.SimpleNamespace
    cfg = ns(base_configs=ns(input_nc=3, output_nc=1, ngf=64, netG="resnet_9blocks", norm="instance", no_dropout=True,
                             init_type="normal", init_gain=0.02),
             satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0,
                        scaling_param=True, scaling_param_init=0.5))
    torch.manual_seed(0)
    net = define_G_inject(cfg)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    rgb, _ = synth(1, 256, 256, 31)
    emb = torch.randn(1, 256, generator=torch.Generator().manual_seed(32))
    net = net.to(DEV)
    net.data_pad = 10
    with torch.no_grad():
        pred = net(rgb.to(DEV), emb.to(DEV)).cpu()
    ref = O.px_forward(sd, rgb, 9, 10, emb, {"style": "multiply", "use_scale": True})
    assert pred.shape == ref.shape == (1, 1, 256, 256)
    close(pred, ref, 1e-3, "inject forward")
    plain = O.px_forward({**sd, "scale_param": torch.tensor(0.0)}, rgb, 9, 10, emb, {"style": "multiply", "use_scale": True})
    assert (ref - plain).abs().max().item() > 1e-2          # the injection changes the output visibly


def test_fullsize_inject_generator_backward_at_512_against_oracle():
    """configs[3] per-tile geometry, full width: ngf 64, 9 blocks, a 512x512 tile with the YAML's reflect pad 10 (532x532 inside the
    network, the 128x128 SatCLIP map resized to the 133x133 feature map), forward AND backward through the autograd bridge
    for a given output gradient against the fp64 oracle: every parameter gradient incl. the 256 -> 16384 fc and scale_param
    (generator_inject.py:105-135)."""
    from model.generator_inject import define_G_inject
    ns = types.SimpleNamespace
    cfg = ns(base_configs=ns(input_nc=3, output_nc=1, ngf=64, netG="resnet_9blocks", norm="instance", no_dropout=True,
                             init_type="normal", init_gain=0.02),
             satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0,
                        scaling_param=True, scaling_param_init=0.5))
    torch.manual_seed(0)
    net = define_G_inject(cfg)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    rgb, _ = synth(1, 512, 512, 33)
    emb = torch.randn(1, 256, generator=torch.Generator().manual_seed(34))
    dout = torch.randn(1, 1, 512, 512, generator=torch.Generator().manual_seed(35))
    net = net.to(DEV)
    net.data_pad = 10
    pred = net(rgb.to(DEV), emb.to(DEV))
    pred.backward(dout.to(DEV))
    # the oracle takes the ReLU branches the device took (see test_fullsize_fused_step_against_oracle): both sides evaluate the same
    # smooth function, so the bounds can be tight -- unforced, the ~1e-4 of activations that sit within fp32 rounding of zero move
    # every gradient tensor by 4e-3 and the scalar scale_param gradient (a sum with heavy cancellation) by 4e-2
    eng = net._pool().free[(1, 512, 512, 10, True)][-1]
    kinks = generator_kinks(eng)
    p64 = leaf64(sd)
    with O.forced_kinks(kinks):
        ref = O.px_forward(p64, rgb.double(), 9, 10, emb.double(), {"style": "multiply", "use_scale": True})
        ref.backward(dout.double())
    assert pred.shape == ref.shape == (1, 1, 512, 512)
    close(pred, ref, 1e-3, "inject 512 pred")
    shadow = O.shadowed_bias_keys("G", 9)
    seen, worst = set(), 0.0
    for k, p in net.named_parameters():
        if k in shadow or k not in p64 or p64[k].grad is None:
            continue
        worst = max(worst, ((p.grad.double().cpu() - p64[k].grad).norm() / p64[k].grad.norm()).item())
        grad_close64(p.grad, p64[k].grad, "inject gG " + k, l2=2e-3 if k == "scale_param" else 3e-4, mx=3e-3)
        seen.add(k)
    assert {"fc.weight", "fc.bias", "scale_param"} <= seen, sorted(seen)[:8]
    print(f"inject generator at 512 (+10), kinks forced: worst rel-L2 over the gradient tensors {worst:.2e}")


def test_size_512_and_128_forward_against_oracle():
    """The other tile sizes of configs[3]/[4] (512x512 and 128x128): generator and PatchGAN forward against the oracle.
    512: res blocks at 128x128 (64 M-tiles per sample), PatchGAN map 62x62; 128: res blocks at 32x32, PatchGAN map 14x14."""
    from model import networks
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    sdG = {k: v.clone() for k, v in netG.state_dict().items()}
    sdD = {k: v.clone() for k, v in netD.state_dict().items()}
    netG, netD = netG.to(DEV).eval(), netD.to(DEV).eval()
    for size, B in ((512, 1), (128, 3)):
        rgb, nir = synth(B, size, size, 50 + size)
        with torch.no_grad():
            pred = netG(rgb.to(DEV)).cpu()
            dout = netD(torch.cat((rgb, nir), 1).to(DEV)).cpu()
        ref = O.generator_forward(sdG, rgb, 6)
        close(pred, ref, 1e-3, f"G forward {size}")
        dref = O.discriminator_forward(sdD, torch.cat((rgb, nir), 1))
        assert dout.shape == dref.shape == (B, 1, size // 8 - 2, size // 8 - 2)
        close(dout, dref, 1e-3, f"D forward {size}")


@pytest.mark.parametrize("cfg", [(32, 2, 64), (64, 2, 256)])
def test_instance_norm_backward_first_pass_inside_the_output_transform(monkeypatch, cfg):
    """The data gradient's output transform in its fused mode (csrc/wino6.hip::wino6_output_inbwd_kernel: reflect fold in registers,
    skip gradient, dense folded gradient, partial sums of the consumer's first backward pass; OPT.fuse_inbwd = False turns it off) against the
    separate first pass: same forward (bitwise), gradients equal to fp32 rounding -- the backward has no branch that a
    rounding difference could flip, the masks come from the forward."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    ngf, B, size = cfg
    rgb, nir = synth(B, size, size, 77)

    def run(fused):
        from nirgan_hip.options import OPT
        monkeypatch.setattr(OPT, "fuse_inbwd", bool(fused))
        torch.manual_seed(3)
        netG = networks.define_G(3, 1, ngf, "resnet_6blocks", "instance", False, "normal", 0.02)
        netD = networks.define_D(4, ngf, "basic", 3, "instance", "normal", 0.02)
        tr = Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=6, lr=0.0)
        tr.step(rgb.to(DEV), nir.to(DEV))
        torch.cuda.synchronize()
        n_fused = sum(1 for n, a in tr.G.bwd.ops if n == "nirgan_wino6_output" and a[0]._obj.fuse_gz)
        n_pre = sum(1 for n, a in tr.G.bwd.ops if n == "nirgan_instnorm_bwd" and a[0]._obj.sums_chunks > 0 and a[0]._obj.gsum_out)
        return tr.G.pred.clone(), tr.flatG.grad.clone(), tr.flatD.grad.clone(), n_fused, n_pre

    p1, g1, d1, nf1, np1 = run(True)
    p0, g0, d0, nf0, np0 = run(False)
    assert nf1 == 12 and np1 == 12 and nf0 == 0 and np0 == 0, (nf1, np1, nf0, np0)
    assert torch.equal(p1, p0)
    assert ((d1 - d0).norm() / d0.norm()).item() < 1e-6        # the discriminator's own backward is untouched (fixed-order reductions: only the fusion's own rounding differs)
    rel = ((g1 - g0).norm() / g0.norm()).item()
    assert rel < 2e-5, rel


def test_instance_norm_backward_first_pass_inside_the_conv_epilogues(monkeypatch):
    """The up/down-sampling layers' data gradients (direct tiles) take the consumer layer's first backward pass in their epilogue
    (nirgan_conv_desc.fuse_*, from 16 K pixels per sample: the 128x128 maps of a 128x128 tile) -- against the separate pass
    (OPT.fuse_inbwd = False): same forward bitwise, gradients equal to fp32 rounding."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    rgb, nir = synth(2, 128, 128, 78)

    def run(fused):
        from nirgan_hip.options import OPT
        monkeypatch.setattr(OPT, "fuse_inbwd", bool(fused))
        torch.manual_seed(3)
        netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02)
        netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
        tr = Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=6, lr=0.0)
        tr.step(rgb.to(DEV), nir.to(DEV))
        torch.cuda.synchronize()
        n = 0
        for plan in (tr.G.bwd, tr.D2.bwd, tr.D1.bwd_pred):
            for name, a in plan.ops:
                if name == "nirgan_conv_igemm" and a[0]._obj.fuse_y:
                    n += 1
                if name == "nirgan_conv_igemm_group":       # (a paired problem -- out_span = 2 -- stands for two sub-pixel phases)
                    n += sum(max(1, a[0][i].contents.out_span) for i in range(a[1]) if a[0][i].contents.fuse_y)
        return tr.G.pred.clone(), tr.flatG.grad.clone(), tr.flatD.grad.clone(), n

    p1, g1, d1, n1 = run(True)
    p0, g0, d0, n0 = run(False)
    assert n1 >= 4 and n0 == 0, (n1, n0)
    assert torch.equal(p1, p0)
    assert ((d1 - d0).norm() / d0.norm()).item() < 1e-5
    rel = ((g1 - g0).norm() / g0.norm()).item()
    assert rel < 1e-5, rel
