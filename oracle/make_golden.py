#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference).

Runs only in the build container (the reference does not travel to the GPU box).  The
fixtures are data: seeded inputs, the reference's weights for small-width nets and the
reference's outputs/losses/gradients/updated parameters.  While generating, every oracle
function is checked against the reference (the oracle is pinned twice: here at generation
time, and by tests/test_oracle_golden.py from the committed vectors).

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NIRGAN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from model import networks as ref_networks                              # noqa: E402
from model.generator_inject import define_G_inject as ref_define_G_inject  # noqa: E402
from utils.remote_sensing_indices import RemoteSensingIndices as RefRS  # noqa: E402
import nirgan_oracle as O                                               # noqa: E402

OUT = os.environ.get("NIRGAN_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))      # (another directory: regenerate and diff against the committed set)
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def synth(B, H, W, seed):
    """SURVEY 8(d) synthetic tiles: rgb = 0.02+0.58U, nir = 0.05+0.75U, embeds ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    rgb = 0.02 + 0.58 * torch.rand(B, 3, H, W, generator=g)
    nir = 0.05 + 0.75 * torch.rand(B, 1, H, W, generator=g)
    emb = torch.randn(B, 256, generator=g)
    return rgb, nir, emb


def sd(net):
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def npd(prefix, d):
    return {prefix + k: v.detach().numpy() for k, v in d.items()}


def close(a, b, tol=1e-6, what=""):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(ref, 1e-30) + 1e-30, f"oracle != reference for {what}: err {err} ref {ref}"


def ref_train_batch(netG, netD, rgb, nir, lam_gan=1.0, lam_l1=100.0, lam_rs=0.0, rs_w=None, padding=0,
                    embeds=None):
    """The reference's per-batch sequence with its own modules and torch.optim.Adam.

    Order and formulas from model/pix2pix.py:165-257, :485-492 (cannot be imported: Lightning).
    """
    crit = ref_networks.GANLoss("lsgan")
    l1 = torch.nn.L1Loss()
    optD = torch.optim.Adam(netD.parameters(), lr=2e-4, betas=(0.5, 0.999))
    optG = torch.optim.Adam(netG.parameters(), lr=2e-4, betas=(0.5, 0.999))

    def fwd(x):
        if padding:
            x = torch.nn.functional.pad(x, (padding,) * 4, mode="reflect")
        y = netG(x) if embeds is None else netG(x, embeds)
        if padding:
            y = y[..., padding:-padding, padding:-padding]
        return y

    res = {}
    # optimizer_idx 0
    pred = fwd(rgb)
    res["pred"] = pred.detach().clone()
    pf = netD(torch.cat((rgb, pred), 1).detach())
    pr = netD(torch.cat((rgb, nir), 1))
    res["d_fake"], res["d_real"] = pf.detach().clone(), pr.detach().clone()
    lf, lr_ = crit(pf, False), crit(pr, True)
    loss_d = lf + lr_
    optD.zero_grad()
    optG.zero_grad()
    loss_d.backward()
    res["loss_D"], res["loss_D_fake"], res["loss_D_real"] = loss_d.detach(), lf.detach(), lr_.detach()
    res["grads_D"] = {k: p.grad.detach().clone() for k, p in netD.named_parameters()}
    assert all(p.grad is None or p.grad.abs().max() == 0 for p in netG.parameters())
    optD.step()
    res["params_D_after"] = sd(netD)
    # optimizer_idx 1 (Lightning toggle_optimizer: D params frozen)
    for p in netD.parameters():
        p.requires_grad_(False)
    pred = fwd(rgb)
    pf = netD(torch.cat((rgb, pred), 1))
    res["d_fake_gstep"] = pf.detach().clone()
    l_gan = crit(pf, True)
    l_l1 = l1(pred, nir)
    loss_g = l_gan * lam_gan + l_l1 * lam_l1
    res["loss_G_gan"], res["loss_G_l1"] = l_gan.detach(), l_l1.detach()
    if lam_rs > 0:
        l_rs = RefRS(mode="loss", criterion="l1").get_and_weight_losses(rgb, nir, pred, loss_config=rs_w)
        res["loss_G_rs"] = l_rs.detach()
        loss_g = loss_g + l_rs * lam_rs
    optG.zero_grad()
    loss_g.backward()
    res["loss_G"] = loss_g.detach()
    res["grads_G"] = {k: p.grad.detach().clone() for k, p in netG.named_parameters()}
    optG.step()
    res["params_G_after"] = sd(netG)
    for p in netD.parameters():
        p.requires_grad_(True)
    return res


def check_trainer(pG0, pD0, n_blocks, rgb, nir, res, **kw):
    embeds = kw.pop("embeds", None)
    tr = O.OracleTrainer(pG0, pD0, n_blocks, **kw)
    out = tr.step(rgb, nir, embeds)
    close(out["loss_D"], res["loss_D"], what="loss_D")
    close(out["loss_G"], res["loss_G"], what="loss_G")
    for k, v in res["grads_D"].items():
        close(tr.last["grads_D"][k], v, 1e-5, "grad D " + k)
    shadow = O.shadowed_bias_keys("G", n_blocks)
    for k, v in res["grads_G"].items():
        if k in shadow:
            continue
        close(tr.last["grads_G"][k], v, 1e-5, "grad G " + k)
    for k, v in res["params_D_after"].items():
        if k in O.shadowed_bias_keys("D"):
            continue
        close(tr.pD[k], v, 1e-6, "param D " + k)
    for k, v in res["params_G_after"].items():
        if k in shadow:
            continue
        close(tr.pG[k], v, 1e-6, "param G " + k)


# ---------------------------------------------------------------- F1: small-width whole nets
def f1(n_blocks, name, lam_rs=0.0, padding=0, H=32):
    torch.manual_seed(0)
    netG = ref_networks.define_G(3, 1, 8, f"resnet_{n_blocks}blocks", "instance", False, "normal", 0.02)
    netD = ref_networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    if lam_rs > 0:
        # the spectral indices are singular where pred + band ~ 0 and pred comes out of tanh in (-1, 1):
        # a positive output bias keeps pred in (0.5, 1) so that the whole-step vectors are well conditioned
        with torch.no_grad():
            list(netG.parameters())[-1].fill_(1.5)
    pG0, pD0 = sd(netG), sd(netD)
    rgb, nir, _ = synth(2, H, H, 1234)
    rs_w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333,
            "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}
    res = ref_train_batch(netG, netD, rgb, nir, lam_rs=lam_rs, rs_w=rs_w, padding=padding)
    # oracle cross-check
    close(O.px_forward(pG0, rgb, n_blocks, padding), res["pred"], what="G forward")
    close(O.discriminator_forward(pD0, torch.cat((rgb, res["pred"]), 1)), res["d_fake"], what="D fake")
    close(O.discriminator_forward(pD0, torch.cat((rgb, nir), 1)), res["d_real"], what="D real")
    check_trainer(pG0, pD0, n_blocks, rgb, nir, res, padding=padding, lambda_rs=lam_rs, rs_weights=rs_w)
    arrs = {"rgb": rgb.numpy(), "nir": nir.numpy(), "n_blocks": np.int32(n_blocks),
            "padding": np.int32(padding), "lambda_rs": np.float32(lam_rs)}
    arrs.update(npd("G0/", pG0))
    arrs.update(npd("D0/", pD0))
    for k in ("pred", "d_fake", "d_real", "d_fake_gstep", "loss_D", "loss_D_fake", "loss_D_real", "loss_G",
              "loss_G_gan", "loss_G_l1", "loss_G_rs"):
        if k in res:
            arrs[k] = res[k].numpy()
    arrs.update(npd("gD/", res["grads_D"]))
    arrs.update(npd("gG/", res["grads_G"]))
    arrs.update(npd("D1/", res["params_D_after"]))
    arrs.update(npd("G1/", res["params_G_after"]))
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, "loss_D", float(res["loss_D"]), "loss_G", float(res["loss_G"]))


# ---------------------------------------------------------------- legacy Pix2PixModel.optimize_parameters
def ref_legacy_batch(netG, netD, real_A, real_B, lam_l1=100.0):
    """The legacy loop with the reference's own modules and torch.optim.Adam, in the order of
    model/pix2pix_model.py:113-154 (forward, backward_D with the 0.5 factor, optimizer_D.step, D frozen, backward_G =
    GAN + lambda_L1 * L1 without a lambda_GAN, optimizer_G.step).  The class itself cannot be constructed as shipped
     and its two ``.backward()`` calls are commented out (:129, :141); they are executed here,
    as upstream pix2pix does and as optimize_parameters needs to do anything."""
    crit = ref_networks.GANLoss("lsgan")
    l1 = torch.nn.L1Loss()
    optG = torch.optim.Adam(netG.parameters(), lr=2e-4, betas=(0.5, 0.999))
    optD = torch.optim.Adam(netD.parameters(), lr=2e-4, betas=(0.5, 0.999))
    res = {}
    fake_B = netG(real_A)                                         # forward()
    res["fake_B"] = fake_B.detach().clone()
    for p in netD.parameters():
        p.requires_grad_(True)
    optD.zero_grad()
    pf = netD(torch.cat((real_A, fake_B), 1).detach())            # backward_D()
    lf = crit(pf, False)
    pr = netD(torch.cat((real_A, real_B), 1))
    lr_ = crit(pr, True)
    loss_d = (lf + lr_) * 0.5
    loss_d.backward()
    res["loss_D"], res["loss_D_fake"], res["loss_D_real"] = loss_d.detach(), lf.detach(), lr_.detach()
    res["grads_D"] = {k: p.grad.detach().clone() for k, p in netD.named_parameters()}
    optD.step()
    res["params_D_after"] = sd(netD)
    for p in netD.parameters():
        p.requires_grad_(False)
    optG.zero_grad()
    pf = netD(torch.cat((real_A, fake_B), 1))                     # backward_G()
    l_gan = crit(pf, True)
    l_l1 = l1(fake_B, real_B) * lam_l1
    loss_g = l_gan + l_l1
    loss_g.backward()
    res["loss_G"], res["loss_G_GAN"], res["loss_G_L1"] = loss_g.detach(), l_gan.detach(), l_l1.detach()
    res["grads_G"] = {k: p.grad.detach().clone() for k, p in netG.named_parameters()}
    optG.step()
    res["params_G_after"] = sd(netG)
    for p in netD.parameters():
        p.requires_grad_(True)
    return res


def f1_legacy(name):
    """Same nets and tiles as f1_g6_d.npz (seed 0 / 1234; its G0/*, D0/*, rgb, nir are the inputs: not stored twice)."""
    torch.manual_seed(0)
    netG = ref_networks.define_G(3, 1, 8, "resnet_6blocks", "instance", False, "normal", 0.02)
    netD = ref_networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    base = np.load(os.path.join(OUT, "f1_g6_d.npz"))
    for k, v in sd(netG).items():
        assert np.array_equal(base["G0/" + k], v.numpy()), k
    for k, v in sd(netD).items():
        assert np.array_equal(base["D0/" + k], v.numpy()), k
    pG0, pD0 = sd(netG), sd(netD)
    rgb, nir, _ = synth(2, 32, 32, 1234)
    assert np.array_equal(base["rgb"], rgb.numpy()) and np.array_equal(base["nir"], nir.numpy())
    res = ref_legacy_batch(netG, netD, rgb, nir)
    # oracle cross-check: the trainer with the D loss halved and lambda_GAN = 1
    tr = O.OracleTrainer(pG0, pD0, 6, d_loss_scale=0.5)
    out = tr.step(rgb, nir)
    close(out["loss_D"], res["loss_D"], what="legacy loss_D")
    close(out["loss_G"], res["loss_G"], what="legacy loss_G")
    for k, v in res["grads_D"].items():
        close(tr.last["grads_D"][k], v, 1e-5, "legacy grad D " + k)
    for k, v in res["params_G_after"].items():
        if k not in O.shadowed_bias_keys("G", 6):
            close(tr.pG[k], v, 1e-6, "legacy param G " + k)
    arrs = {"base": np.array("f1_g6_d.npz")}
    for k in ("fake_B", "loss_D", "loss_D_fake", "loss_D_real", "loss_G", "loss_G_GAN", "loss_G_L1"):
        arrs[k] = res[k].numpy()
    arrs.update(npd("gD/", res["grads_D"]))
    arrs.update(npd("gG/", res["grads_G"]))
    arrs.update(npd("D1/", res["params_D_after"]))
    arrs.update(npd("G1/", res["params_G_after"]))
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, "loss_D", float(res["loss_D"]), "loss_G", float(res["loss_G"]))


# ---------------------------------------------------------------- inject generator
def ns(**kw):
    return types.SimpleNamespace(**kw)


def f_inject(name, post_correction=False, post_correction_init=1.0):
    """post_correction=True (generator_inject.py:97-100,133-134: the prediction times a learnable scalar): fixture f1_inject_pc."""
    cfg = ns(base_configs=ns(input_nc=3, output_nc=1, ngf=8, netG="resnet_9blocks", norm="instance",
                             no_dropout=True, init_type="normal", init_gain=0.02),
             satclip=ns(satclip_inject_style="multiply", post_correction=post_correction, post_correction_init=post_correction_init,
                        scaling_param=True, scaling_param_init=0.01))
    torch.manual_seed(0)
    netG = ref_define_G_inject(cfg)
    # fc (256 -> 16384) is 4.2 M values: too large for a fixture.  Replace it by a seeded draw that the
    # tests regenerate (torch CPU generator, same torch build on the GPU box) and store its checksum.
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        netG.fc.weight.copy_(torch.randn(16384, 256, generator=g) * 0.02)
        netG.fc.bias.copy_(torch.randn(16384, generator=g) * 0.02)
        netG.scale_param.fill_(0.5)   # large enough that the modulation is visible at 1e-3
    netD = ref_networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    pG0, pD0 = sd(netG), sd(netD)
    rgb, nir, emb = synth(2, 40, 40, 100)     # inject map is 20x20 (seed chosen so that no ReLU input is within 1e-6 of 0)
    res = ref_train_batch(netG, netD, rgb, nir, embeds=emb)
    close(O.generator_inject_forward(pG0, rgb, emb, 9, post_correction=post_correction), res["pred"], what="inject forward")
    check_trainer(pG0, pD0, 9, rgb, nir, res, embeds=emb, inject_cfg={"style": "multiply", "use_scale": True, "post_correction": post_correction})
    arrs = {"rgb": rgb.numpy(), "nir": nir.numpy(), "embeds": emb.numpy(), "fc_seed": np.int64(4321),
            "fc_weight_sum": pG0["fc.weight"].double().sum().numpy(),
            "fc_weight_abs": pG0["fc.weight"].double().abs().sum().numpy()}
    arrs.update(npd("G0/", {k: v for k, v in pG0.items() if not k.startswith("fc.")}))
    arrs.update(npd("D0/", pD0))
    for k in ("pred", "loss_D", "loss_G", "loss_G_gan", "loss_G_l1"):
        arrs[k] = res[k].numpy()
    gG = res["grads_G"]
    arrs["g_scale_param"] = gG["scale_param"].numpy()
    arrs["g_fc_bias"] = gG["fc.bias"].numpy()
    arrs["g_fc_weight_rows0_8"] = gG["fc.weight"][:8].numpy()
    arrs["g_fc_weight_sum"] = gG["fc.weight"].double().sum().numpy()
    arrs["g_fc_weight_abs"] = gG["fc.weight"].double().abs().sum().numpy()
    arrs.update(npd("gG/", {k: v for k, v in gG.items() if not k.startswith("fc.")}))
    # the reference's torch.optim.Adam results (fc: first rows + checksums only, it is 4.2 M values)
    arrs.update(npd("D1/", res["params_D_after"]))
    G1 = res["params_G_after"]
    arrs.update(npd("G1/", {k: v for k, v in G1.items() if not k.startswith("fc.")}))
    arrs["G1_fc_bias"] = G1["fc.bias"].numpy()
    arrs["G1_fc_weight_rows0_8"] = G1["fc.weight"][:8].numpy()
    arrs["G1_fc_weight_sum"] = G1["fc.weight"].double().sum().numpy()
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, "loss_G", float(res["loss_G"]), "dscale", float(gG["scale_param"]))


# ---------------------------------------------------------------- F3: loss known-answer tests
def f3(name):
    g = torch.Generator().manual_seed(7)
    pred_d = torch.randn(2, 1, 30, 30, generator=g)
    crit = ref_networks.GANLoss("lsgan")
    arrs = {"pred_d": pred_d.numpy()}
    for real in (True, False):
        t = crit.get_target_tensor(pred_d, real)
        assert t.stride() == (0, 0, 0, 0)
        tag = "real" if real else "fake"
        arrs["mask_" + tag] = t.contiguous().numpy()
        p = pred_d.clone().requires_grad_(True)
        l = crit(p, real)
        l.backward()
        arrs["lsgan_" + tag] = l.detach().numpy()
        arrs["lsgan_grad_" + tag] = p.grad.numpy()
        close(O.lsgan_loss(pred_d, real), l, what="lsgan")
        assert torch.equal(O.gan_target_tensor(pred_d, real), t)
    rgb = 0.05 + 0.95 * torch.rand(2, 3, 64, 64, generator=g)
    nir = 0.01 + 0.99 * torch.rand(2, 1, 64, 64, generator=g)
    pred = 0.01 + 0.99 * torch.rand(2, 1, 64, 64, generator=g)
    arrs.update({"rgb": rgb.numpy(), "nir": nir.numpy(), "pred": pred.numpy()})
    p = pred.clone().requires_grad_(True)
    l = torch.nn.L1Loss()(p, nir)
    l.backward()
    arrs["l1"], arrs["l1_grad"] = l.detach().numpy(), p.grad.numpy()
    w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333,
         "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}
    for crit_name in ("l1", "l2"):
        rs = RefRS(mode="loss", criterion=crit_name)
        p = pred.clone().requires_grad_(True)
        l = rs.get_and_weight_losses(rgb, nir, p, loss_config=w)
        l.backward()
        arrs[f"rs_{crit_name}"], arrs[f"rs_{crit_name}_grad"] = l.detach().numpy(), p.grad.numpy()
        close(O.rs_weighted_loss(rgb, nir, pred, w, crit_name), l, what="rs " + crit_name)
        d = rs.get_and_weight_losses(rgb, nir, pred, mode="logging_dict")
        od = O.rs_logging_dict(rgb, nir, pred, crit_name)
        for k, v in d.items():
            arrs[f"rslog_{crit_name}/{k}"] = v.numpy()
            close(od[k], v, what=k)
    # default loss_config (None) path
    l = RefRS().get_and_weight_losses(rgb, nir, pred)
    arrs["rs_default"] = l.numpy()
    close(O.rs_weighted_loss(rgb, nir, pred), l, what="rs default")
    # index mode
    rsi = RefRS(mode="index")
    oi = O.rs_index_pairs(rgb, nir, pred, "index")
    for nm, fn in (("ndvi", rsi.ndvi_calculation), ("ndwi", rsi.ndwi_calculation), ("evi", rsi.evi_calculation),
                   ("gndvi", rsi.gndvi_calculation), ("savi", rsi.savi_calculation), ("msavi", rsi.msavi_calculation)):
        a, b = fn(rgb, nir, pred)
        close(oi[nm][0], a, what=nm)
        close(oi[nm][1], b, what=nm)
        arrs[f"index_{nm}_pred"] = b.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name)


# ---------------------------------------------------------------- F5: full-size checksums
def f5(name):
    arrs = {}
    for nb in (6, 9):
        torch.manual_seed(0)
        netG = ref_networks.define_G(3, 1, 64, f"resnet_{nb}blocks", "instance", False, "normal", 0.02)
        rgb, _, _ = synth(1, 256, 256, 1234)
        with torch.no_grad():
            y = netG(rgb)
            close(O.generator_forward(sd(netG), rgb, nb), y, what=f"full G{nb}")
        gi = torch.Generator().manual_seed(5)
        idx = torch.randint(0, 256 * 256, (64,), generator=gi)
        arrs[f"g{nb}_stats"] = np.array([y.mean(), y.std(), y.min(), y.max()], dtype=np.float64)
        arrs[f"g{nb}_idx"] = idx.numpy()
        arrs[f"g{nb}_samples"] = y.flatten()[idx].numpy()
        arrs[f"g{nb}_nparams"] = np.int64(sum(p.numel() for p in netG.parameters()))
        arrs[f"g{nb}_wsum"] = np.float64(sum(p.detach().double().sum() for p in netG.parameters()))
    torch.manual_seed(0)
    netD = ref_networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    x = torch.cat(synth(1, 256, 256, 1234)[:2], 1)
    with torch.no_grad():
        y = netD(x)
        close(O.discriminator_forward(sd(netD), x), y, what="full D")
    arrs["d_out"] = y.numpy()
    arrs["d_nparams"] = np.int64(sum(p.numel() for p in netD.parameters()))
    arrs["d_wsum"] = np.float64(sum(p.detach().double().sum() for p in netD.parameters()))
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name)


def f6(name):
    """SatCLIP harmonics: the reference's closed-form SH (model/satclip/positional_encoding/spherical_harmonics_closed_form.py,
    a self-contained file: math + torch) driven exactly as SphericalHarmonics.forward drives it
    (positional_encoding/spherical_harmonics.py:26-42; that module itself is not importable: its sibling
    spherical_harmonics_ylm.py is missing from the reference tree).  The Siren part of the fixture is the oracle's own
    output for seeded weights (text restatement, not pinned by the reference)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_sh_closed_form", os.path.join(REF, "model", "satclip", "positional_encoding", "spherical_harmonics_closed_form.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(77)
    lon = torch.rand(12, generator=g, dtype=torch.float64) * 360 - 180
    lat = torch.rand(12, generator=g, dtype=torch.float64) * 180 - 90
    lonlat = torch.stack((lon, lat), dim=-1)
    lonlat[0] = torch.tensor([0.0, 0.0], dtype=torch.float64)
    lonlat[1] = torch.tensor([-180.0, 89.5], dtype=torch.float64)
    lonlat[2] = torch.tensor([179.99, -89.5], dtype=torch.float64)
    arrs = {"lonlat": lonlat.numpy()}
    for L in (10, 16):
        phi, theta = torch.deg2rad(lonlat[:, 0] + 180), torch.deg2rad(lonlat[:, 1] + 90)
        Y = []
        for l in range(L):
            for m in range(-l, l + 1):
                y = mod.SH(m, l, phi, theta)
                if isinstance(y, float):
                    y = y * torch.ones_like(phi)
                Y.append(y)
        Y = torch.stack(Y, dim=-1)
        mine = O.spherical_harmonics(lonlat, L)
        close(mine, Y, 1e-13, f"spherical harmonics L={L}")
        arrs[f"Y{L}"] = Y.numpy()
    # Siren (oracle's restatement) on seeded weights: 100 -> 64 -> 64 -> 32
    g = torch.Generator().manual_seed(78)
    p = {}
    dims = [(64, 100), (64, 64)]
    for i, (o, n) in enumerate(dims):
        std = (1 / n) if i == 0 else (np.sqrt(6 / n) / 1.0)
        p[f"nnet.layers.{i}.weight"] = (torch.rand(o, n, generator=g, dtype=torch.float64) * 2 - 1) * std
        p[f"nnet.layers.{i}.bias"] = (torch.rand(o, generator=g, dtype=torch.float64) * 2 - 1) * std
    std = np.sqrt(6 / 64)
    p["nnet.last_layer.weight"] = (torch.rand(32, 64, generator=g, dtype=torch.float64) * 2 - 1) * std
    p["nnet.last_layer.bias"] = (torch.rand(32, generator=g, dtype=torch.float64) * 2 - 1) * std
    arrs.update({"siren/" + k: v.numpy() for k, v in p.items()})
    arrs["siren_out"] = O.location_encoder_forward(p, lonlat, 10, 2).numpy()
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name)


def f7(name, L=10):
    """SatCLIP 'analytic' harmonics (the SphericalHarmonics default, positional_encoding/spherical_harmonics.py:10,24-25).
    The tabulated file spherical_harmonics_ylm.py is absent from the reference tree (.MISSING_LARGE_BLOBS) but its generator
    spherical_harmonics_generate_ylms.py is there.  That script prints all 101^2 functions at import, so only its
    imports, the two sympy symbols and ``calc_ylm`` (:11-36) are executed here (taken from its AST, unmodified); each
    function is then evaluated exactly as the generated file would: ``str(calc_ylm(l, m).evalf())`` with torch's cos / sin
    in float64, through SphericalHarmonics.forward's loop (spherical_harmonics.py:26-42)."""
    import ast
    path = os.path.join(REF, "model", "satclip", "positional_encoding", "spherical_harmonics_generate_ylms.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") in ("theta", "phi"))
            or (isinstance(n, ast.FunctionDef) and n.name == "calc_ylm")]
    env = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), env)
    lonlat = torch.from_numpy(np.load(os.path.join(OUT, "f6_locenc.npz"))["lonlat"])
    phi, theta = torch.deg2rad(lonlat[:, 0] + 180), torch.deg2rad(lonlat[:, 1] + 90)
    Y = []
    for l in range(L):
        for m in range(-l, l + 1):
            src = str(env["calc_ylm"](l, m).evalf())
            y = eval(src, {"cos": torch.cos, "sin": torch.sin, "theta": theta, "phi": phi})
            if isinstance(y, float):
                y = y * torch.ones_like(phi)
            Y.append(y)
    Y = torch.stack(Y, dim=-1)
    mine = O.spherical_harmonics(lonlat, L, "analytic-generator-text")
    close(mine, Y, 1e-12, f"analytic spherical harmonics L={L}")
    cf = O.spherical_harmonics(lonlat, L, "closed-form")
    print("analytic vs closed-form: max diff", float((Y - cf).abs().max()), "of", float(Y.abs().max()))
    np.savez_compressed(os.path.join(OUT, name), lonlat=lonlat.numpy(), **{f"Y{L}": Y.numpy()})
    print("wrote", name)


def _load_ref_by_path(modname, relpath, stubs=()):
    """A reference source file loaded by path with named modules pre-seeded as EMPTY modules.  Only for imports the code under test
    never touches: a stub for an unused import is not a stand-in for arithmetic (nothing that computes is replaced)."""
    import importlib.util
    saved = {}
    for name in stubs:
        saved[name] = sys.modules.get(name)
        m = types.ModuleType(name)
        m.__path__ = []                                       # (importable as a package too)
        sys.modules[name] = m
    try:
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for name, old in saved.items():
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
    return mod


def f8(name):
    """utils/losses.py::emd_loss (:64-78) -- pure torch; the file's top-level `import kornia` (used by ssim_loss only, absent here) is
    satisfied with an empty module.  Values and autograd gradients wrt `pred` in float32 (as the training step would evaluate it) and
    float64, three shapes."""
    mod = _load_ref_by_path("ref_losses", os.path.join("utils", "losses.py"), stubs=("kornia",))
    arrs = {}
    for i, shape in enumerate(((2, 1, 16, 16), (3, 1, 8, 12), (1, 1, 32, 32))):
        g = torch.Generator().manual_seed(80 + i)
        pred = torch.rand(*shape, generator=g, dtype=torch.float64)
        target = (pred + 0.2 * torch.randn(*shape, generator=g, dtype=torch.float64)).clamp(0, 1)
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            p = pred.to(dt).clone().requires_grad_(True)
            v = mod.emd_loss(p, target.to(dt))
            gr, = torch.autograd.grad(v, p)
            mine = O.emd_loss(pred.to(dt), target.to(dt))
            close(mine, v.detach(), 1e-12 if dt == torch.float64 else 1e-6, f"emd_loss {shape} {tag}")
            arrs[f"value_{tag}_{i}"] = v.detach().numpy()
            arrs[f"grad_{tag}_{i}"] = gr.numpy()
        arrs[f"pred_{i}"] = pred.numpy()
        arrs[f"target_{i}"] = target.numpy()
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name)


def f9(name):
    """model/satclip/location_encoder.py::SirenNet (:73-151) -- pure torch + einops; the file's top-level
    `import model.satclip.positional_encoding as PE` (used by get_positional_encoding only; the package pulls in pytorch_lightning and
    the missing spherical_harmonics_ylm.py) is satisfied with empty modules.  A seeded SirenNet(100 -> 64 -> 64 -> 32) in float64,
    eval mode (the reference runs the encoder frozen: model/pix2pix.py builds it through SatClIP_wrapper and never trains it): its
    state_dict and its outputs on seeded inputs; the oracle's siren_forward is checked against it here."""
    mod = _load_ref_by_path("ref_location_encoder", os.path.join("model", "satclip", "location_encoder.py"),
                            stubs=("model.satclip", "model.satclip.positional_encoding"))
    torch.manual_seed(90)
    net = mod.SirenNet(dim_in=100, dim_hidden=64, dim_out=32, num_layers=2).double().eval()
    g = torch.Generator().manual_seed(91)
    x = torch.randn(12, 100, generator=g, dtype=torch.float64) * 0.3
    with torch.no_grad():
        y = net(x)
    p = {"nnet." + k: v.detach().clone() for k, v in net.state_dict().items()}
    close(O.siren_forward(p, x, 2), y, 1e-13, "SirenNet forward")
    arrs = {"x": x.numpy(), "y": y.numpy()}
    # the whole LocationEncoder.forward (:267-274) from reference parts: the reference's closed-form harmonics of fixture f6's
    # coordinates (f6 holds them as the reference computed them) through this reference SirenNet
    z6 = np.load(os.path.join(OUT, "f6_locenc.npz"))
    with torch.no_grad():
        y6 = net(torch.from_numpy(z6["Y10"]))
    close(O.location_encoder_forward(p, torch.from_numpy(z6["lonlat"]), 10, 2, "closed-form"), y6, 1e-12, "LocationEncoder forward")
    arrs["lonlat"] = z6["lonlat"]
    arrs["y_lonlat"] = y6.numpy()
    arrs.update({"siren/" + k: v.numpy() for k, v in p.items()})
    # a wider / deeper one (the SatCLIP encoder's own sizes are 512 x 2 -> 256): 100 -> 3 x 48 -> 16
    torch.manual_seed(92)
    net3 = mod.SirenNet(dim_in=100, dim_hidden=48, dim_out=16, num_layers=3).double().eval()
    with torch.no_grad():
        y3 = net3(x)
    p3 = {"nnet." + k: v.detach().clone() for k, v in net3.state_dict().items()}
    close(O.siren_forward(p3, x, 3), y3, 1e-13, "SirenNet forward (3 layers)")
    arrs["y3"] = y3.numpy()
    arrs.update({"siren3/" + k: v.numpy() for k, v in p3.items()})
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name)



def to_ns(x):
    if isinstance(x, dict):
        return types.SimpleNamespace(**{k: to_ns(v) for k, v in x.items()})
    return x


def f0(name):
    """configs[0] plumbing: the PARSED key/value tree of the reference's two training configs (configs/config_px2px.yaml,
    configs/config_px2px_SatCLIP.yaml -- data, not the files' text) and the state_dict key list + shapes of the modules Px2Px_PL builds
    from each (model/pix2pix.py:18-63: netG through define_G / define_G_inject, netD through define_D, criterionGAN), for the YAML's
    own netG and for resnet_6blocks (BASELINE.json configs[0] / configs[1]).  Px2Px_PL itself cannot be instantiated here (Lightning is
    absent) and ``satclip_model.*`` needs model/satclip/satclip-resnet50-l10.ckpt (git-ignored upstream): both are named in the fixture."""
    import json
    import yaml
    out = {}
    for fname in ("config_px2px.yaml", "config_px2px_SatCLIP.yaml"):
        with open(os.path.join(REF, "configs", fname)) as f:
            tree = yaml.safe_load(f)
        entry = {"tree": tree, "state_dict": {}, "not_constructible_here": []}
        for netG_name in (tree["base_configs"]["netG"], "resnet_6blocks"):
            t = yaml.safe_load(yaml.safe_dump(tree))
            t["base_configs"]["netG"] = netG_name
            o = t["base_configs"]
            torch.manual_seed(0)
            sc = t.get("satclip", {})
            if sc.get("use_satclip") and sc.get("satclip_style") == "inject":
                try:
                    netG = ref_define_G_inject(to_ns(t))      # attribute access on the tree, as OmegaConf gives the reference
                except NotImplementedError as e:              # (model/generator_inject.py:199: only resnet_9blocks with SatCLIP)
                    entry["state_dict"][netG_name] = {"raises": "NotImplementedError", "message": str(e)}
                    continue
            else:
                netG = ref_networks.define_G(o["input_nc"], o["output_nc"], o["ngf"], o["netG"], o["norm"], not o["no_dropout"], o["init_type"], o["init_gain"])
            netD = ref_networks.define_D(o["input_nc"] + o["output_nc"], o["ndf"], o["netD"], o["n_layers_D"], o["norm"], o["init_type"], o["init_gain"])
            crit = ref_networks.GANLoss(o["gan_mode"])
            keys = {}
            for prefix, mod in (("netG.", netG), ("netD.", netD), ("criterionGAN.", crit)):
                for k, v in mod.state_dict().items():
                    keys[prefix + k] = list(v.shape)
            entry["state_dict"][netG_name] = keys
        if tree.get("satclip", {}).get("use_satclip"):
            entry["not_constructible_here"].append("satclip_model.* (SatClIP_wrapper needs model/satclip/satclip-resnet50-l10.ckpt, git-ignored upstream)")
        entry["not_constructible_here"].append("Px2Px_PL itself (pytorch_lightning is not installed): the key list is assembled from the modules its __init__ builds")
        out[fname] = entry
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", name, {k: {g: ("raises" if "raises" in v else len(v)) for g, v in e["state_dict"].items()} for k, e in out.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "f0":
        f0("f0_config.json")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("f1_legacy", "f7", "f8", "f9", "f1_inject_pc"):       # add one fixture without touching the others
        {"f1_legacy": lambda: f1_legacy("f1_legacy.npz"), "f7": lambda: f7("f7_sh_analytic.npz"), "f8": lambda: f8("f8_emd.npz"),
         "f9": lambda: f9("f9_siren.npz"), "f1_inject_pc": lambda: f_inject("f1_inject_pc.npz", True, 0.8)}[sys.argv[1]]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "f6":       # add the location-encoder fixture without touching the others
        f6("f6_locenc.npz")
        sys.exit(0)
    f1(6, "f1_g6_d.npz")
    f1(9, "f1_g9_rs_pad.npz", lam_rs=1.0, padding=10, H=44)
    f_inject("f1_inject.npz")
    f_inject("f1_inject_pc.npz", True, 0.8)
    f3("f3_losses.npz")
    f5("f5_fullsize.npz")
    f6("f6_locenc.npz")
    f1_legacy("f1_legacy.npz")
    f7("f7_sh_analytic.npz")
    f8("f8_emd.npz")
    f9("f9_siren.npz")
    f0("f0_config.json")
