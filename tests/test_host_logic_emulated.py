"""Host logic (descriptor geometry, halo plumbing, plan order, flat parameters, fused trainer,
autograd bridges) against the golden vectors of the reference, with the C ABI served by the
numpy emulator of tests/emu_backend.py.  No GPU, no HIP code runs here: these tests pin the
*host side*; the kernels themselves are pinned by the -m gpu tests through the same ABI."""
import os
import types

import numpy as np
import pytest
import torch

import nirgan_oracle as O
from emu_backend import EmuBackend
from nirgan_hip import lib as L

torch.set_num_threads(4)


@pytest.fixture()
def emu():
    be = EmuBackend()
    L.set_backend(be)
    yield be
    L.set_backend(None)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def sub(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in z.items() if k.startswith(prefix)}


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(ref, 1e-20), f"{what}: err {err:.3e} ref {ref:.3e}"


def adam_close(p_new, p_ref, grad, p_old, what, lr=2e-4):
    """First Adam step: p1 = p0 - lr*g/(|g| + eps').  Exact given OUR gradient (pins the Adam kernel);
    against the reference's p1 only where |g| >> eps (elsewhere the step is a sign flip of rounding noise)."""
    p_new, p_ref, grad = (torch.as_tensor(t).float().cpu() for t in (p_new, p_ref, grad))
    p_old = torch.as_tensor(p_old).float().clone()
    m, v = torch.zeros_like(p_old), torch.zeros_like(p_old)
    O.adam_step(p_old, grad, m, v, 1, lr=lr, b1=0.5)
    close(p_new, p_old, 1e-6, what + " (adam kernel)")
    live = grad.abs() > 1e-6 * grad.abs().max().clamp_min(1e-30)
    err = ((p_new - p_ref).abs() * live).max().item()
    assert err <= 0.02 * lr, f"{what}: {err:.3e}"


def make_nets(z, n_blocks, ngf=8):
    from model import networks
    netG = networks.define_G(3, 1, ngf, f"resnet_{n_blocks}blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, ngf, "basic", 3, "instance", "normal", 0.02)
    netG.load_state_dict(sub(z, "G0/"))
    netD.load_state_dict(sub(z, "D0/"))
    return netG, netD


def test_state_dict_keys_and_rng_parity(golden_dir):
    """Same keys/layouts as the reference and the same weights for the same seed (networks.py:68-117)."""
    from model import networks
    z = load(golden_dir, "f5_fullsize.npz")
    for nb in (6, 9):
        torch.manual_seed(0)
        net = networks.define_G(3, 1, 64, f"resnet_{nb}blocks", "instance", False, "normal", 0.02)
        assert sum(p.numel() for p in net.parameters()) == int(z[f"g{nb}_nparams"])
        wsum = float(sum(p.detach().double().sum() for p in net.parameters()))
        assert abs(wsum - float(z[f"g{nb}_wsum"])) < 1e-9
    torch.manual_seed(0)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    assert sum(p.numel() for p in netD.parameters()) == int(z["d_nparams"])
    assert abs(float(sum(p.detach().double().sum() for p in netD.parameters())) - float(z["d_wsum"])) < 1e-9
    g = load(golden_dir, "f1_g6_d.npz")
    torch.manual_seed(0)
    small = networks.define_G(3, 1, 8, "resnet_6blocks", "instance", False, "normal", 0.02)
    sd = small.state_dict()
    ref = sub(g, "G0/")
    assert list(sd.keys()) == list(ref.keys())
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k


def test_reference_error_conventions():
    from model import networks
    from utils.remote_sensing_indices import RemoteSensingIndices
    with pytest.raises(NotImplementedError):
        networks.define_G(3, 1, 8, "resnet_3blocks", "instance")
    with pytest.raises(NotImplementedError):
        networks.define_D(4, 8, "fancy", 3, "instance")
    with pytest.raises(NotImplementedError):
        networks.get_norm_layer("layer")
    with pytest.raises(NotImplementedError):
        networks.GANLoss("hinge")
    with pytest.raises(NotImplementedError):
        RemoteSensingIndices(mode="loss", criterion="huber")
    with pytest.raises(AssertionError):
        RemoteSensingIndices(mode="train")


def test_no_cpu_fallback():
    """Without the test seam a CPU tensor must be refused loudly."""
    from model import networks
    net = networks.define_G(3, 1, 8, "resnet_6blocks", "instance")
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 3, 32, 32))


@pytest.mark.parametrize("name", ["f1_g6_d.npz", "f1_g9_rs_pad.npz"])
def test_fused_trainer_matches_reference(emu, golden_dir, name):
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, name)
    nb, pad, lam_rs = int(z["n_blocks"]), int(z["padding"]), float(z["lambda_rs"])
    netG, netD = make_nets(z, nb)
    rs_w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333, "lambda_savi": 0.0,
            "lambda_msavi": 0.0, "lambda_gndvi": 0.0}
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, lambda_rs=lam_rs, rs_weights=rs_w, padding=pad)
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    view = tr.step(rgb, nir)
    out = view.as_dict()
    close(tr.G.pred, z["pred"], 2e-5, "pred")
    close(out["loss_D"], z["loss_D"], 1e-5, "loss_D")
    close(out["loss_G"], z["loss_G"], 1e-5, "loss_G")
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    for k, v in sub(z, "gD/").items():
        if k not in O.shadowed_bias_keys("D"):
            close(gD[k], v, 2e-4, "gD " + k)
    shadow = O.shadowed_bias_keys("G", nb)
    for k, v in sub(z, "gG/").items():
        if k not in shadow:
            close(gG[k], v, 2e-4, "gG " + k)
    pD, pG = dict(netD.named_parameters()), dict(netG.named_parameters())
    for k, v in sub(z, "D1/").items():
        if k not in O.shadowed_bias_keys("D"):
            adam_close(pD[k], v, gD[k], z["D0/" + k], "D1 " + k)
    for k, v in sub(z, "G1/").items():
        if k not in shadow:
            adam_close(pG[k], v, gG[k], z["G0/" + k], "G1 " + k)
    # a second step runs on the same buffers (halo invariants hold, weights re-packed)
    ref = O.OracleTrainer(sub(z, "G0/"), sub(z, "D0/"), nb, padding=pad, lambda_rs=lam_rs, rs_weights=rs_w)
    ref.step(rgb, nir)
    o2 = ref.step(rgb, nir)
    v2 = tr.step(rgb, nir).as_dict()
    close(v2["loss_D"], o2["loss_D"], 1e-4, "loss_D step 2")
    close(v2["loss_G"], o2["loss_G"], 1e-4, "loss_G step 2")


def regen_fc(z):
    g = torch.Generator().manual_seed(int(z["fc_seed"]))
    return torch.randn(16384, 256, generator=g) * 0.02, torch.randn(16384, generator=g) * 0.02


def inject_config(post_correction=False, post_correction_init=1.0):
    ns = types.SimpleNamespace
    return ns(base_configs=ns(input_nc=3, output_nc=1, ngf=8, netG="resnet_9blocks", norm="instance", no_dropout=True,
                              init_type="normal", init_gain=0.02),
              satclip=ns(satclip_inject_style="multiply", post_correction=post_correction, post_correction_init=post_correction_init,
                         scaling_param=True, scaling_param_init=0.01))


@pytest.mark.parametrize("name,pc", [("f1_inject.npz", False), ("f1_inject_pc.npz", True)])
def test_inject_generator_trainer(emu, golden_dir, capsys, name, pc):
    """(f1_inject_pc: post_correction=True, generator_inject.py:97-100,133-134 -- the prediction times a learnable scalar, init 0.8)"""
    from model import networks
    from model.generator_inject import define_G_inject
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, name)
    netG = define_G_inject(inject_config(pc, 0.8))
    sd = sub(z, "G0/")
    sd["fc.weight"], sd["fc.bias"] = regen_fc(z)
    netG.load_state_dict(sd)
    netD = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    netD.load_state_dict(sub(z, "D0/"))
    rgb, nir, emb = (torch.from_numpy(z[k]) for k in ("rgb", "nir", "embeds"))
    tr = Pix2PixTrainer(netG, netD, n_blocks=9, inject={"style": "multiply", "use_scale": True, "post_correction": pc})
    out = tr.step(rgb, nir, emb).as_dict()
    close(tr.G.pred, z["pred"], 2e-5, "pred")
    close(out["loss_G"], z["loss_G"], 1e-5, "loss_G")
    g = tr.flatG.grad_views()
    if pc:
        close(g["post_correction_param"].reshape(()), z["gG/post_correction_param"], 2e-4, "dpost_correction_param")
    close(g["scale_param"], z["g_scale_param"], 2e-4, "dscale")
    close(g["fc.bias"], z["g_fc_bias"], 2e-4, "dfc.bias")
    close(g["fc.weight"][:8], z["g_fc_weight_rows0_8"], 2e-4, "dfc.weight")
    shadow = O.shadowed_bias_keys("G", 9)
    for k, v in sub(z, "gG/").items():
        if k not in shadow:
            close(g[k], v, 2e-4, "gG " + k)


def test_autograd_bridges_follow_the_reference_loop(emu, golden_dir):
    """training_step-style use: modules + losses + any optimizer, as Lightning drives the reference."""
    from model import networks
    from model.pix2pix import HipL1Loss
    from nirgan_hip.optim import HipAdam
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    crit, l1 = networks.GANLoss("lsgan"), HipL1Loss()
    optD = HipAdam(netD.parameters(), lr=2e-4, betas=(0.5, 0.999), net=netD)
    optG = HipAdam(netG.parameters(), lr=2e-4, betas=(0.5, 0.999), net=netG)
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    pred = netG(rgb)
    close(pred, z["pred"], 2e-5, "pred")
    loss_d = crit(netD(torch.cat((rgb, pred), 1).detach()), False) + crit(netD(torch.cat((rgb, nir), 1)), True)
    close(loss_d, z["loss_D"], 1e-5, "loss_D")
    optD.zero_grad()
    loss_d.backward()
    for k, p in netD.named_parameters():
        if k not in O.shadowed_bias_keys("D"):
            close(p.grad, z["gD/" + k], 2e-4, "gD " + k)
    assert all(p.grad is None for p in netG.parameters())
    optD.step()
    for p in netD.parameters():
        p.requires_grad_(False)
    pred = netG(rgb)
    loss_g = crit(netD(torch.cat((rgb, pred), 1)), True) * 1.0 + l1(pred, nir) * 100.0
    close(loss_g, z["loss_G"], 1e-5, "loss_G")
    optG.zero_grad()
    loss_g.backward()
    shadow = O.shadowed_bias_keys("G", 6)
    for k, p in netG.named_parameters():
        if k not in shadow:
            close(p.grad, z["gG/" + k], 2e-4, "gG " + k)
    optG.step()
    for k, p in netG.named_parameters():
        if k not in shadow:
            close(p, z["G1/" + k], 1e-5, "G1 " + k)
    # label mask: bit exact, stride-0 expansion of the 0-dim buffer (networks.py:241-256)
    f3 = load(golden_dir, "f3_losses.npz")
    pd_ = torch.from_numpy(f3["pred_d"])
    for real, tag in ((True, "real"), (False, "fake")):
        t = crit.get_target_tensor(pd_, real)
        assert t.stride() == (0, 0, 0, 0)
        assert t.contiguous().numpy().tobytes() == f3["mask_" + tag].tobytes()


def test_rs_indices_module(emu, golden_dir):
    from utils.remote_sensing_indices import RemoteSensingIndices
    z = load(golden_dir, "f3_losses.npz")
    rgb, nir, pred = (torch.from_numpy(z[k]) for k in ("rgb", "nir", "pred"))
    w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333, "lambda_savi": 0.0, "lambda_msavi": 0.0,
         "lambda_gndvi": 0.0}
    for c in ("l1", "l2"):
        rs = RemoteSensingIndices(mode="loss", criterion=c)
        p = pred.clone().requires_grad_(True)
        l = rs.get_and_weight_losses(rgb, nir, p, loss_config=w)
        l.backward()
        close(l.detach(), z["rs_" + c], 1e-5, "rs " + c)
        close(p.grad, z[f"rs_{c}_grad"], 1e-4, "rs grad " + c)
        d = rs.get_and_weight_losses(rgb, nir, pred, mode="logging_dict")
        for k, v in d.items():
            close(v, z[f"rslog_{c}/{k}"], 1e-5, k)
    close(RemoteSensingIndices().get_and_weight_losses(rgb, nir, pred), z["rs_default"], 1e-5, "default")
    with pytest.raises(NotImplementedError):
        RemoteSensingIndices().get_and_weight_losses(rgb, nir, pred, mode="nope")
    idx = RemoteSensingIndices(mode="index")
    for n in ("ndvi", "ndwi", "evi", "gndvi", "savi", "msavi"):
        _, b = getattr(idx, n + "_calculation")(rgb, nir, pred)
        close(b, z[f"index_{n}_pred"], 1e-5, n)


def test_px2px_pl_surface(emu, golden_dir):
    """configs[0]: the shipped YAML keys, 6-block override, plumbing of Px2Px_PL (losses finite, keys, one step)."""
    from model.pix2pix import Px2Px_PL
    from utils.config import to_attr
    cfg = to_attr({
        "base_configs": {"isTrain": True, "input_nc": 3, "output_nc": 1, "ngf": 8, "ndf": 8, "netD": "basic",
                         "netG": "resnet_6blocks", "norm": "instance", "no_dropout": True, "init_type": "normal",
                         "init_gain": 0.02, "n_layers_D": 3, "gan_mode": "lsgan", "lr": 0.0002, "beta1": 0.5,
                         "lambda_GAN": 1.0, "lambda_L1": 100.0, "lambda_ssim": 0.0, "lambda_hist": 0.0,
                         "lambda_rs_losses": 0.0, "rs_losses_criterium": "l1",
                         "internal_rs_loss_weights": {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333}},
        "satclip": {"use_satclip": False},
        "Schedulers": {"metric": "val/L1", "patience_g": 25, "patience_d": 25},
        "Data": {"padding": True, "padding_amount": 10}})
    m = Px2Px_PL(cfg)
    for attr in ("netG", "netD", "criterionGAN", "criterionL1", "satclip", "config", "opt"):
        assert hasattr(m, attr)
    keys = list(m.state_dict().keys())
    assert "netG.model.1.weight" in keys and "netD.model.11.bias" in keys and "criterionGAN.real_label" in keys
    g = torch.Generator().manual_seed(3)
    batch = {"rgb": 0.02 + 0.58 * torch.rand(2, 3, 40, 40, generator=g), "nir": 0.05 + 0.75 * torch.rand(2, 1, 40, 40, generator=g)}
    m.train()
    l0 = m.training_step(batch, 0, 0)
    l1 = m.training_step(batch, 0, 1)
    assert torch.isfinite(l0) and torch.isfinite(l1)
    (od, og), scheds = m.configure_optimizers()
    assert len(scheds) == 2 and scheds[0]["monitor"] == "val/L1"
    out = m.train_batch(batch).as_dict()
    assert abs(out["loss_D"] - float(l0.detach())) < 1e-4 * abs(float(l0.detach()))
    # loss_G of the fused batch is taken against the UPDATED discriminator (optimizer 0 stepped first)
    assert np.isfinite(out["loss_G"]) and abs(out["loss_G_l1"] * 100.0 - float(l1.detach())) < 2.0
    m.eval()
    with pytest.raises(AssertionError):
        m.training_step(batch, 0, 0)
    p = m.predict_step(batch["rgb"])
    assert p.shape == (2, 1, 40, 40)
    # validation scalars (pix2pix.py:259-283): the four image metrics, computed by the fused device pass
    m.logged.clear()
    v = m.validation_step(batch, 0)
    ref = O.calculate_metrics(p, batch["nir"], "val")
    for k in ("val/L1", "val/L2", "val/PSNR", "val/SSIM"):
        assert abs(float(m.logged[k]) - ref[k]) <= 1e-4 * abs(ref[k]), k
    assert abs(v - ref["val/L1"]) <= 1e-4 * ref["val/L1"]
    m.train()
    with pytest.raises(AssertionError):
        m.predict_step(batch["rgb"])
    m.logged.clear()
    m.training_step(batch, 10, 0)                      # every 10th batch, first optimizer pass (pix2pix.py:181-185)
    assert "train/SSIM" in m.logged and "train/PSNR" in m.logged
    m.logged.clear()
    m.training_step(batch, 11, 0)
    assert "train/SSIM" not in m.logged
    # lambda_ssim > 0 (pix2pix.py:233-237): the autograd path adds lambda * ssim_loss(pred, nir) and logs it; the fused trainer gets the
    # same weight; lambda_hist > 0 cannot run in the reference either (undefined hist_loss) and is refused
    cfg.base_configs.lambda_ssim = 25.0
    m2 = Px2Px_PL(cfg)
    m2.load_state_dict(m.state_dict())
    m2.train()
    lg = m2.training_step(batch, 0, 1)
    want = m.training_step(batch, 0, 1).detach() + 25.0 * float(m2.logged["model_loss/generator_ssim"])
    assert abs(float(lg.detach()) - float(want)) <= 1e-5 * abs(float(want))
    lg.backward()
    assert all(torch.isfinite(p_.grad).all() for p_ in m2.netG.parameters() if p_.grad is not None)
    assert m2.fused_trainer().lambda_ssim == 25.0 and "loss_G_ssim" in m2.train_batch(batch).as_dict()
    cfg.base_configs.lambda_hist = 1.0
    with pytest.raises(NotImplementedError):
        Px2Px_PL(cfg)


# ---------------------------------------------------------------------------------- operand precision modes
def _nchw(h):
    t = h.t[:, h.pad:h.pad + h.H, h.pad:h.pad + h.W, :] if h.pad else h.t
    return t.permute(0, 3, 1, 2).contiguous()


def test_bf16_operand_mode_follows_the_bf16_restatement(emu, golden_dir):
    """precision='bf16' (BASELINE.json configs[4]): every contraction rounds both operands to bf16 and accumulates in
    fp32.  The reference has no such path (parity unpinned against it); the host logic is checked against the oracle's
    restatement of exactly that rule.  Rounding is discontinuous, so two evaluations that differ by fp32 summation
    noise round a few activations to neighbouring bf16 values and drift apart layer by layer up to the bf16 noise
    level itself.  Therefore: (1) teacher-forced, every generator convolution fed the engine's own input reproduces the
    engine's output to 1e-5; (2) end to end the two stay well inside the distance between the bf16 and fp32 results."""
    import torch.nn.functional as F
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision="bf16")
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    out = tr.step(rgb, nir).as_dict()
    pG = sub(z, "G0/")
    eng = tr.G
    with O.operand_precision("bf16"):
        ref = O.OracleTrainer(pG, sub(z, "D0/"), 6)
        o = ref.step(rgb, nir)
        # (1) teacher-forced layers (weights are the pre-step ones: the engine's buffers hold the step's forward)
        y = O._conv2d(F.pad(rgb, (3, 3, 3, 3), mode="reflect"), pG["model.1.weight"], pG["model.1.bias"])
        close(_nchw(eng.L1.y), y, 1e-5, "L1")
        y = O._conv2d(_nchw(eng.L1.out), pG["model.4.weight"], pG["model.4.bias"], stride=2, padding=1)
        close(_nchw(eng.L2.y), y, 1e-5, "L2")
        y = O._conv2d(_nchw(eng.L2.out), pG["model.7.weight"], pG["model.7.bias"], stride=2, padding=1)
        close(_nchw(eng.L3.y), y, 1e-5, "L3")
        x = eng.L3.out
        for i, c1, c2 in eng.blocks:
            y = O._conv2d(F.pad(_nchw(x), (1, 1, 1, 1), mode="reflect"), pG[f"model.{i}.conv_block.1.weight"], pG[f"model.{i}.conv_block.1.bias"])
            close(_nchw(c1.y), y, 1e-5, f"block {i} conv 1")
            y = O._conv2d(F.pad(_nchw(c1.out), (1, 1, 1, 1), mode="reflect"), pG[f"model.{i}.conv_block.5.weight"], pG[f"model.{i}.conv_block.5.bias"])
            close(_nchw(c2.y), y, 1e-5, f"block {i} conv 2")
            x = c2.out
        i1, i2 = eng.lay["up"]
        y = O._conv_transpose2d(_nchw(x), pG[f"model.{i1}.weight"], pG[f"model.{i1}.bias"], stride=2, padding=1, output_padding=1)
        close(_nchw(eng.U1.y), y, 1e-5, "U1")
        y = O._conv_transpose2d(_nchw(eng.U1.out), pG[f"model.{i2}.weight"], pG[f"model.{i2}.bias"], stride=2, padding=1, output_padding=1)
        close(_nchw(eng.U2.y), y, 1e-5, "U2")
        il = eng.lay["last"]
        y = torch.tanh(O._conv2d(F.pad(_nchw(eng.U2.out), (3, 3, 3, 3), mode="reflect"), pG[f"model.{il}.weight"], pG[f"model.{il}.bias"]))
        close(eng.pred, y, 1e-5, "last + tanh")
    # (2) end to end
    pred32 = torch.from_numpy(z["pred"])
    noise = (ref.last["pred"] - pred32).abs().max().item()
    assert noise > 1e-3, "the bf16 restatement should differ visibly from fp32"
    assert (tr.G.pred.reshape(-1) - ref.last["pred"].reshape(-1)).abs().max().item() < 0.5 * noise
    close(out["loss_D"], o["loss_D"], 1e-2, "loss_D")
    close(out["loss_G"], o["loss_G"], 1e-2, "loss_G")
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    worst = 0.0
    for name, mine, theirs, shadow in (("D", gD, ref.last["grads_D"], O.shadowed_bias_keys("D")),
                                       ("G", gG, ref.last["grads_G"], O.shadowed_bias_keys("G", 6))):
        for k, v in theirs.items():
            if k not in shadow:
                a, b = mine[k].reshape(-1), v.reshape(-1)
                worst = max(worst, ((a - b).norm() / (b.norm() + 1e-20)).item())
    # gradients: sign(pred - nir) flips wherever |pred - nir| is inside the 5e-3 drift; a fraction f of flipped pixels moves
    # the L1 gradient by 2*sqrt(f) in relative L2 (f = 0.4 %% -> 13 %%).  The backward RULE is checked teacher-forced in
    # test_bf16_contraction_backward_rule; here only that nothing is grossly off.
    assert worst < 0.3, worst


def test_bf16_mode_stores_convolution_outputs_as_bf16(emu, golden_dir, monkeypatch):
    """bf16 operand mode, storage rule: a convolution output in front of an InstanceNorm whose launch leaves the statistics from its fp32
    accumulators (maps of >= OPT.epilogue_min_pixels_bf16 pixels, H*W % 128 == 0 per launch, no split-K) is kept as bf16; the norm's apply
    and both passes of its backward read the rounded tensor.  Thresholds lowered so that the golden net's 32 x 32 and 16 x 16 maps
    qualify.  (1) the stored tensor is the oracle's convolution rounded to bf16, the statistics are those of the UNROUNDED values;
    (2) the step -- with the data gradients that feed an instance-norm backward stored as bf16 too (OPT.bf16_g) -- stays inside the bf16
    noise band of the oracle's restatement of the same rules (operand_precision(y_bf16_min_pixels, g_bf16_min_tiles))."""
    import torch.nn.functional as F
    from nirgan_hip.options import OPT
    from nirgan_hip.trainer import Pix2PixTrainer
    monkeypatch.setattr(OPT, "epilogue_min_pixels", 64)
    monkeypatch.setattr(OPT, "epilogue_min_pixels_bf16", 64)
    monkeypatch.setattr(OPT, "bf16_store_min_tiles", 0)
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision="bf16")
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    out = tr.step(rgb, nir).as_dict()
    eng, pG = tr.G, sub(z, "G0/")
    stored = {n: getattr(eng, n).y.t.dtype for n in ("L1", "L2", "L3", "U1", "U2")}
    assert stored == {"L1": torch.bfloat16, "L2": torch.bfloat16, "L3": torch.float32, "U1": torch.float32, "U2": torch.bfloat16}, stored
    assert "in_fwd_pre" in emu.calls
    assert sum(a[0]._obj.g_bf16 for n, a in eng.bwd.ops if n == "nirgan_instnorm_bwd") >= 14      # block layers, both down layers, first up layer
    with O.operand_precision("bf16"):
        y = O._conv2d(F.pad(rgb, (3, 3, 3, 3), mode="reflect"), pG["model.1.weight"], pG["model.1.bias"])
    got = _nchw(eng.L1.y).float()
    assert ((got - y).abs() <= y.abs() * 2.0 ** -8 + 1e-6).all(), "stored y is not the convolution rounded to bf16"
    close(eng.L1.stats[0], y.mean((2, 3)), 1e-5, "mean of the unrounded values")
    close(eng.L1.stats[1], torch.rsqrt(y.var((2, 3), unbiased=False) + 1e-5), 1e-4, "rstd of the unrounded values")
    with O.operand_precision("bf16", y_bf16_min_pixels=64, g_bf16_min_tiles=0):
        ref = O.OracleTrainer(pG, sub(z, "D0/"), 6)
        o = ref.step(rgb, nir)
    pred32 = torch.from_numpy(z["pred"])
    noise = (ref.last["pred"] - pred32).abs().max().item()
    assert noise > 1e-3
    assert (tr.G.pred.reshape(-1) - ref.last["pred"].reshape(-1)).abs().max().item() < 0.5 * noise
    close(out["loss_D"], o["loss_D"], 1e-2, "loss_D")
    close(out["loss_G"], o["loss_G"], 1e-2, "loss_G")
    worst = 0.0
    for mine, theirs, shadow in ((tr.flatD.grad_views(), ref.last["grads_D"], O.shadowed_bias_keys("D")),
                                 (tr.flatG.grad_views(), ref.last["grads_G"], O.shadowed_bias_keys("G", 6))):
        for k, v in theirs.items():
            if k not in shadow:
                a, b = mine[k].reshape(-1), v.reshape(-1)
                worst = max(worst, ((a - b).norm() / (b.norm() + 1e-20)).item())
    assert worst < 0.3, worst


@pytest.mark.parametrize("case", ["down3x3_s2", "small_c8_n16", "n64_tail"])
def test_bf16_contraction_backward_rule(emu, case):
    """One convolution as the engines emit it (forward, split weight gradient, data gradient by correlation or sub-pixel
    phases) with precision='bf16', against the oracle's rule on the same x, w, dy: y = conv(bf(x), bf(w)),
    dx = conv^T(bf(dy), bf(w)), dw = corr(bf(x), bf(dy)).  Per-contraction there is no drift: 1e-5."""
    from conv_cases import CONV_CASES, build_conv_case
    from nirgan_hip.engine import Ctx, Halo
    c = [c for c in CONV_CASES if c[0] == case][0]
    _, B, H, W, Cin, Cout, k, s, p = c
    gen = torch.Generator().manual_seed(3)
    ctx = Ctx("cpu", "bf16")
    x = Halo(ctx, B, H, W, Cin, p)
    x.interior().copy_(torch.randn(B, H, W, Cin, generator=gen))          # zero halo = Conv2d(padding=p)
    w = torch.randn(Cout, Cin, k, k, generator=gen) * 0.05
    b = torch.randn(Cout, generator=gen)
    plan, plan2, y, dy, gw, gx = build_conv_case(ctx, x, w, b, c)
    plan.run()
    xt = x.interior().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    with O.operand_precision("bf16"):
        yo = O._conv2d(xt, wt, b, stride=s, padding=p)
    close(y.t.permute(0, 3, 1, 2), yo.detach(), 1e-5, "forward")
    dyt = torch.randn(yo.shape, generator=gen)
    dy.interior().copy_(dyt.permute(0, 2, 3, 1))
    plan2.run()
    with O.operand_precision("bf16"):
        dxo, dwo = torch.autograd.grad(yo, (xt, wt), dyt)
    close(gw, dwo, 1e-5, "weight gradient")
    close(gx.interior().permute(0, 3, 1, 2), dxo, 1e-5, "data gradient")


def test_bf16x3_split_mode_stays_at_fp32_parity(emu, golden_dir):
    """precision='bf16x3': fp32 operands as hi+mid bf16 terms, three products -> the fp32 golden vectors at 1e-3."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision="bf16x3")
    out = tr.step(torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])).as_dict()
    close(tr.G.pred, z["pred"], 1e-3, "pred")
    close(out["loss_D"], z["loss_D"], 1e-3, "loss_D")
    close(out["loss_G"], z["loss_G"], 1e-3, "loss_G")
    with pytest.raises(NotImplementedError):
        Pix2PixTrainer(netG, netD, n_blocks=6, precision="fp8").step(torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"]))


def test_mixed_resolution_buckets_share_one_trainer(emu, golden_dir):
    """configs[4]: each step draws one resolution bucket.  One trainer keeps an engine set per (B, H, W); the sequence of
    losses equals the oracle's on the same sequence of batches."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6)
    ref = O.OracleTrainer(sub(z, "G0/"), sub(z, "D0/"), 6)
    g = torch.Generator().manual_seed(5)
    batches = []
    for B, S in ((2, 32), (1, 48), (2, 32), (1, 48)):
        batches.append((0.02 + 0.58 * torch.rand(B, 3, S, S, generator=g), 0.05 + 0.75 * torch.rand(B, 1, S, S, generator=g)))
    for i, (rgb, nir) in enumerate(batches):
        out = tr.step(rgb, nir).as_dict()
        o = ref.step(rgb, nir)
        # Adam's first steps turn rounding-level differences of near-zero gradients into +-lr steps: the two
        # trajectories separate slowly (same effect between any two fp32 implementations)
        tol = 2e-4 if i < 2 else 2e-3
        close(out["loss_D"], o["loss_D"], tol, f"loss_D step {i}")
        close(out["loss_G"], o["loss_G"], tol, f"loss_G step {i}")
    assert len(tr._states) == 2


def test_calculate_metrics_surface(emu):
    """utils.calculate_metrics.calculate_metrics keeps the reference's signature/keys (utils/calculate_metrics.py:5-36)
    and agrees with the oracle's restatement; ssim_loss carries its gradient wrt the prediction."""
    from utils.calculate_metrics import calculate_metrics, image_metrics_device
    from utils.losses import emd_loss, ssim_loss
    g = torch.Generator().manual_seed(2)
    pred = torch.rand(2, 1, 40, 50, generator=g)
    target = (pred + 0.05 * torch.randn(2, 1, 40, 50, generator=g)).clamp(0, 1)
    got, ref = calculate_metrics(pred, target, phase="val"), O.calculate_metrics(pred, target, "val")
    assert list(got) == ["val/L1", "val/L2", "val/PSNR", "val/SSIM"]
    for k in ref:
        close(got[k], ref[k], 1e-5, k)
    close(ssim_loss(pred, target), 1.0 - O.ssim_map(pred, target, 11).mean(), 1e-5, "ssim_loss")
    p1, p2 = pred.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    (3.0 * ssim_loss(p1, target)).backward()
    (3.0 * O.ssim_loss(p2, target)).backward()
    close(p1.grad, p2.grad, 1e-5, "ssim_loss gradient")
    with pytest.raises(NotImplementedError):
        ssim_loss(pred, target.clone().requires_grad_(True))
    close(emd_loss(pred, target), O.emd_loss(pred, target), 5e-4, "emd_loss")      # float32 CDFs near 1.0 carry 6e-8, their differences are ~6e-4
    e1, e2 = pred.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    (5.0 * emd_loss(e1, target)).backward()
    (5.0 * O.emd_loss(e2.double(), target.double())).backward()
    close(e1.grad, e2.grad, 1e-4, "emd_loss gradient")
    with pytest.raises(NotImplementedError):
        emd_loss(pred, target.clone().requires_grad_(True))
    with pytest.raises(ValueError):
        image_metrics_device(pred, target[:, :, :-1])
    with pytest.raises(RuntimeError):
        image_metrics_device(pred[:, :, :2, :2], target[:, :, :2, :2])      # smaller than the window radius


def _locenc_checkpoint(z, calculation="analytic"):
    hp = {"le_type": "sphericalharmonics", "legendre_polys": 10, "harmonics_calculation": calculation, "min_radius": 1,
          "max_radius": 360, "frequency_num": 10, "pe_type": "siren", "embed_dim": 32, "capacity": 64, "num_hidden_layers": 2}
    sd = {"model.location." + k: v for k, v in sub(z, "siren/").items()}
    sd["model.visual.conv1.weight"] = torch.zeros(1)            # the rest of SatCLIP is ignored by the loader
    return {"hyper_parameters": hp, "state_dict": sd}


def test_satclip_location_encoder_surface(emu, golden_dir, tmp_path):
    """model.satclip.{location_encoder, load_lightweight, satclip_wrapper}: the reference's names, parameter keys and
    checkpoint format (load_lightweight.py:5-35, satclip_wrapper.py:8-35); harmonics against the reference's own
    closed-form values (fixture f6), the whole encoder against the oracle."""
    from model.satclip.location_encoder import (LocationEncoder, SphericalHarmonics, get_neural_network,
                                                get_positional_encoding)
    from model.satclip.load_lightweight import get_satclip_loc_encoder
    from model.satclip.satclip_wrapper import SatClIP_wrapper
    z = load(golden_dir, "f6_locenc.npz")
    lonlat = torch.from_numpy(z["lonlat"])
    for L_ in (10, 16):
        close(SphericalHarmonics(L_, "closed-form")(lonlat), z[f"Y{L_}"], 1e-12, f"harmonics L={L_}")
    # Fixture f7 = the reference's generator script (sympy) evaluated as its text reads ('analytic-generator-text': (-1)^m on m != 0
    # and -- an operator-precedence slip -- a factor pi on m == 0).  The default 'analytic' keeps the generator's signs and the
    # ORTHONORMAL zonal constant the published table shows (Yl0_m0 = 0.28209...): pinned by f7 on m != 0, by f6 on m == 0.
    z7 = load(golden_dir, "f7_sh_analytic.npz")
    assert np.array_equal(z7["lonlat"], z["lonlat"])
    yg = SphericalHarmonics(10, "analytic-generator-text")(lonlat)
    close(yg, z7["Y10"], 1e-12, "generator-text harmonics L=10")
    assert abs(float(yg[0, 0]) - 0.886226925452758) < 1e-14
    ya = SphericalHarmonics(10)(lonlat)
    zonal = torch.tensor([l * l + l for l in range(10)])
    other = torch.tensor([f for f in range(100) if f not in set(zonal.tolist())])
    close(ya[:, other], torch.from_numpy(z7["Y10"])[:, other], 1e-12, "analytic harmonics, m != 0 (f7)")
    close(ya[:, zonal], torch.from_numpy(z["Y10"])[:, zonal], 1e-12, "analytic harmonics, m == 0 (f6)")
    assert abs(float(ya[0, 0]) - 0.28209479177387814) < 1e-14 and (ya - torch.from_numpy(z["Y10"])).abs().max() > 0.5
    with pytest.raises(NotImplementedError):
        SphericalHarmonics(10, "discretized")
    net = get_neural_network("siren", 100, 32, 64, 2)
    assert sorted(net.state_dict()) == sorted(k[len("nnet."):] for k in sub(z, "siren/"))
    assert float(net.layers[0].weight.detach().abs().max()) <= 1 / 100 and net.layers[0].activation.w0 == 30.0
    path = tmp_path / "satclip.ckpt"
    torch.save(_locenc_checkpoint(z), path)
    enc = get_satclip_loc_encoder(str(path), "cpu")
    assert isinstance(enc, LocationEncoder) and not enc.training and enc.nnet.last_layer.weight.dtype == torch.float64
    out = enc(lonlat)
    assert out.dtype == torch.float64 and enc.posenc.harmonics_calculation == "analytic"
    close(out, O.location_encoder_forward(sub(z, "siren/"), lonlat, 10, 2, "analytic"), 1e-12, "encoder vs oracle (analytic)")
    path_cf = tmp_path / "satclip_cf.ckpt"
    torch.save(_locenc_checkpoint(z, "closed-form"), path_cf)
    close(get_satclip_loc_encoder(str(path_cf), "cpu")(lonlat), z["siren_out"], 1e-12, "encoder vs oracle (closed-form)")
    emb = SatClIP_wrapper(str(path), device="cpu").predict(lonlat.float())
    assert emb.dtype == torch.float32 and emb.shape == (lonlat.shape[0], 32) and not emb.requires_grad
    close(emb, O.location_encoder_forward(sub(z, "siren/"), lonlat.float().double(), 10, 2, "analytic").float(), 1e-6, "wrapper")
    with pytest.raises(NotImplementedError):
        get_positional_encoding("grid")
    with pytest.raises(NotImplementedError):
        get_neural_network("fcnet", 100)
    with pytest.raises(ValueError):
        get_positional_encoding("nope")
    with pytest.raises(NotImplementedError):
        enc.train()(lonlat)
    with pytest.raises(ValueError):
        enc.eval()(lonlat[:, :1])


def test_lightning_free_fit_loop(emu, tmp_path):
    """nirgan_hip.fit.fit: fused train batches, validation scalars, ReduceLROnPlateau on val/L1 driving both Adam steps,
    a checkpoint with the reference's state_dict keys (train.py:61-65 loads it with strict=False)."""
    from model.pix2pix import Px2Px_PL
    from nirgan_hip.fit import fit

    class A(dict):
        __getattr__ = dict.__getitem__

    def to_attr(d):
        return A({k: to_attr(v) if isinstance(v, dict) else v for k, v in d.items()})
    cfg = to_attr({
        "base_configs": {"isTrain": True, "input_nc": 3, "output_nc": 1, "ngf": 8, "ndf": 8, "netD": "basic",
                         "netG": "resnet_6blocks", "norm": "instance", "no_dropout": True, "init_type": "normal",
                         "init_gain": 0.02, "n_layers_D": 3, "gan_mode": "lsgan", "lr": 0.0002, "beta1": 0.5,
                         "lambda_GAN": 1.0, "lambda_L1": 100.0, "lambda_ssim": 0.0, "lambda_hist": 0.0,
                         "lambda_rs_losses": 0.0, "rs_losses_criterium": "l1",
                         "internal_rs_loss_weights": {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333}},
        "satclip": {"use_satclip": False},
        "Schedulers": {"metric": "val/L1", "patience_g": 0, "patience_d": 5},
        "custom_configs": {"Logging": {"num_val_images": 0}},
        "Data": {"padding": False, "padding_amount": 0}})
    torch.manual_seed(0)
    m = Px2Px_PL(cfg)
    g = torch.Generator().manual_seed(1)
    mk = lambda: {"rgb": 0.02 + 0.58 * torch.rand(2, 3, 32, 32, generator=g), "nir": 0.05 + 0.75 * torch.rand(2, 1, 32, 32, generator=g)}  # noqa: E731
    train, val = [mk(), mk()], [mk()]
    seen = []
    ckpt = tmp_path / "last.ckpt"
    # a validation metric that cannot improve: the monitor is patched to a constant through on_log's records
    hist = fit(m, train, val, max_epochs=3, log_every=1, on_log=seen.append, ckpt_path=str(ckpt))
    assert len(hist["train"]) == 6 and len(hist["val"]) == 3 and {"val/L1", "val/L2", "val/PSNR", "val/SSIM"} <= set(hist["val"][0])
    assert all(np.isfinite(r["loss_G"]) and np.isfinite(r["loss_D"]) for r in hist["train"])
    tr = m.fused_trainer()
    assert tr.steps == 6
    # lr bookkeeping: whatever the schedulers decided is what the fused Adam steps use
    (od, og), _ = m.configure_optimizers()
    assert hist["lr"][-1]["lr_g"] <= 2e-4 and hist["lr"][-1]["lr_d"] == 2e-4
    sd = torch.load(str(ckpt))["state_dict"]
    assert "netG.model.1.weight" in sd and "netD.model.11.bias" in sd
    m2 = Px2Px_PL(cfg)
    missing = m2.load_state_dict(sd, strict=False)
    assert not missing.missing_keys
    assert torch.equal(m2.netG.model[1].weight.detach(), m.netG.model[1].weight.detach().cpu())


def test_micro_batched_step_equals_the_whole_batch(emu, golden_dir):
    """micro_batches=2: the batch runs as two parts (separate HIP streams on the device) whose scaled gradients are summed
    before each Adam step -- same golden vectors as the single-part step (instance norm is per sample, losses are means)."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g9_rs_pad.npz")
    nb, pad, lam_rs = int(z["n_blocks"]), int(z["padding"]), float(z["lambda_rs"])
    netG, netD = make_nets(z, nb)
    rs_w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333}
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, lambda_rs=lam_rs, rs_weights=rs_w, padding=pad, micro_batches=2)
    out = tr.step(torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])).as_dict()
    assert tr._state.n == 2 and tr._state.micros[0].B == 1
    close(tr.pred, z["pred"], 2e-5, "pred")
    close(out["loss_D"], z["loss_D"], 1e-5, "loss_D")
    close(out["loss_G"], z["loss_G"], 1e-5, "loss_G")
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    for k, v in sub(z, "gD/").items():
        if k not in O.shadowed_bias_keys("D"):
            close(gD[k], v, 2e-4, "gD " + k)
    shadow = O.shadowed_bias_keys("G", nb)
    for k, v in sub(z, "gG/").items():
        if k not in shadow:
            close(gG[k], v, 2e-4, "gG " + k)
    # an odd batch falls back to fewer parts
    tr3 = Pix2PixTrainer(*make_nets(z, nb), n_blocks=nb, micro_batches=2)
    tr3.step(torch.rand(3, 3, 32, 32) * 0.5 + 0.1, torch.rand(3, 1, 32, 32) * 0.5 + 0.1)
    assert tr3._state.n == 1


def test_fused_trainer_sees_external_weight_changes(emu, golden_dir):
    """Weights changed behind the trainer's back (load_state_dict / in-place surgery between steps) are re-packed: the next
    step equals a fresh trainer's first step on the modified weights."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    netG, netD = make_nets(z, 6)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, lr=0.0)      # lr 0: the steps themselves leave the weights alone
    tr.step(rgb, nir)
    with torch.no_grad():
        netG.model[1].weight.mul_(1.5)
        sd = netD.state_dict()
        sd["model.11.weight"] = sd["model.11.weight"] * 0.5
        netD.load_state_dict(sd)
    after = tr.step(rgb, nir).as_dict()
    g2, d2 = make_nets(z, 6)
    g2.load_state_dict(netG.state_dict())
    d2.load_state_dict(netD.state_dict())
    fresh = Pix2PixTrainer(g2, d2, n_blocks=6, lr=0.0).step(rgb, nir).as_dict()
    for k in ("loss_D", "loss_G", "loss_G_l1"):
        close(after[k], fresh[k], 1e-6, k)
    assert abs(after["loss_G"] - float(z["loss_G"])) > 1e-3 * abs(float(z["loss_G"]))      # and it did change something


@pytest.mark.parametrize("variant", ["F(4x4,4x4)", "direct"])
def test_discriminator_winograd_layer_through_the_trainer(emu, monkeypatch, variant):
    """ndf = 32 makes the PatchGAN's stride-1 4x4 layer 128 -> 256 channels, wide enough for the Winograd paths (forward, data + weight
    gradient in the D step, data gradient alone in the G step): one fused step against the oracle's trainer, with the default
    F(4x4,4x4) (49 plane GEMMs, csrc/wino6.hip) and with the direct tiles (OPT.winograd = 'off')."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    from nirgan_hip.options import OPT
    monkeypatch.setattr(OPT, "winograd", "off" if variant == "direct" else "f6")
    torch.manual_seed(11)     # (seed 3 puts a ReLU pre-activation of the last block at 2e-7: the mask flips between two valid fp32 evaluations)
    netG = networks.define_G(3, 1, 8, "resnet_6blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, 32, "basic", 3, "instance", "normal", 0.02)
    G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()}
    rgb, nir = torch.rand(2, 3, 32, 32), torch.rand(2, 1, 32, 32)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, lr=0.0)      # lr 0: the generator step sees the same D as the oracle's (no Adam sign noise)
    out = tr.step(rgb, nir).as_dict()
    if variant == "direct":
        assert not any(c.startswith("wino6") for c in emu.calls)
    else:       # D2 forward + its data gradient, D1 forward + its data gradient: 4 GEMM launches; one weight gradient (D step)
        assert emu.calls.count("wino6_gemm") == 4 and emu.calls.count("wino6_fin") == 1 and emu.calls.count("wino6_dy") == 1
    ref = O.OracleTrainer(G0, D0, 6, lr=0.0)
    o = ref.step(rgb, nir)
    close(out["loss_D"], o["loss_D"], 1e-5, "loss_D")
    close(out["loss_G"], o["loss_G"], 1e-5, "loss_G")
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    for k, v in ref.last["grads_D"].items():
        if k not in O.shadowed_bias_keys("D"):
            close(gD[k], v, 2e-4, "gD " + k)
    for k, v in ref.last["grads_G"].items():
        if v is not None and k not in O.shadowed_bias_keys("G", 6):
            close(gG[k], v, 2e-4, "gG " + k)


def test_fused_trainer_with_the_ssim_term(emu, golden_dir):
    """lambda_ssim > 0 (model/pix2pix.py:233-237): the fused step adds lambda_ssim * (1 - mean SSIM_11(pred, nir)) to the generator
    objective and its gradient to dpred; against the oracle's trainer with the same term.  Also with two micro-batches."""
    from nirgan_hip.trainer import Pix2PixTrainer
    z = load(golden_dir, "f1_g6_d.npz")
    rgb, nir = torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])
    ref = O.OracleTrainer(sub(z, "G0/"), sub(z, "D0/"), 6, lr=0.0, lambda_ssim=40.0)
    o = ref.step(rgb, nir)
    assert abs(float(o["loss_G"]) - float(z["loss_G"])) > 1.0                  # the term is not negligible here
    for micro in (1, 2):
        netG, netD = make_nets(z, 6)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6, lr=0.0, lambda_ssim=40.0, micro_batches=micro)
        out = tr.step(rgb, nir).as_dict()
        assert "ssim_loss" in emu.calls
        close(out["loss_G"], o["loss_G"], 1e-5, "loss_G")
        close(out["loss_G_ssim"], o["loss_G_ssim"], 1e-5, "loss_G_ssim")
        gG = tr.flatG.grad_views()
        for k, v in ref.last["grads_G"].items():
            if v is not None and k not in O.shadowed_bias_keys("G", 6):
                close(gG[k], v, 2e-4, f"gG {k} (micro {micro})")


@pytest.mark.parametrize("variant", ["F(6x6,3x3)", "F(4x4,3x3)", "direct"])
def test_generator_winograd_layers_through_the_trainer(emu, monkeypatch, variant):
    """ngf = 32 makes the residual-block convolutions 128 -> 128 channels, wide enough for the Winograd paths: forward with the
    instance-norm apply of each block's first convolution folded into the second one's input transform, data gradient and
    transform-domain weight gradient with one transform pass over dY.  One fused step against the oracle, with the default
    F(6x6,3x3) path (csrc/wino6.hip: 64 plane GEMMs + separate output transform), with F(4x4,3x3) (OPT.winograd = "f4") and with
    the direct tiles (OPT.winograd = "off")."""
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    from nirgan_hip.options import OPT
    monkeypatch.setattr(OPT, "winograd", {"F(6x6,3x3)": "f6", "F(4x4,3x3)": "f4", "direct": "off"}[variant])
    torch.manual_seed(5)
    netG = networks.define_G(3, 1, 32, "resnet_6blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()}
    rgb, nir = torch.rand(2, 3, 32, 32), torch.rand(2, 1, 32, 32)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, lr=0.0)
    out = tr.step(rgb, nir).as_dict()
    if variant == "direct":
        assert not any(c.startswith("wino6") for c in emu.calls)
    else:
        # 12 forward + 12 data-gradient GEMM launches, 6 normalising input transforms, 12 weight-gradient finishes, V kept by the forward
        assert emu.calls.count("wino6_in_norm") == 6 and emu.calls.count("wino6_gemm") == 24 and emu.calls.count("wino6_out") == 24
        assert emu.calls.count("wino6_fin") == 12 and emu.calls.count("wino6_dy") == 12
        assert emu.calls.count("wino6_in") == 6 + 12          # c1 forwards + the dY transforms (no re-transform of the forward input)
        assert emu.calls.count("in_fwd_pre") == 12             # instance-norm statistics from the output transforms' partial sums
        if variant == "F(6x6,3x3)":
            # the first pass of the instance-norm backward inside the data gradient's output transform: the 12 layers whose gradient
            # arrives from a Winograd data gradient (11 block convolutions + the last stride-2 layer); 8x8 maps padded to 10x10 put
            # the far halo line and its fold partner into one 6x6 tile (with 4x4 tiles they straddle two: not fused)
            assert emu.calls.count("wino6_out_inbwd") == 12 and emu.calls.count("in_bwd_pre") == 12
            assert emu.calls.count("wino6_dy_norm") == 12      # the instance-norm backward's second pass inside the dY transform
        else:
            assert "wino6_out_inbwd" not in emu.calls
    ref = O.OracleTrainer(G0, D0, 6, lr=0.0)
    o = ref.step(rgb, nir)
    close(tr.G.pred, ref.last["pred"], 2e-5, "pred")
    close(out["loss_G"], o["loss_G"], 1e-5, "loss_G")
    gG = tr.flatG.grad_views()
    for k, v in ref.last["grads_G"].items():
        if v is not None and k not in O.shadowed_bias_keys("G", 6):
            close(gG[k], v, 2e-4, "gG " + k)


def test_direct_last_layer_restatement_matches_conv2d(emu):
    """The numpy emulation of nirgan_endconv_* (what the emulated trainer tests run the last layer on when ngf = 64) against
    torch's conv2d + autograd."""
    from endconv_case import run_endconv
    run_endconv("cpu", 2, 9, 10, 1, tol=1e-6)
    run_endconv("cpu", 1, 7, 5, 0, tol=1e-6, act=L.ACT_NONE)


def test_conv_epilogue_statistics_through_the_trainer(emu, monkeypatch, golden_dir):
    """The direct-tile layers leave the instance norm's partial sums in the convolution's epilogue (the engines do that from 16 K
    pixels per sample; OPT.epilogue_min_pixels = 0 here): one fused step on the golden small nets, same bounds as without."""
    from nirgan_hip.trainer import Pix2PixTrainer
    from nirgan_hip.options import OPT
    monkeypatch.setattr(OPT, "epilogue_min_pixels", 0)
    z = load(golden_dir, "f1_g9_rs_pad.npz")
    nb = int(z["n_blocks"])
    netG, netD = make_nets(z, nb)
    rs_w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333, "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, lambda_rs=float(z["lambda_rs"]), rs_weights=rs_w, padding=int(z["padding"]))
    out = tr.step(torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])).as_dict()
    assert emu.calls.count("conv_stats") >= 10 and emu.calls.count("in_fwd_pre") >= 10
    close(tr.G.pred, z["pred"], 1e-4, "pred")
    for k in ("loss_D", "loss_G", "loss_G_l1"):
        close(out[k], z[k], 1e-4, k)


def test_conv_epilogue_takes_the_instance_norm_backward_first_pass(emu, monkeypatch, golden_dir):
    """The sub-pixel phase launches of a stride-2 data gradient leave the consumer layer's first backward pass (sums of g_z and g_z z per
    128-pixel tile, nirgan_conv_desc.fuse_*; the engines do that from 16 K pixels per sample, OPT.epilogue_min_pixels = 0 here, forward
    statistics kept on their own pass so that the forward is the golden one): gradients of the golden small net within the usual bounds."""
    from nirgan_hip.trainer import Pix2PixTrainer
    from nirgan_hip.options import OPT
    monkeypatch.setattr(OPT, "epilogue_min_pixels", 0)
    monkeypatch.setattr(OPT, "epilogue_stats", False)
    z = load(golden_dir, "f1_g6_d.npz")
    nb = int(z["n_blocks"])
    netG, netD = make_nets(z, nb)
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, padding=int(z["padding"]))
    out = tr.step(torch.from_numpy(z["rgb"]), torch.from_numpy(z["nir"])).as_dict()
    assert emu.calls.count("conv_inbwd") >= 5 and emu.calls.count("in_bwd_pre") >= 2, (emu.calls.count("conv_inbwd"), emu.calls.count("in_bwd_pre"))
    close(tr.G.pred, z["pred"], 2e-5, "pred")
    close(out["loss_G"], z["loss_G"], 1e-5, "loss_G")
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    for k, v in sub(z, "gD/").items():
        if k not in O.shadowed_bias_keys("D"):
            close(gD[k], v, 2e-4, "gD " + k)
    shadow = O.shadowed_bias_keys("G", nb)
    for k, v in sub(z, "gG/").items():
        if k not in shadow:
            close(gG[k], v, 2e-4, "gG " + k)


def test_three_term_split_planes_follow_the_weights(emu, monkeypatch):
    """Descriptor precision 3 (csrc/igemm_x3.h) through the engine: in exact-fp32 mode the 32-channel-run convolutions / weight gradients
    and the Winograd plane GEMMs carry the bf16 planes of their packed / transformed weights.  The emulator refuses a launch whose planes
    are not the three-term split of the weights it is given, so two optimizer steps at lr > 0 prove the planes are refreshed with the
    weights; OPT.split3 = False emits none of it; the first step's results stay at the fp32 oracle's."""
    from emu_backend import obj
    from model import networks
    from nirgan_hip.options import OPT
    from nirgan_hip.trainer import Pix2PixTrainer

    def run(split3, steps):
        monkeypatch.setattr(OPT, "split3", split3)
        torch.manual_seed(5)
        netG = networks.define_G(3, 1, 32, "resnet_6blocks", "instance", False, "normal", 0.02)
        netD = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
        G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
        D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()}
        g = torch.Generator().manual_seed(6)
        rgb, nir = torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 1, 32, 32, generator=g)
        seen = {"conv": [], "wgrad": [], "gemm": []}
        for name, key, probe in (("nirgan_conv_igemm", "conv", lambda d: (d.precision, bool(d.w_x3))),
                                 ("nirgan_wgrad_igemm", "wgrad", lambda d: (d.precision, d.nplanes)),
                                 ("nirgan_wino6_gemm", "gemm", lambda d: bool(d.U3))):
            orig = getattr(emu, name)

            def spy(ref, stream=None, _orig=orig, _key=key, _probe=probe):
                seen[_key].append(_probe(obj(ref)))
                return _orig(ref, stream)
            monkeypatch.setattr(emu, name, spy)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6)
        outs = [tr.step(rgb, nir).as_dict() for _ in range(steps)]
        return outs, seen, (G0, D0, rgb, nir), tr

    outs, seen, (G0, D0, rgb, nir), tr = run(True, 2)
    assert any(p == 3 and has for p, has in seen["conv"]), "no convolution took the split tile"
    assert all(has for p, has in seen["conv"] if p == 3)
    assert any(p == 3 and n == 1 for p, n in seen["wgrad"]) and any(p == 3 and n > 1 for p, n in seen["wgrad"]), seen["wgrad"]
    assert seen["gemm"] and all(seen["gemm"])
    assert emu.calls.count("split3") >= 2, "the planes are written with every pack"
    ref = O.OracleTrainer(G0, D0, 6)
    o = ref.step(rgb, nir)
    close(outs[0]["loss_G"], o["loss_G"], 1e-5, "loss_G, first step")
    o2 = ref.step(rgb, nir)
    close(outs[1]["loss_G"], o2["loss_G"], 2e-3, "loss_G, second step (after both Adam updates)")
    n_split3 = emu.calls.count("split3")
    outs_off, seen_off, _, _ = run(False, 1)
    assert not any(p == 3 for p, _ in seen_off["conv"]) and not any(p == 3 for p, _ in seen_off["wgrad"]) and not any(seen_off["gemm"])
    assert emu.calls.count("split3") == n_split3, "OPT.split3 = False still writes planes"
    close(outs_off[0]["loss_G"], outs[0]["loss_G"], 1e-5, "split tiles on / off")


def test_paired_sub_pixel_phases_keep_the_step(emu, monkeypatch):
    """nirgan_conv_desc.out_span = 2 through the engine (ngf = 64: ConvTranspose2d(128, 64, 3, s2) forward and the data gradient of
    Conv2d(64, 128, 3, s2) as two problems of 128 columns over the union of two phases' taps, with the instance-norm partial sums and the
    fused first backward pass in per-channel records): same losses and gradients as the four-phase launches (OPT.pair_phases = False)
    and as the oracle."""
    from emu_backend import obj
    from model import networks
    from nirgan_hip.options import OPT
    from nirgan_hip.trainer import Pix2PixTrainer

    def run(pair):
        monkeypatch.setattr(OPT, "pair_phases", pair)
        monkeypatch.setattr(OPT, "pair_pixels", pair)        # (the first convolution with two output pixels per GEMM row: the same descriptor field)
        monkeypatch.setattr(OPT, "epilogue_min_pixels", 0)
        torch.manual_seed(11)
        netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02)
        netD = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
        G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
        D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()}
        g = torch.Generator().manual_seed(12)
        rgb, nir = torch.rand(1, 3, 64, 64, generator=g), torch.rand(1, 1, 64, 64, generator=g)      # (64 wide: 32 pixel pairs per row, the weight gradient's K-tile)
        seen = []
        orig = emu.nirgan_conv_igemm_group

        def spy(descs, n, stream=None):
            seen.append([(descs[i].contents.out_span, descs[i].contents.N, descs[i].contents.ntaps, bool(descs[i].contents.stats_ws),
                          bool(descs[i].contents.fuse_y), descs[i].contents.precision) for i in range(n)])
            return orig(descs, n, stream)
        monkeypatch.setattr(emu, "nirgan_conv_igemm_group", spy)
        orig1 = emu.nirgan_conv_igemm

        def spy1(ref, stream=None):
            d = obj(ref)
            if d.out_span == 2 and d.out_cs == d.N:          # (the group launches come through here too: theirs have out_cs == N / 2)
                seen.append([("first", d.N, d.ntaps, d.run, d.in_cs, d.out_cs, bool(d.stats_ws), d.precision)])
            return orig1(ref, stream)
        monkeypatch.setattr(emu, "nirgan_conv_igemm", spy1)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6)
        out = tr.step(rgb, nir).as_dict()
        return out, tr.flatG.grad.clone(), tr.flatD.grad.clone(), tr.pred.clone(), seen, (G0, D0, rgb, nir)

    out, gG, gD, pred, seen, (G0, D0, rgb, nir) = run(True)
    assert emu.calls.count("reduce_part") == 2, "the first convolution's weight gradient folds its two pixel-parity bands with two calls"
    firsts = [grp[0] for grp in seen if grp[0][0] == "first"]
    assert firsts and all(f == ("first", 128, 7, 32, 8, 128, True, 3) for f in firsts), firsts      # Conv2d(3, 64, 7): 128 columns, 7 taps of 32, pixel-pair views, statistics
    seen[:] = [grp for grp in seen if grp[0][0] != "first"]
    pairs = [grp for grp in seen if any(s == 2 for s, *_ in grp)]
    assert len(pairs) >= 2, seen
    for grp in pairs:
        assert [(s, n, t, p) for s, n, t, _, _, p in grp] == [(2, 128, 2, 3), (2, 128, 4, 3)], grp
    assert any(all(st for _, _, _, st, _, _ in grp) for grp in pairs), "the forward pair leaves the instance-norm partial sums"
    assert any(all(fy for _, _, _, _, fy, _ in grp) for grp in pairs), "the data-gradient pair takes the consumer's first backward pass"
    out4, gG4, gD4, pred4, seen4, _ = run(False)
    assert not any(grp[0][0] == "first" or any(s == 2 for s, *_ in grp) for grp in seen4)
    close(pred, pred4, 1e-5, "prediction, paired against four phases")
    for k in out:
        close(out[k], out4[k], 1e-5, k)
    assert ((gG - gG4).norm() / gG4.norm()).item() < 1e-5 and ((gD - gD4).norm() / gD4.norm()).item() < 1e-5
    o = O.OracleTrainer(G0, D0, 6).step(rgb, nir)
    close(out["loss_G"], o["loss_G"], 1e-5, "loss_G against the oracle")
