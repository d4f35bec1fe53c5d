"""Bodies of the API-level parity tests, shared by the CPU suite (C ABI served by the numpy emulator:
tests/test_api_emulated.py) and the MI355X suite (the HIP library: tests/test_gpu_api.py).

They drive the classes the reference's scripts drive -- ``Px2Px_PL`` exactly as Lightning 1.9's two-optimizer loop
does (training_step(batch, i, 0) -> backward -> optimizer_D.step -> toggle -> training_step(batch, i, 1) -> backward ->
optimizer_G.step; model/pix2pix.py:165-257, :485-492), ``Pix2PixModel.optimize_parameters``
(model/pix2pix_model.py:100-154), the Lightning-free ``fit`` loop, tiled inference and checkpoint loading
(create_synthetic_dataset.py:21-28,100-118) -- and compare with the golden vectors of the reference itself:
losses, every non-shadowed gradient and the parameters after the reference's ``torch.optim.Adam`` steps.
"""
import os

import numpy as np
import pytest
import torch

import nirgan_oracle as O

RS_W = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333, "lambda_savi": 0.0, "lambda_msavi": 0.0,
        "lambda_gndvi": 0.0}


class Tol:
    """out: outputs / losses (max error relative to the reference's max); grad: (rel-L2, max-rel) per gradient tensor."""

    def __init__(self, out, grad_l2, grad_max):
        self.out, self.grad_l2, self.grad_max = out, grad_l2, grad_max


CPU_TOL = Tol(2e-5, 1e-4, 2e-4)       # numpy emulator: same arithmetic as the oracle up to summation order
GPU_TOL = Tol(1e-3, 1e-3, 1e-2)       # BASELINE.json: 1e-3 relative fp32 (measured 1e-6..1e-5)


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


def grad_close(a, b, tol: Tol, what):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape and torch.isfinite(a).all(), what
    nrm, mx = b.norm().item(), b.abs().max().item()
    e2, em = (a - b).norm().item(), (a - b).abs().max().item()
    assert e2 <= tol.grad_l2 * max(nrm, 1e-20), f"{what}: rel L2 {e2 / max(nrm, 1e-20):.3e}"
    assert em <= tol.grad_max * max(mx, 1e-20), f"{what}: max err {em:.3e} of {mx:.3e}"


def adam_close(p_new, p_ref, g_ref, what, lr=2e-4):
    """Parameters after the FIRST Adam step against the reference's torch.optim.Adam result.  The first step is
    p - lr * g / (|g| + eps'): +-lr wherever |g| >> 1e-8, so the comparison is tight (1e-5 of max|p|, i.e. ~2 % of one
    step) on every element whose reference gradient has a definite sign, and bounded by one step elsewhere (an element
    whose gradient is rounding noise may step the other way in two correct fp32 evaluations)."""
    p_new, p_ref, g_ref = (torch.as_tensor(t).detach().float().cpu() for t in (p_new, p_ref, g_ref))
    assert p_new.shape == p_ref.shape and torch.isfinite(p_new).all(), what
    live = g_ref.abs() > 1e-4 * g_ref.abs().max().clamp_min(1e-30)
    zero = g_ref == 0                                     # e.g. fc rows no resized pixel reads: parameter untouched
    d = (p_new - p_ref).abs()
    assert live.sum().item() >= 0.9 * (~zero).sum().item() or g_ref.numel() < 16, f"{what}: gradient mostly noise?"
    tight = max(1e-5 * p_ref.abs().max().item(), 0.02 * lr)
    assert (d * live).max().item() <= tight, f"{what}: {(d * live).max().item():.3e} > {tight:.3e}"
    assert (d * zero).max().item() <= 1e-7, f"{what}: a parameter with zero gradient moved"
    assert d.max().item() <= 2.0 * lr * 1.001 + 1e-7, f"{what}: moved by more than a step: {d.max().item():.3e}"


def regen_fc(z):
    g = torch.Generator().manual_seed(int(z["fc_seed"]))
    return torch.randn(16384, 256, generator=g) * 0.02, torch.randn(16384, generator=g) * 0.02


def px_config(n_blocks=6, ngf=8, padding=0, lambda_rs=0.0, inject=False, patience_g=25, patience_d=25, lambda_ssim=0.0):
    """The keys of configs/config_px2px.yaml / config_px2px_SatCLIP.yaml that the hot path reads, small widths."""
    from utils.config import to_attr
    return to_attr({
        "base_configs": {"isTrain": True, "input_nc": 3, "output_nc": 1, "ngf": ngf, "ndf": ngf, "netD": "basic",
                         "netG": f"resnet_{n_blocks}blocks", "norm": "instance", "no_dropout": True, "init_type": "normal",
                         "init_gain": 0.02, "n_layers_D": 3, "gan_mode": "lsgan", "lr": 0.0002, "beta1": 0.5, "direction": "AtoB",
                         "lambda_GAN": 1.0, "lambda_L1": 100.0, "lambda_ssim": lambda_ssim, "lambda_hist": 0.0,
                         "lambda_rs_losses": lambda_rs, "rs_losses_criterium": "l1", "internal_rs_loss_weights": dict(RS_W)},
        "satclip": {"use_satclip": bool(inject), "satclip_style": "inject", "satclip_inject_style": "multiply",
                    "scaling_param": True, "scaling_param_init": 0.01, "post_correction": False, "post_correction_init": 1.0},
        "Schedulers": {"metric": "val/L1", "patience_g": patience_g, "patience_d": patience_d},
        "custom_configs": {"Logging": {"num_val_images": 0, "log_input_stats": False}},
        "Data": {"padding": padding > 0, "padding_amount": padding}})


def _load_golden_weights(m, z, inject):
    sd = {"netG." + k: v for k, v in sub(z, "G0/").items()}
    if inject:
        sd["netG.fc.weight"], sd["netG.fc.bias"] = regen_fc(z)
    sd.update({"netD." + k: v for k, v in sub(z, "D0/").items()})
    res = m.load_state_dict(sd, strict=False)
    assert not [k for k in res.missing_keys if k.startswith(("netG.", "netD."))], res.missing_keys
    assert not res.unexpected_keys


# ------------------------------------------------------------------------------------------------ a9: Px2Px_PL
def px2px_pl_lightning_sequence(dev, golden_dir, name, tol: Tol):
    from model.pix2pix import Px2Px_PL
    z = load(golden_dir, name)
    inject = "embeds" in z
    nb = 9 if inject else int(z["n_blocks"])
    pad = 0 if inject else int(z["padding"])
    lam_rs = 0.0 if inject else float(z["lambda_rs"])
    m = Px2Px_PL(px_config(nb, 8, pad, lam_rs, inject))
    _load_golden_weights(m, z, inject)
    m = m.to(dev)
    batch = {"rgb": torch.from_numpy(z["rgb"]).to(dev), "nir": torch.from_numpy(z["nir"]).to(dev)}
    if inject:
        batch["coords"] = torch.from_numpy(z["embeds"]).to(dev)      # B x 256: precomputed SatCLIP embeddings (pix2pix.py:509-526)
    shadowG, shadowD = O.shadowed_bias_keys("G", nb), O.shadowed_bias_keys("D")

    # predict_step: eval mode only (pix2pix.py:135), pad -> G -> crop
    with pytest.raises(AssertionError):
        m.train().predict_step(batch["rgb"])
    m.eval()
    p = m.predict_step(batch["rgb"], batch["coords"]) if inject else m.predict_step(batch["rgb"])
    assert p.shape == z["pred"].shape and not p.requires_grad
    close(p, z["pred"], tol.out, "predict_step")
    with pytest.raises(AssertionError):
        m.training_step(batch, 0, 0)
    m.train()

    (opt_d, opt_g), scheds = m.configure_optimizers()                # [D, G]: optimizer_idx 0 = D (pix2pix.py:490)
    assert opt_d.net is m.netD and opt_g.net is m.netG and len(scheds) == 2
    # ---- optimizer_idx 0
    loss_d = m.training_step(batch, 0, 0)
    close(loss_d, z["loss_D"], tol.out, "loss_D")
    if "loss_D_fake" in z:
        close(m.logged["model_loss/discriminator_fake"], z["loss_D_fake"], tol.out, "loss_D_fake")
        close(m.logged["model_loss/discriminator_real"], z["loss_D_real"], tol.out, "loss_D_real")
    opt_d.zero_grad()
    loss_d.backward()
    assert all(q.grad is None for q in m.netG.parameters()), "the D step must not reach the generator (fake_AB.detach())"
    for k, q in m.netD.named_parameters():
        if k not in shadowD and "gD/" + k in z:
            grad_close(q.grad, z["gD/" + k], tol, "gD " + k)
    opt_d.step()
    if "D1/model.0.weight" in z:
        for k, q in m.netD.named_parameters():
            if k not in shadowD:
                adam_close(q, z["D1/" + k], z["gD/" + k] if "gD/" + k in z else q.grad, "D1 " + k)
    # ---- optimizer_idx 1 (Lightning's toggle_optimizer: the other optimizer's parameters are frozen)
    for q in m.netD.parameters():
        q.requires_grad_(False)
    loss_g = m.training_step(batch, 0, 1)
    close(loss_g, z["loss_G"], tol.out, "loss_G")
    close(m.logged["model_loss/generator_GAN_loss"], z["loss_G_gan"], tol.out, "loss_G_gan")
    close(m.logged["model_loss/generator_L1"], z["loss_G_l1"], tol.out, "loss_G_l1")
    if lam_rs > 0:
        close(m.logged["model_loss/indices_loss_weighted"], z["loss_G_rs"], tol.out, "loss_G_rs")
    opt_g.zero_grad()
    loss_g.backward()
    gG = sub(z, "gG/")
    if inject:
        gG["scale_param"], gG["fc.bias"] = torch.from_numpy(z["g_scale_param"]), torch.from_numpy(z["g_fc_bias"])
    for k, q in m.netG.named_parameters():
        if k in shadowG:
            continue
        if k == "fc.weight":
            grad_close(q.grad[:8], z["g_fc_weight_rows0_8"], tol, "gG fc.weight[:8]")
            s = float(q.grad.double().sum())
            assert abs(s - float(z["g_fc_weight_sum"])) <= 1e-3 * float(z["g_fc_weight_abs"]) * tol.grad_max
        else:
            grad_close(q.grad, gG[k], tol, "gG " + k)
    opt_g.step()
    for q in m.netD.parameters():
        q.requires_grad_(True)
    if "G1/model.1.weight" in z:
        for k, q in m.netG.named_parameters():
            if k in shadowG:
                continue
            if k == "fc.weight":
                adam_close(q[:8], z["G1_fc_weight_rows0_8"], z["g_fc_weight_rows0_8"], "G1 fc.weight[:8]")
            elif k == "fc.bias":
                adam_close(q, z["G1_fc_bias"], z["g_fc_bias"], "G1 fc.bias")
            else:
                adam_close(q, z["G1/" + k], gG[k], "G1 " + k)
    # the optimizers' state is torch.optim.Adam's: one step taken, moments = (1-b) * g  (and it round-trips)
    sd = opt_g.state_dict()
    assert len(sd["state"]) == len(list(m.netG.parameters())) and float(sd["state"][0]["step"]) == 1.0
    names = [n for n, _ in m.netG.named_parameters()]
    i = names.index("model.1.weight")
    close(sd["state"][i]["exp_avg"], 0.5 * m.netG.model[1].weight.grad, 1e-6, "exp_avg")
    return m, batch, z


def px2px_pl_train_batch(dev, golden_dir, name, tol: Tol):
    """The fused path of the same class: train_batch = both optimizer passes in one call; same golden vectors."""
    from model.pix2pix import Px2Px_PL
    z = load(golden_dir, name)
    nb, pad, lam_rs = int(z["n_blocks"]), int(z["padding"]), float(z["lambda_rs"])
    m = Px2Px_PL(px_config(nb, 8, pad, lam_rs))
    _load_golden_weights(m, z, False)
    m = m.to(dev).train()
    batch = {"rgb": torch.from_numpy(z["rgb"]).to(dev), "nir": torch.from_numpy(z["nir"]).to(dev)}
    out = m.train_batch(batch).as_dict()
    for k in ("loss_D", "loss_G", "loss_G_gan", "loss_G_l1"):
        close(out[k], z[k], tol.out, k)
    tr = m.fused_trainer()
    close(tr.pred, z["pred"], tol.out, "pred")
    gD, gG = tr.flatD.grad_views(), tr.flatG.grad_views()
    for k, q in m.netD.named_parameters():
        if k not in O.shadowed_bias_keys("D"):
            grad_close(gD[k], z["gD/" + k], tol, "gD " + k)
            adam_close(q, z["D1/" + k], z["gD/" + k], "D1 " + k)
    for k, q in m.netG.named_parameters():
        if k not in O.shadowed_bias_keys("G", nb):
            grad_close(gG[k], z["gG/" + k], tol, "gG " + k)
            adam_close(q, z["G1/" + k], z["gG/" + k], "G1 " + k)
    # HipAdam built afterwards sees the fused steps' state (shared flat moments)
    (opt_d, opt_g), _ = m.configure_optimizers()
    assert float(opt_d.state_dict()["state"][0]["step"]) == 1.0 and float(opt_g.state_dict()["state"][0]["step"]) == 1.0


def lightning_toggled_sequence_reuses_the_forward(dev, golden_dir, tol: Tol):
    """Lightning 1.9's toggle_optimizer freezes the OTHER optimizer's parameters around each training_step: the D pass then runs the
    generator with nothing to differentiate, the G pass on the same batch tensors must still get its gradients -- from the first
    pass's activations (functional._ForwardRecord), not from a second forward.  HipAdam reads the bridge's gradient buffer in place."""
    from model.pix2pix import Px2Px_PL
    from nirgan_hip import lib as L_
    z = load(golden_dir, "f1_g6_d.npz")
    m = Px2Px_PL(px_config(6, 8))
    _load_golden_weights(m, z, False)
    m = m.to(dev).train()
    batch = {"rgb": torch.from_numpy(z["rgb"]).to(dev), "nir": torch.from_numpy(z["nir"]).to(dev)}
    (opt_d, opt_g), _ = m.configure_optimizers()
    shadowG, shadowD = O.shadowed_bias_keys("G", 6), O.shadowed_bias_keys("D")
    fwd_runs = [0]
    for q in m.netG.parameters():                      # toggle_optimizer(opt_d)
        q.requires_grad_(False)
    loss_d = m.training_step(batch, 0, 0)
    eng = m.netG.__dict__["_fwd_record"].lease.eng
    orig = eng.fwd.run
    eng.fwd.run = lambda: (fwd_runs.__setitem__(0, fwd_runs[0] + 1), orig())[1]
    close(loss_d, z["loss_D"], tol.out, "loss_D")
    opt_d.zero_grad()
    loss_d.backward()
    for k, q in m.netD.named_parameters():
        if k not in shadowD:
            grad_close(q.grad, z["gD/" + k], tol, "gD " + k)
    assert opt_d._aliased_flat_gradient(m.netD._flat(), [q.grad for q in m.netD.parameters()]) is not None, \
        "the D gradients should be views of one flat-layout buffer"
    opt_d.step()
    for q in m.netG.parameters():                      # untoggle; toggle_optimizer(opt_g)
        q.requires_grad_(True)
    for q in m.netD.parameters():
        q.requires_grad_(False)
    loss_g = m.training_step(batch, 0, 1)
    assert fwd_runs[0] == 0 and m.netG.__dict__.get("_fwd_reused", 0) == 1, "the generator forward ran twice for one batch"
    close(loss_g, z["loss_G"], tol.out, "loss_G")
    opt_g.zero_grad()
    loss_g.backward()
    for k, q in m.netG.named_parameters():
        if k not in shadowG:
            grad_close(q.grad, z["gG/" + k], tol, "gG " + k)
    assert opt_g._aliased_flat_gradient(m.netG._flat(), [q.grad for q in m.netG.parameters()]) is not None
    opt_g.step()
    for q in m.netD.parameters():
        q.requires_grad_(True)
    for k, q in m.netG.named_parameters():
        if k not in shadowG:
            adam_close(q, z["G1/" + k], z["gG/" + k], "G1 " + k)
    for k, q in m.netD.named_parameters():
        if k not in shadowD:
            adam_close(q, z["D1/" + k], z["gD/" + k], "D1 " + k)
    # a NEW batch object (or an in-place edit of the old one) is a different forward
    batch2 = {"rgb": batch["rgb"].clone(), "nir": batch["nir"]}
    m.training_step(batch2, 1, 0)
    assert fwd_runs[0] == 1
    batch2["rgb"].mul_(1.0)
    m.training_step(batch2, 1, 1)
    assert fwd_runs[0] == 2 and m.netG.__dict__.get("_fwd_reused", 0) == 1


def two_generator_graphs_on_one_input(dev, golden_dir, tol: Tol):
    """netG(x) twice on the SAME tensor with gradients wanted both times (advisor, round 3): each call owns its engine lease, both
    backwards run and give the same parameter gradients (the reference's nn.Module semantics); a second backward through ONE graph
    fails with a clear RuntimeError instead of an AttributeError on a released lease."""
    from model.pix2pix import Px2Px_PL
    z = load(golden_dir, "f1_g6_d.npz")
    m = Px2Px_PL(px_config(6, 8))
    _load_golden_weights(m, z, False)
    m = m.to(dev).train()
    rgb = torch.from_numpy(z["rgb"]).to(dev)
    p1 = m.netG(rgb)
    p2 = m.netG(rgb)
    assert m.netG.__dict__.get("_fwd_reused", 0) == 0, "a lease bound to one autograd node was handed to a second forward"
    assert torch.equal(p1, p2)
    w = torch.linspace(-1.0, 1.0, p1.numel(), device=dev).reshape(p1.shape)
    (p1 * w).sum().backward()
    g1 = [q.grad.clone() for q in m.netG.parameters()]
    for q in m.netG.parameters():
        q.grad = None
    (p2 * w).sum().backward()
    for a, q in zip(g1, m.netG.parameters()):
        assert torch.equal(a, q.grad), "the two graphs of one input differ"
    p3 = m.netG(rgb)
    loss = (p3 * w).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already consumed"):
        loss.backward()


def reference_config_key_set(dev, golden_dir, tol: Tol, full_width: bool):
    """BASELINE.json configs[0]: Px2Px_PL built from the reference's OWN config key set -- tests/golden/f0_config.json holds the parsed
    trees of configs/config_px2px.yaml and configs/config_px2px_SatCLIP.yaml (train.py:32-48) and the reference's state_dict keys and
    shapes for each (generated by oracle/make_golden.py::f0 from the reference's modules).  Checked: the key list and every shape for
    the YAML's netG and for resnet_6blocks (which the reference refuses with SatCLIP: so does this build); then one batch of 4 tiles
    through both optimizer passes with the YAML's Data.padding_amount = 10: finite losses, both networks stepped once.
    full_width = False (the CPU run, C ABI served by the numpy emulator): the same trees with ngf = ndf = 8 on 64 x 64 tiles."""
    import json
    from model.pix2pix import Px2Px_PL
    from utils.config import to_attr
    with open(os.path.join(golden_dir, "f0_config.json")) as f:
        fix = json.load(f)
    assert set(fix) == {"config_px2px.yaml", "config_px2px_SatCLIP.yaml"}
    for fname, entry in fix.items():
        for netG_name, ref_keys in entry["state_dict"].items():
            tree = json.loads(json.dumps(entry["tree"]))
            tree["base_configs"]["netG"] = netG_name
            if "raises" in ref_keys:
                with pytest.raises(NotImplementedError):
                    Px2Px_PL(to_attr(tree))
                continue
            m = Px2Px_PL(to_attr(tree))
            got = {k: list(v.shape) for k, v in m.state_dict().items() if not k.startswith("satclip_model.")}
            assert list(got) == list(ref_keys) or sorted(got) == sorted(ref_keys), (fname, netG_name, sorted(set(got) ^ set(ref_keys)))
            for k, shp in ref_keys.items():
                assert got[k] == shp, (fname, netG_name, k, got[k], shp)
            del m
    # one step from the plain config's tree, 6-block generator (configs[0]: bs 4, 256 x 256, padding 10)
    tree = json.loads(json.dumps(fix["config_px2px.yaml"]["tree"]))
    tree["base_configs"]["netG"] = "resnet_6blocks"
    size = 256
    if not full_width:
        tree["base_configs"]["ngf"] = tree["base_configs"]["ndf"] = 8
        size = 64
    assert tree["Data"]["padding"] is True and tree["Data"]["padding_amount"] == 10
    torch.manual_seed(0)
    m = Px2Px_PL(to_attr(tree)).to(dev).train()
    g = torch.Generator().manual_seed(11)
    batch = {"rgb": (0.02 + 0.58 * torch.rand(4, 3, size, size, generator=g)).to(dev), "nir": (0.05 + 0.75 * torch.rand(4, 1, size, size, generator=g)).to(dev)}
    before = {k: v.detach().clone() for k, v in m.state_dict().items() if k.endswith("model.1.weight")}
    out = m.train_batch(batch).as_dict()
    for k in ("loss_D", "loss_G", "loss_G_gan", "loss_G_l1"):
        assert np.isfinite(float(out[k])), (k, out[k])
    assert m.fused_trainer().padding == 10 and m.fused_trainer().G.data_pad == 10, "the YAML's padding did not reach the generator engine"
    after = m.state_dict()
    for k, v in before.items():
        assert not torch.equal(v, after[k]), f"{k} was not stepped"
    (opt_d, opt_g), _ = m.configure_optimizers()
    assert float(opt_d.state_dict()["state"][0]["step"]) == 1.0 and float(opt_g.state_dict()["state"][0]["step"]) == 1.0


def ganloss_labels_and_adam_without_gradients(dev, golden_dir, tol: Tol):
    """networks.py:229-256: the label values are registered buffers -- a checkpoint that carries other values (train.py:61-65,
    strict=False) must reach the loss; torch.optim.Adam leaves a parameter whose .grad is None untouched (values and moments)."""
    from model import networks
    from nirgan_hip.optim import HipAdam
    crit = networks.GANLoss("lsgan").to(dev)
    pred = torch.linspace(-1, 2, 2 * 30 * 30, device=dev).reshape(2, 1, 30, 30).contiguous().requires_grad_(True)
    mask = crit.get_target_tensor(pred, True)
    assert mask.shape == pred.shape and mask.stride() == (0, 0, 0, 0)
    assert mask.cpu().numpy().tobytes() == np.ones((2, 1, 30, 30), np.float32).tobytes()            # bit-exact label mask (a3)
    assert crit.get_target_tensor(pred, False).cpu().numpy().tobytes() == np.zeros((2, 1, 30, 30), np.float32).tobytes()
    l1 = crit(pred, True)
    close(l1, ((pred.detach().cpu() - 1.0) ** 2).mean(), tol.out, "lsgan real")
    crit.load_state_dict({"real_label": torch.tensor(0.9), "fake_label": torch.tensor(0.1)})
    l2 = crit(pred, True)
    close(l2, ((pred.detach().cpu() - 0.9) ** 2).mean(), tol.out, "lsgan with the checkpoint's real_label")
    close(crit(pred, False), ((pred.detach().cpu() - 0.1) ** 2).mean(), tol.out, "lsgan with the checkpoint's fake_label")
    l2.backward()
    close(pred.grad, 2.0 * (pred.detach().cpu() - 0.9) / pred.numel(), tol.out, "lsgan gradient")
    crit.real_label.fill_(1.0)                                                                        # in-place edits count too
    close(crit(pred, True), l1, 1e-6, "after fill_")
    # ---- Adam: tensors without a gradient stay put
    torch.manual_seed(3)
    net = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02).to(dev)
    opt = HipAdam(net.parameters(), lr=1e-2, betas=(0.5, 0.999), net=net)
    before = {k: q.detach().clone() for k, q in net.named_parameters()}
    opt.step()                                                                                        # nothing has a gradient: no step
    assert net._flat().step_count == 0
    gen = torch.Generator().manual_seed(4)
    with_grad = {"model.0.weight", "model.5.weight", "model.5.bias", "model.11.bias"}
    for k, q in net.named_parameters():
        q.grad = torch.randn(q.shape, generator=gen).to(dev) if k in with_grad else None
    opt.step()
    for k, q in net.named_parameters():
        moved = (q.detach() - before[k]).abs().max().item()
        if k in with_grad:
            assert 0.5e-2 < moved <= 1.001e-2, (k, moved)                 # first Adam step: lr * g / (|g| + eps)
        else:
            assert moved == 0.0, f"{k} has no gradient and moved by {moved}"
    m_, v_ = net._flat().moments()
    o, n_, _ = net._flat().slices["model.2.weight"]
    assert float(m_[o:o + n_].abs().max()) == 0.0 and float(v_[o:o + n_].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ a10: Pix2PixModel
def pix2pix_model_optimize_parameters(dev, golden_dir, tol: Tol):
    from model.pix2pix_model import Pix2PixModel
    base, z = load(golden_dir, "f1_g6_d.npz"), load(golden_dir, "f1_legacy.npz")
    assert str(z["base"]) == "f1_g6_d.npz"
    model = Pix2PixModel(px_config(6, 8))
    model.netG.load_state_dict(sub(base, "G0/"))
    model.netD.load_state_dict(sub(base, "D0/"))
    model.to(dev)
    model.set_input({"A": torch.from_numpy(base["rgb"]).to(dev), "B": torch.from_numpy(base["nir"]).to(dev)})
    assert model.real_A.shape[1] == 3 and model.real_B.shape[1] == 1
    model.optimize_parameters()
    close(model.fake_B, z["fake_B"], tol.out, "fake_B")
    close(model.loss_D, z["loss_D"], tol.out, "loss_D (x0.5)")
    close(model.loss_D_fake, z["loss_D_fake"], tol.out, "loss_D_fake")
    close(model.loss_D_real, z["loss_D_real"], tol.out, "loss_D_real")
    close(model.loss_G_GAN, z["loss_G_GAN"], tol.out, "loss_G_GAN")
    close(model.loss_G_L1, z["loss_G_L1"], tol.out, "loss_G_L1 (x lambda_L1)")
    close(model.loss_G, z["loss_G"], tol.out, "loss_G")
    # 0.5 * (the Lightning module's loss_D): pix2pix_model.py:128 vs pix2pix.py:206
    close(2.0 * model.loss_D, base["loss_D"], tol.out, "factor 0.5")
    for k, q in model.netD.named_parameters():
        assert not q.requires_grad                      # left frozen after the G update (pix2pix_model.py:151)
        if k not in O.shadowed_bias_keys("D"):
            grad_close(q.grad, z["gD/" + k], tol, "gD " + k)
            adam_close(q, z["D1/" + k], z["gD/" + k], "D1 " + k)
    for k, q in model.netG.named_parameters():
        if k not in O.shadowed_bias_keys("G", 6):
            grad_close(q.grad, z["gG/" + k], tol, "gG " + k)
            adam_close(q, z["G1/" + k], z["gG/" + k], "G1 " + k)
    model.optimize_parameters()                         # a second batch runs (D re-enabled, grads zeroed, state kept)
    assert np.isfinite(float(model.loss_G.detach())) and np.isfinite(float(model.loss_D.detach()))
    assert float(model.optimizer_G.state_dict()["state"][0]["step"]) == 2.0


# ------------------------------------------------------------------------------------------------ N4: fit loop
def _loaders(dev, n_train=2, n_val=1, size=32, seed=1):
    g = torch.Generator().manual_seed(seed)

    def mk():
        return {"rgb": 0.02 + 0.58 * torch.rand(2, 3, size, size, generator=g), "nir": 0.05 + 0.75 * torch.rand(2, 1, size, size, generator=g)}
    return [mk() for _ in range(n_train)], [mk() for _ in range(n_val)]


def fit_loop_schedulers_checkpoint_resume(dev, tmp_path, tol: Tol):
    """fit(): LR drops decided by ReduceLROnPlateau reach both fused Adam steps; the checkpoint has Lightning's layout
    (state_dict / optimizer_states / lr_schedulers) and a resumed run continues exactly like the uninterrupted one."""
    from model.pix2pix import Px2Px_PL
    from nirgan_hip.fit import fit
    cfg = px_config(6, 8, patience_g=0, patience_d=0)

    def fresh():
        torch.manual_seed(0)
        return Px2Px_PL(cfg).to(dev)
    train, val = _loaders(dev)
    # patience 0 + 'min' mode: any epoch that does not improve val/L1 by > 1e-4 relative cuts the lr by 10
    m_full = fresh()
    ck3 = tmp_path / "full.ckpt"
    hist = fit(m_full, train, val, max_epochs=3, log_every=1, ckpt_path=str(ck3), device=dev)
    assert len(hist["train"]) == 6 and len(hist["val"]) == 3 and {"val/L1", "val/L2", "val/PSNR", "val/SSIM"} <= set(hist["val"][0])
    tr = m_full.fused_trainer()
    assert tr.steps == 6 and tr.flatG.step_count == 6 and tr.flatD.step_count == 6
    # validation scalars against the oracle's metrics on the model's own prediction
    m_full.eval()
    p = m_full.predict_step(val[0]["rgb"].to(dev)).cpu()
    ref = O.calculate_metrics(p, val[0]["nir"], "val")
    m_full.logged.clear()
    m_full.validation_step({k: v.to(dev) for k, v in val[0].items()}, 0)
    for k in ("val/L1", "val/L2", "val/PSNR", "val/SSIM"):
        assert abs(float(m_full.logged[k]) - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-7, k
    # whatever the schedulers decided is what the fused steps use
    last = hist["lr"][-1]
    assert tr.lr_g == last["lr_g"] and tr.lr_d == last["lr_d"] and last["lr_g"] <= 2e-4
    vals = [h["val/L1"] for h in hist["val"]]
    expect = 2e-4
    best = float("inf")
    for v in vals:                                   # ReduceLROnPlateau(mode='min', patience=0, factor=0.1, threshold=1e-4 rel)
        if v < best * (1 - 1e-4):
            best = v
        else:
            expect *= 0.1
    assert abs(last["lr_g"] - expect) <= 1e-12 and abs(last["lr_d"] - expect) <= 1e-12
    # checkpoint layout + strict=False load the way train.py:61-65 / create_synthetic_dataset.py:24-26 do
    ck = torch.load(str(ck3), weights_only=False)
    assert {"state_dict", "optimizer_states", "lr_schedulers", "epoch", "global_step"} <= set(ck)
    assert "netG.model.1.weight" in ck["state_dict"] and "netD.model.11.bias" in ck["state_dict"] and "criterionGAN.real_label" in ck["state_dict"]
    assert float(ck["optimizer_states"][0]["state"][0]["step"]) == 6.0
    m2 = fresh()
    res = m2.load_state_dict(ck["state_dict"], strict=False)
    assert not res.missing_keys
    close(m2.netG.model[1].weight, m_full.netG.model[1].weight, 0.0, "reloaded weights")
    # interrupted after 2 epochs + resumed == uninterrupted (weights, Adam moments, step counts, scheduler state)
    m_a = fresh()
    ck2 = tmp_path / "two.ckpt"
    fit(m_a, train, val, max_epochs=2, log_every=0, ckpt_path=str(ck2), device=dev)
    m_b = fresh()
    with torch.no_grad():
        for q in m_b.parameters():
            q.add_(0.123)                               # everything must come from the checkpoint
    hist_b = fit(m_b, train, val, max_epochs=3, log_every=0, resume_from=str(ck2), device=dev)
    assert len(hist_b["val"]) == 1 and hist_b["lr"][-1] == hist["lr"][-1]
    trb = m_b.fused_trainer()
    assert trb.flatG.step_count == 6 and trb.flatD.step_count == 6
    for (k, qa), (_, qb) in zip(m_full.named_parameters(), m_b.named_parameters()):
        assert torch.equal(qb.detach().cpu(), qa.detach().cpu()), "resumed " + k      # every reduction has a fixed order: bitwise
    assert torch.equal(trb.flatG.m.cpu(), tr.flatG.m.cpu()) and torch.equal(trb.flatG.v.cpu(), tr.flatG.v.cpu()), "resumed Adam moments"
    # torch.optim.Adam accepts the optimizer state written here (a Lightning resume with the stock optimizer)
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(q, device="cpu")) for q in m_full.netG.parameters()],
                               lr=2e-4, betas=(0.5, 0.999))
    ref_opt.load_state_dict(ck["optimizer_states"][1])
    assert ref_opt.param_groups[0]["lr"] == last["lr_g"]


# ------------------------------------------------------------------------------------------------ N1: inference
def tiled_inference_and_checkpoint_loading(dev, golden_dir, tmp_path, tol: Tol):
    """create_synthetic_dataset.py:21-28,100-118: load a Lightning checkpoint with strict=False, eval, model(hr) under
    no_grad; predict_tiled on a scene that is no multiple of the tile core, against the oracle run through the same
    tiling; fp16 .npz writer."""
    from model.pix2pix import Px2Px_PL
    from nirgan_hip.inference import predict_tiled, save_nir_npz
    z = load(golden_dir, "f1_g9_rs_pad.npz")
    pad = int(z["padding"])
    ck = {"state_dict": {**{"netG." + k: v for k, v in sub(z, "G0/").items()}, **{"netD." + k: v for k, v in sub(z, "D0/").items()},
                         "criterionGAN.real_label": torch.tensor(1.0), "criterionGAN.fake_label": torch.tensor(0.0),
                         "satclip_model.model.nnet.last_layer.weight": torch.zeros(3)},      # unexpected key: strict=False tolerates it
          "epoch": 3, "global_step": 7, "pytorch-lightning_version": "1.9.0"}
    path = tmp_path / "S2.ckpt"
    torch.save(ck, str(path))
    torch.manual_seed(5)
    model = Px2Px_PL(px_config(9, 8, pad, 1.0))
    res = model.load_state_dict(torch.load(str(path), weights_only=False)["state_dict"], strict=False)
    assert res.unexpected_keys == ["satclip_model.model.nnet.last_layer.weight"] and not res.missing_keys
    model = model.eval().to(dev)
    hr = torch.from_numpy(z["rgb"])
    with torch.no_grad():
        pred = model(hr.to(dev))
    close(pred, z["pred"], tol.out, "model(hr) after checkpoint load")
    # tiled: 70 x 90 scene, 32-pixel tiles with 8 pixels of context per side (core 16: neither 70 nor 90 is a multiple)
    g = torch.Generator().manual_seed(11)
    scene = 0.02 + 0.58 * torch.rand(2, 3, 70, 90, generator=g)
    pG = sub(z, "G0/")
    got = predict_tiled(model, scene.to(dev), tile=32, margin=8, batch=5)
    want = O.predict_tiled(lambda t: O.px_forward(pG, t, 9, pad), scene, tile=32, margin=8, batch=7)
    assert got.shape == (2, 1, 70, 90)
    close(got, want, tol.out, "predict_tiled")
    fn = save_nir_npz(got[0], str(tmp_path), "tile_0")
    back = np.load(fn)["nir"]
    assert back.dtype == np.float16 and back.shape == (1, 70, 90)
    np.testing.assert_allclose(back.astype(np.float32), got[0].cpu().numpy(), atol=1e-3)
