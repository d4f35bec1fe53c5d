"""the inference flow of create_synthetic_dataset.py (:21-28 checkpoint loading with strict=False,
:106-107 model(hr) under no_grad, :49-52/:117 fp16 .npz) on the emulated C ABI (host logic), plus tiling."""
import os

import numpy as np
import pytest
import torch

from emu_backend import EmuBackend
from nirgan_hip import lib as L
from utils.config import to_attr


@pytest.fixture()
def emu():
    L.set_backend(EmuBackend())
    yield
    L.set_backend(None)


def small_cfg(pad=True):
    return to_attr({
        "base_configs": {"isTrain": True, "input_nc": 3, "output_nc": 1, "ngf": 8, "ndf": 8, "netD": "basic",
                         "netG": "resnet_6blocks", "norm": "instance", "no_dropout": True, "init_type": "normal",
                         "init_gain": 0.02, "n_layers_D": 3, "gan_mode": "lsgan", "lr": 0.0002, "beta1": 0.5,
                         "lambda_GAN": 1.0, "lambda_L1": 100.0, "lambda_ssim": 0.0, "lambda_hist": 0.0,
                         "lambda_rs_losses": 0.0, "rs_losses_criterium": "l1", "internal_rs_loss_weights": {}},
        "satclip": {"use_satclip": False},
        "Schedulers": {"metric": "val/L1", "patience_g": 25, "patience_d": 25},
        "Data": {"padding": pad, "padding_amount": 10}})


def test_reference_checkpoint_loads_and_predicts(emu, golden_dir, tmp_path, capsys):
    import nirgan_oracle as O
    from model.pix2pix import Px2Px_PL
    from nirgan_hip.inference import save_nir_npz
    z = np.load(os.path.join(golden_dir, "f1_g6_d.npz"))
    sd = {"netG." + k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("G0/")}
    sd.update({"netD." + k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("D0/")})
    sd["criterionGAN.real_label"], sd["criterionGAN.fake_label"] = torch.tensor(1.0), torch.tensor(0.0)
    sd["satclip_model.some.unexpected.weight"] = torch.zeros(3)           # ignored by strict=False, as in the reference
    ckpt = tmp_path / "S2.ckpt"
    torch.save({"state_dict": sd, "epoch": 3}, ckpt)
    model = Px2Px_PL(small_cfg())
    res = model.load_state_dict(torch.load(ckpt)["state_dict"], strict=False)
    assert res.unexpected_keys == ["satclip_model.some.unexpected.weight"] and not res.missing_keys
    model = model.eval()
    rgb = torch.from_numpy(z["rgb"])
    with torch.no_grad():
        pred = model(rgb)
    assert not pred.requires_grad and pred.shape == (2, 1, 32, 32)
    pG = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("G0/")}
    ref = O.px_forward(pG, rgb, 6, padding=10)
    assert (pred - ref).abs().max() < 1e-4 * ref.abs().max()
    # forward-only engines were used: no backward plan was built
    eng = next(iter(model.netG._pool().free.values()))[0]
    assert len(eng.bwd.ops) == 0
    fn = save_nir_npz(pred[0], str(tmp_path), "tile_0")
    back = np.load(fn)["nir"]
    assert back.dtype == np.float16 and back.shape == (1, 32, 32)
    assert np.abs(back.astype(np.float32) - pred[0].numpy()).max() < 1e-3


def test_predict_tiled_shapes_and_single_tile_identity(emu, capsys):
    from model.pix2pix import Px2Px_PL
    from nirgan_hip.inference import predict_tiled
    model = Px2Px_PL(small_cfg(pad=False)).eval()
    g = torch.Generator().manual_seed(1)
    rgb = 0.02 + 0.58 * torch.rand(1, 3, 24, 24, generator=g)
    one = predict_tiled(model, rgb, tile=32, margin=4)
    assert one.shape == (1, 1, 24, 24)
    with torch.no_grad():
        ref = model(torch.nn.functional.pad(rgb, (4, 4, 4, 4), mode="reflect"))[:, :, 4:28, 4:28]
    assert (one - ref).abs().max() < 1e-5
    big = 0.02 + 0.58 * torch.rand(2, 3, 50, 37, generator=g)
    out = predict_tiled(model, big, tile=32, margin=4, batch=3)
    assert out.shape == (2, 1, 50, 37) and torch.isfinite(out).all()


def test_histogram_match_helper(emu):
    """nirgan_hip.inference.histogram_match = the helper of create_synthetic_dataset.py:34-47 (resize + per-tile matching)."""
    import nirgan_oracle as O
    from nirgan_hip.inference import histogram_match
    g = torch.Generator().manual_seed(3)
    pred = torch.rand(2, 1, 24, 28, generator=g)
    s2 = torch.rand(2, 1, 6, 7, generator=g) * 0.5 + 0.2
    got = histogram_match(pred, s2)
    ref = O.histogram_match(pred, s2)
    assert got.shape == (2, 1, 24, 28)
    assert (got - ref).abs().max().item() <= 1e-6
    q = (pred * 8).round() / 8                                  # heavy ties on both sides
    r = (torch.rand(2, 1, 24, 28, generator=g) * 4).round() / 4
    assert (histogram_match(q, r) - O.histogram_match(q, r)).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        histogram_match(pred[:, 0], s2)
