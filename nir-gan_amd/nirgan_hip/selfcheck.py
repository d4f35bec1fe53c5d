"""smoke(): one small fused train step on the GPU, checked against the CPU oracle.

Called by ``__graft_entry__.smoke()``.  The oracle (oracle/nirgan_oracle.py, test
infrastructure) is imported HERE as the checker only.
"""
from __future__ import annotations

import torch


def smoke_train_step(device="cuda:0", ngf=16, size=64, batch=2, n_blocks=6):
    import nirgan_oracle as O            # checker
    from model import networks
    from .trainer import Pix2PixTrainer

    torch.manual_seed(0)
    netG = networks.define_G(3, 1, ngf, f"resnet_{n_blocks}blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, ngf, "basic", 3, "instance", "normal", 0.02)
    pG = {k: v.clone() for k, v in netG.state_dict().items()}
    pD = {k: v.clone() for k, v in netD.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    rgb = 0.02 + 0.58 * torch.rand(batch, 3, size, size, generator=g)
    nir = 0.05 + 0.75 * torch.rand(batch, 1, size, size, generator=g)
    netG, netD = netG.to(device), netD.to(device)
    tr = Pix2PixTrainer(netG, netD, n_blocks=n_blocks)
    out = tr.step(rgb.to(device), nir.to(device)).as_dict()
    ref = O.OracleTrainer(pG, pD, n_blocks)
    o = ref.step(rgb, nir)
    pred = tr.G.pred.cpu()
    err = (pred - ref.last["pred"]).abs().max().item() / ref.last["pred"].abs().max().item()
    assert err < 1e-3, f"generator output differs from the oracle: {err:.3e}"
    for k in ("loss_D", "loss_G"):
        e = abs(out[k] - float(o[k])) / abs(float(o[k]))
        assert e < 1e-3, f"{k}: {out[k]} vs {float(o[k])}"
    gG = tr.flatG.grad_views()
    for k, v in ref.last["grads_G"].items():
        if k in O.shadowed_bias_keys("G", n_blocks):
            continue
        e = (gG[k].cpu() - v).norm().item() / max(v.norm().item(), 1e-20)
        assert e < 2e-3, f"grad {k}: rel L2 {e:.3e}"
    print(f"smoke ok: pred err {err:.2e}, loss_D {out['loss_D']:.5f}, loss_G {out['loss_G']:.5f}")
    return out
