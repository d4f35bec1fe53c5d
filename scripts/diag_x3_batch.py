#!/usr/bin/env python3
"""Does a prediction depend on how the batch is cut?  bs 4 @128 as one batch against the same tiles as two batches of two (no streams),
per setting of the split tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer
from nirgan_hip.options import OPT
DEV = "cuda:0"
g = torch.Generator().manual_seed(21)
rgb = (0.02 + 0.58 * torch.rand(4, 3, 128, 128, generator=g)).to(DEV)
nir = (0.05 + 0.75 * torch.rand(4, 1, 128, 128, generator=g)).to(DEV)
for conf in ({"split3": False}, {"split3": True, "split3_wino": False}, {"split3": True, "split3_wino": True}):
    OPT.reset()
    for k, v in conf.items():
        setattr(OPT, k, v)
    preds = []
    for parts in (1, 2):
        torch.manual_seed(0)
        G = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
        D = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
        tr = Pix2PixTrainer(G, D, n_blocks=6, lr=0.0)
        n = 4 // parts
        ps = []
        for i in range(parts):
            tr.step(rgb[i * n:(i + 1) * n].contiguous(), nir[i * n:(i + 1) * n].contiguous())
            ps.append(tr.pred.clone())
        preds.append(torch.cat(ps))
        names = sorted({n_ for pl in (tr.G.fwd,) for n_, _ in pl.ops if isinstance(n_, str)})
    err = (preds[0] - preds[1]).abs().max().item() / preds[0].abs().max().item()
    print(conf, "pred one batch vs two halves: rel max err", f"{err:.3e}", "bitwise" if torch.equal(preds[0], preds[1]) else "")
