#!/usr/bin/env python3
"""Loss curves of the exact-fp32 tiles (OPT.split3 = False) against the three-term split tiles with every round-5 form on (default):
the same data and initial weights, 240 steps over 8 fixed synthetic batches with a learnable relation, bs 16 @256^2 (the benchmark's
geometry: paired phases, pixel-paired first layer, spread walk, Winograd planes all active), 6-block generator.  GAN training amplifies
rounding noise, so the curves separate in the last digits after some steps; what must hold is that they stay on top of each other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.options import OPT
from nirgan_hip.trainer import Pix2PixTrainer

dev = "cuda:0"
g = torch.Generator().manual_seed(3)
batches = []
for _ in range(8):
    base = torch.nn.functional.interpolate(torch.rand(16, 3, 32, 32, generator=g), size=(256, 256), mode="bilinear", align_corners=False)
    rgb = (0.05 + 0.5 * base + 0.02 * torch.rand(16, 3, 256, 256, generator=g))
    nir = (0.1 + 0.6 * rgb[:, 0:1] + 0.3 * rgb[:, 1:2] * rgb[:, 2:3]).clamp(0, 1)
    batches.append((rgb.to(dev), nir.to(dev)))
names = ("exact fp32 tiles", "exact fp32, direct tiles only", "split tiles (default)")
print("step   " + "".join(f"{p:>42s}" for p in names))
rows = {}
for split, wino in ((False, "f6"), (False, "off"), (True, "f6")):      # (two exact-fp32 summation orders show what rounding noise alone does to a GAN's trajectory)
    OPT.reset()
    OPT.split3 = split
    OPT.winograd = wino
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6)
    for step in range(240):
        v = tr.step(*batches[step % 8])
        if step in (0, 1, 4, 9) or step % 20 == 19:
            d = v.as_dict()
            rows.setdefault(step + 1, []).append(f"  L1 {d['loss_G_l1']:.5f} D {d['loss_D']:.4f} Ggan {d['loss_G_gan']:.4f}")
OPT.reset()
for step, cols in rows.items():
    print(f"{step:4d}   " + "".join(f"{c:>42s}" for c in cols))
