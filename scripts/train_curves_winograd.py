#!/usr/bin/env python3
"""Loss curves of the exact-fp32 step with the three convolution algorithms of the residual blocks -- F(6x6,3x3) (default),
F(4x4,3x3) (OPT.winograd = 'f4'), direct tiles (OPT.winograd = 'off') -- on the same data and initial weights: 240 steps over 8 fixed
synthetic batches with a learnable relation (nir = smooth function of rgb), bs 16 @128^2, 6-block generator (argv[1] = steps, default 240;
round 3: 2 000 steps with the L1 on 4 held-out batches at the end, profiles/r03_winograd_training_curves_2000.txt).  GAN training is
chaotic: the curves separate after a few dozen steps whatever the rounding (see the fp32 / bf16x3 columns of
profiles/r01_precision_training_curves.txt); what to look for is that they stay in one band and reach the same level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer

dev = "cuda:0"
g = torch.Generator().manual_seed(3)
batches = []
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 240
EVERY = max(20, STEPS // 12)
for _ in range(12):
    base = torch.nn.functional.interpolate(torch.rand(16, 3, 16, 16, generator=g), size=(128, 128), mode="bilinear", align_corners=False)
    rgb = (0.05 + 0.5 * base + 0.02 * torch.rand(16, 3, 128, 128, generator=g))
    nir = (0.1 + 0.6 * rgb[:, 0:1] + 0.3 * rgb[:, 1:2] * rgb[:, 2:3]).clamp(0, 1)
    batches.append((rgb.to(dev), nir.to(dev)))
variants = (("F(6x6,3x3)", "f6"), ("F(4x4,3x3)", "f4"), ("direct tiles", "off"))
from nirgan_hip.options import OPT
print("step   " + "".join(f"{n:>30s}" for n, _ in variants))
held_out, batches = batches[8:], batches[:8]
rows, first, val = {}, {}, {}
for name, env in variants:
    OPT.winograd = env
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6)
    for step in range(STEPS):
        v = tr.step(*batches[step % 8])
        if step == 0:
            first[name] = v.as_dict()
        if step % EVERY == EVERY - 1:
            d = v.as_dict()
            rows.setdefault(step + 1, []).append(f"  L1 {d['loss_G_l1']:.4f} D {d['loss_D']:.3f} Ggan {d['loss_G_gan']:.3f}")
    netG.eval()
    with torch.no_grad():
        val[name] = sum(float((netG(r) - n_).abs().mean()) for r, n_ in held_out) / len(held_out)
    netG.train()
for step, cols in rows.items():
    print(f"{step:4d}   " + "".join(f"{c:>30s}" for c in cols))
print("first step (same weights, same batch):")
for n, d in first.items():
    print(f"  {n:14s} loss_D {d['loss_D']:.7f}  loss_G {d['loss_G']:.7f}  L1 {d['loss_G_l1']:.7f}")
print("L1 on 4 held-out batches after", STEPS, "steps:", ", ".join(f"{n} {v:.4f}" for n, v in val.items()))
