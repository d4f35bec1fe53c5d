#!/usr/bin/env python3
"""Loss curves at the benchmark shape (bs 16 @256^2, 6-block generator, so that every storage rule of DESIGN 3.3 applies to the trunk):
exact fp32, the bf16 operand mode with its storage rules (convolution outputs and data gradients in front of an instance norm stored as
bf16, no fp32 store where every reader takes the twin) and the same mode with every tensor kept in fp32 -- same data, same initial
weights; argv[1] = steps (default 600); L1 on 4 held-out batches at the end.  GAN training is chaotic: the curves separate after a few
dozen steps whatever the rounding; what to look for is that they stay in one band and reach the same level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.options import OPT
from nirgan_hip.trainer import Pix2PixTrainer

dev = "cuda:0"
g = torch.Generator().manual_seed(3)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
EVERY = max(20, STEPS // 12)
batches = []
for _ in range(12):
    base = torch.nn.functional.interpolate(torch.rand(16, 3, 32, 32, generator=g), size=(256, 256), mode="bilinear", align_corners=False)
    rgb = (0.05 + 0.5 * base + 0.02 * torch.rand(16, 3, 256, 256, generator=g))
    nir = (0.1 + 0.6 * rgb[:, 0:1] + 0.3 * rgb[:, 1:2] * rgb[:, 2:3]).clamp(0, 1)
    batches.append((rgb.to(dev), nir.to(dev)))
held_out, batches = batches[8:], batches[:8]
variants = (("fp32", "fp32", True), ("bf16 + storage rules", "bf16", True), ("bf16, all tensors fp32", "bf16", False))
print("step   " + "".join(f"{n:>34s}" for n, _, _ in variants), flush=True)
rows, first, val, stored = {}, {}, {}, {}
for name, prec, rules in variants:
    OPT.bf16_y = OPT.bf16_g = OPT.bf16_twin_only = rules
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision=prec)
    for step in range(STEPS):
        v = tr.step(*batches[step % 8])
        if step == 0:
            first[name] = v.as_dict()
            stored[name] = (sum(1 for l in [tr.G.L1, tr.G.L2, tr.G.L3, tr.G.U1, tr.G.U2] + [c for _, a, b in tr.G.blocks for c in (a, b)] if l.y.is16),
                            sum(1 for h in tr.G.twinned + tr.D2.twinned if h.fp32_dead))
        if step % EVERY == EVERY - 1:
            d = v.as_dict()
            rows.setdefault(step + 1, []).append(f"  L1 {d['loss_G_l1']:.4f} D {d['loss_D']:.3f} Ggan {d['loss_G_gan']:.3f}")
    netG.eval()
    with torch.no_grad():
        val[name] = sum(float((netG(r) - n_).abs().mean()) for r, n_ in held_out) / len(held_out)
    del tr
OPT.reset()
for step, cols in rows.items():
    print(f"{step:4d}   " + "".join(f"{c:>34s}" for c in cols))
print("first step (same weights, same batch):")
for n, d in first.items():
    print(f"  {n:24s} loss_D {d['loss_D']:.7f}  loss_G {d['loss_G']:.7f}  L1 {d['loss_G_l1']:.7f}   bf16-stored conv outputs {stored[n][0]}, twin-only buffers {stored[n][1]}")
print("L1 on 4 held-out batches after", STEPS, "steps:", ", ".join(f"{n} {v:.4f}" for n, v in val.items()))
