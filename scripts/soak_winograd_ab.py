#!/usr/bin/env python3
"""400 training steps at the benchmark shape (bs 16, 256^2, 6-block, exact fp32) with the Winograd layers and, in a second process,
with the direct tiles (NIRGAN_OPTIONS=winograd=off): same data, same initial weights.  The first steps agree to rounding; afterwards the two
GAN trajectories separate as any two fp32 runs do, and must stay finite and converge alike."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
    import torch
    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    batches = []
    for _ in range(4):
        base = torch.nn.functional.interpolate(torch.rand(16, 3, 32, 32, generator=g), size=(256, 256), mode="bilinear", align_corners=False)
        rgb = (0.05 + 0.5 * base + 0.02 * torch.rand(16, 3, 256, 256, generator=g))
        nir = (0.1 + 0.6 * rgb[:, 0:1] + 0.3 * rgb[:, 1:2] * rgb[:, 2:3]).clamp(0, 1)
        batches.append((rgb.to(dev), nir.to(dev)))
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6)
    for step in range(400):
        v = tr.step(*batches[step % 4])
        if step < 3 or step % 50 == 49:
            d = v.as_dict()
            assert all(x == x and abs(x) < 1e6 for x in d.values()), d
            print(f"{step + 1:4d} L1 {d['loss_G_l1']:.6f} D {d['loss_D']:.5f} G {d['loss_G']:.5f}", flush=True)
    fin = all(bool(torch.isfinite(p).all()) for p in list(netG.parameters()) + list(netD.parameters()))
    print("parameters finite:", fin)
else:
    for name, env in (("winograd (default)", {}), ("direct tiles (NIRGAN_OPTIONS=winograd=off)", {"NIRGAN_OPTIONS": "winograd=off"})):
        print("==", name, flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env}, capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines()), flush=True)
        if r.returncode:
            print(r.stderr[-2000:])
            sys.exit(r.returncode)
