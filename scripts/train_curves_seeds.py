#!/usr/bin/env python3
"""Multi-seed training evidence (round 3's review, item 6): does a kernel variant train like its exact partner, or is a difference in the
end-point of one run just seed noise?  For every variant, SEEDS runs (seed = initial weights AND data order) of STEPS fused steps on 8 fixed
synthetic batches with a learnable relation (nir = smooth function of rgb); reported per variant: held-out L1 (mean over 4 held-out batches,
averaged over the last 5 evaluations, one every STEPS/20 steps -- a single end-point of a GAN run is itself noisy), mean loss_D and mean
training L1 over the last 10 % of the steps; then mean +- standard deviation over the seeds and the difference to the first variant in
units of the pooled seed spread.

    python3 scripts/train_curves_seeds.py winograd [steps] [seeds]     exact fp32: F(6x6,3x3) / F(4x4,3x3) / direct tiles, bs 16 @128^2
    python3 scripts/train_curves_seeds.py bf16 [steps] [seeds]         bs 16 @256^2: fp32, bf16 with the storage rules (256-wide tiles),
                                                                       the same on the 128-row tiles, bf16 with every tensor kept in fp32
"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.options import OPT
from nirgan_hip.trainer import Pix2PixTrainer

kind = sys.argv[1] if len(sys.argv) > 1 else "winograd"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
SEEDS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda:0"
size, coarse = (128, 16) if kind == "winograd" else (256, 32)
g = torch.Generator().manual_seed(3)
batches = []
for _ in range(12):
    base = torch.nn.functional.interpolate(torch.rand(16, 3, coarse, coarse, generator=g), size=(size, size), mode="bilinear", align_corners=False)
    rgb = (0.05 + 0.5 * base + 0.02 * torch.rand(16, 3, size, size, generator=g))
    nir = (0.1 + 0.6 * rgb[:, 0:1] + 0.3 * rgb[:, 1:2] * rgb[:, 2:3]).clamp(0, 1)
    batches.append((rgb.to(dev), nir.to(dev)))
held_out, batches = batches[8:], batches[:8]
if kind == "winograd":
    variants = [("direct tiles", dict(winograd="off"), "fp32"), ("F(6x6,3x3)", dict(winograd="f6"), "fp32"), ("F(4x4,3x3)", dict(winograd="f4"), "fp32")]
else:
    rules = dict(bf16_y=True, bf16_g=True, bf16_twin_only=True)
    norules = dict(bf16_y=False, bf16_g=False, bf16_twin_only=False)
    variants = [("fp32", {}, "fp32"), ("bf16 + storage rules, 256-wide tiles", dict(rules, tile256=True), "bf16"),
                ("bf16 + storage rules, 128-row tiles", dict(rules, tile256=False), "bf16"), ("bf16, all tensors fp32", dict(norules, tile256=True), "bf16")]
EVAL = max(1, STEPS // 20)
TAIL = max(1, STEPS // 10)
res = {n: [] for n, _, _ in variants}
for seed in range(SEEDS):
    order = torch.randperm(8 * ((STEPS + 7) // 8), generator=torch.Generator().manual_seed(1000 + seed)) % 8
    for name, opts, prec in variants:
        OPT.reset()
        for k, v in opts.items():
            setattr(OPT, k, v)
        torch.manual_seed(seed)
        netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
        netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
        tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision=prec)
        evals, tail_d, tail_l1 = [], [], []
        for step in range(STEPS):
            v = tr.step(*batches[int(order[step])])
            if step >= STEPS - TAIL:
                d = v.as_dict()
                tail_d.append(d["loss_D"]); tail_l1.append(d["loss_G_l1"])
            if step % EVAL == EVAL - 1 and step >= STEPS - 5 * EVAL:
                netG.eval()
                with torch.no_grad():
                    evals.append(sum(float((netG(r) - n_).abs().mean()) for r, n_ in held_out) / len(held_out))
                netG.train()
        r = (statistics.mean(evals), statistics.mean(tail_d), statistics.mean(tail_l1))
        res[name].append(r)
        print(f"seed {seed}  {name:40s} held-out L1 {r[0]:.4f}   loss_D (last {TAIL}) {r[1]:.3f}   train L1 (last {TAIL}) {r[2]:.4f}", flush=True)
        del tr, netG, netD
        torch.cuda.empty_cache()
OPT.reset()
print(f"\n{kind}: {SEEDS} seeds x {STEPS} steps, bs 16 @{size}^2, 6-block generator; mean +- standard deviation over the seeds")
base = variants[0][0]
for name, _, _ in variants:
    cols = list(zip(*res[name]))
    line = f"  {name:40s}"
    for label, c, b in (("held-out L1", cols[0], list(zip(*res[base]))[0]), ("loss_D", cols[1], list(zip(*res[base]))[1]), ("train L1", cols[2], list(zip(*res[base]))[2])):
        m, sd = statistics.mean(c), (statistics.stdev(c) if len(c) > 1 else 0.0)
        line += f"   {label} {m:.4f} +- {sd:.4f}"
        if name != base and len(c) > 1:
            pooled = ((sd ** 2 + statistics.stdev(b) ** 2) / 2) ** 0.5
            line += f" ({(m - statistics.mean(b)) / max(pooled, 1e-12):+.2f} sd vs {base})"
    print(line)
