#!/usr/bin/env python3
"""Diagnostic (GPU box): error of the HIP gradients and of the fp32 CPU oracle's gradients, both
measured against the same step evaluated by the oracle in fp64.  Shows how much of a
HIP-vs-oracle difference is fp32 conditioning of the problem itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nir-gan_amd"), os.path.join(ROOT, "oracle")]
import torch
import nirgan_oracle as O
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.manual_seed(0)
netG = networks.define_G(3, 1, 64, f"resnet_{nb}blocks", "instance", False, "normal", 0.02)
torch.manual_seed(0)
netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
pG = {k: v.clone() for k, v in netG.state_dict().items()}
pD = {k: v.clone() for k, v in netD.state_dict().items()}
g = torch.Generator().manual_seed(1234)
rgb = 0.02 + 0.58 * torch.rand(1, 3, 256, 256, generator=g)
nir = 0.05 + 0.75 * torch.rand(1, 1, 256, 256, generator=g)
tr = Pix2PixTrainer(netG.to("cuda:0"), netD.to("cuda:0"), n_blocks=nb, padding=pad)
tr.step(rgb.cuda(), nir.cuda())
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
r32 = O.OracleTrainer(pG, pD, nb, padding=pad); r32.step(rgb, nir)
r64 = O.OracleTrainer({k: v.double() for k, v in pG.items()}, {k: v.double() for k, v in pD.items()}, nb, padding=pad)
r64.step(rgb.double(), nir.double())
def rel(a, b): return ((a.double() - b).norm() / b.norm().clamp_min(1e-30)).item()
print(f"pred: hip {rel(tr.G.pred.cpu(), r64.last['pred']):.2e} oracle32 {rel(r32.last['pred'], r64.last['pred']):.2e}")
for name, gh, which in (("D", tr.flatD.grad_views(), "grads_D"), ("G", tr.flatG.grad_views(), "grads_G")):
    sh = O.shadowed_bias_keys(name, nb)
    for k, v64 in r64.last[which].items():
        if k in sh: continue
        print(f"{name} {k:32s} hip-vs-f64 {rel(gh[k].cpu(), v64):.2e}  oracle32-vs-f64 {rel(r32.last[which][k], v64):.2e}  hip-vs-oracle32 {rel(gh[k].cpu(), r32.last[which][k].double()):.2e}")
