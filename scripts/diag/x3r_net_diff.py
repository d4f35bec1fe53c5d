#!/usr/bin/env python3
"""Where does a fused step with conv_x3r_kernel (OPT.x3_r4) first differ from the same step on the eight-wave tile?  configs[3]'s shape."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from model.generator_inject import define_G_inject
from nirgan_hip.options import OPT
from nirgan_hip.trainer import Pix2PixTrainer
DEV = "cuda:0"
B, H, nb, pad = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 9, 10
ns = types.SimpleNamespace
def build(flag):
    OPT.reset(); OPT.x3_r4 = flag
    torch.manual_seed(0)
    netG = define_G_inject(ns(base_configs=ns(input_nc=3, output_nc=1, ngf=64, netG=f"resnet_{nb}blocks", norm="instance", no_dropout=True, init_type="normal", init_gain=0.02),
                              satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0, scaling_param=True, scaling_param_init=0.5)))
    torch.manual_seed(0)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    return Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=nb, padding=pad, inject={"style": "multiply", "use_scale": True}, lr=0.0)
g = torch.Generator().manual_seed(553)
rgb = (0.02 + 0.58 * torch.rand(B, 3, H, H, generator=g)).to(DEV); nir = (0.05 + 0.75 * torch.rand(B, 1, H, H, generator=g)).to(DEV)
emb = torch.randn(B, 256, generator=torch.Generator().manual_seed(554)).to(DEV)
res = {}
for flag in (False, True):
    tr = build(flag)
    tr.step(rgb, nir, emb)
    torch.cuda.synchronize()
    G_, D2 = tr.G, tr.D2
    t = {"L1": G_.L1.out.t, "L2": G_.L2.out.t, "L3": G_.L3.out.t}
    for j, (_, c1, c2) in enumerate(G_.blocks):
        t[f"b{j}c1.y"] = c1.y.t; t[f"b{j}c2.out"] = c2.out.t
    t.update({"U1": G_.U1.out.t, "U2": G_.U2.out.t, "pred": G_.pred})
    for i, c in enumerate((D2.C1, D2.C2, D2.C3, D2.C4)):
        t[f"D.C{i+1}"] = c.out.t
    for k, v in tr.flatD.grad_views().items(): t["gD " + k] = v
    for k, v in tr.flatG.grad_views().items(): t["gG " + k] = v
    res[flag] = {k: v.detach().clone() for k, v in t.items()}
    names = [n for pl in (tr.G.fwd, tr.G.bwd, tr.D2.fwd, tr.D2.bwd, tr.D1.bwd_pred) for n, _ in pl.ops if isinstance(n, str)]
    del tr
for k in res[False]:
    a, b = res[False][k], res[True][k]
    if not torch.equal(a, b):
        e = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
        print(f"{k}: differs, max rel {e:.3e}")
print("compared", len(res[False]), "tensors")
