#!/usr/bin/env python3
"""Diagnostic (GPU box): full-size (ngf 64) gradient errors against the oracle in fp64 when the kinks are taken out of the
comparison: (a) engines with a given smooth output gradient, (b) the fused step with `nir` moved off the L1 kink
(|pred - nir| >= 1e-3 everywhere).  Prints per-tensor rel-L2 of hip-vs-f64 and oracle32-vs-f64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nir-gan_amd"), os.path.join(ROOT, "oracle")]
import torch
import nirgan_oracle as O
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
DEV = "cuda:0"
def rel(a, b): return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()
def synth(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return 0.02 + 0.58 * torch.rand(B, 3, H, W, generator=g), 0.05 + 0.75 * torch.rand(B, 1, H, W, generator=g)

nb = 6
torch.manual_seed(0)
netG = networks.define_G(3, 1, 64, f"resnet_{nb}blocks", "instance", False, "normal", 0.02)
torch.manual_seed(0)
netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
pG = {k: v.clone() for k, v in netG.state_dict().items()}
pD = {k: v.clone() for k, v in netD.state_dict().items()}
rgb, nir = synth(1, 256, 256, 1234)
p64 = {k: v.double() for k, v in pG.items()}
with torch.no_grad():
    pred64 = O.px_forward(p64, rgb.double(), nb, 0)
d = pred64.float() - nir
near = d.abs() < 1e-3
print("pixels within 1e-3 of the L1 kink:", int(near.sum()))
nir = torch.where(near, pred64.float() - torch.where(d >= 0, 2e-3, -2e-3), nir)
assert ((pred64.float() - nir).abs() >= 1e-3).all()
tr = Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=nb)
tr.step(rgb.to(DEV), nir.to(DEV))
r32 = O.OracleTrainer(pG, pD, nb); r32.step(rgb, nir)
r64 = O.OracleTrainer(p64, {k: v.double() for k, v in pD.items()}, nb); r64.step(rgb.double(), nir.double())
print(f"pred: hip {rel(tr.G.pred, r64.last['pred']):.2e} oracle32 {rel(r32.last['pred'], r64.last['pred']):.2e}")
for name, gh, which in (("D", tr.flatD.grad_views(), "grads_D"), ("G", tr.flatG.grad_views(), "grads_G")):
    sh = O.shadowed_bias_keys(name, nb)
    for k, v64 in r64.last[which].items():
        if k in sh: continue
        print(f"fused {name} {k:32s} hip-vs-f64 {rel(gh[k], v64):.2e}  oracle32-vs-f64 {rel(r32.last[which][k], v64):.2e}")
# (a) generator engine, smooth dout
for nb_, pad in ((6, 0), (9, 10)):
    torch.manual_seed(0)
    net = networks.define_G(3, 1, 64, f"resnet_{nb_}blocks", "instance", False, "normal", 0.02)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    rgb, _ = synth(1, 256, 256, 1234)
    dout = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(2))
    net = net.to(DEV); net.data_pad = pad
    pred = net(rgb.to(DEV)); pred.backward(dout.to(DEV))
    res = {}
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        p = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        O.px_forward(p, rgb.to(dt), nb_, pad).backward(dout.to(dt))
        res[tag] = p
    for k, q in net.named_parameters():
        if k in O.shadowed_bias_keys("G", nb_): continue
        print(f"engine G{nb_} pad{pad} {k:32s} hip-vs-f64 {rel(q.grad, res['64'][k].grad):.2e}  oracle32-vs-f64 {rel(res['32'][k].grad, res['64'][k].grad):.2e}")
# discriminator engine
torch.manual_seed(0)
netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
sd = {k: v.clone() for k, v in netD.state_dict().items()}
rgb, nir = synth(2, 256, 256, 99)
x = torch.cat((rgb, nir), 1)
dout = torch.randn(2, 1, 30, 30, generator=torch.Generator().manual_seed(4))
netD = netD.to(DEV)
xg = x.to(DEV).requires_grad_(True)
netD(xg).backward(dout.to(DEV))
res = {}
for tag, dt in (("32", torch.float32), ("64", torch.float64)):
    p = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.detach().to(dt).clone().requires_grad_(True)
    O.discriminator_forward(p, xr).backward(dout.to(dt))
    res[tag] = (p, xr)
for k, q in netD.named_parameters():
    if k in O.shadowed_bias_keys("D"): continue
    print(f"engine D {k:32s} hip-vs-f64 {rel(q.grad, res['64'][0][k].grad):.2e}  oracle32-vs-f64 {rel(res['32'][0][k].grad, res['64'][0][k].grad):.2e}")
print(f"engine D dx hip-vs-f64 {rel(xg.grad, res['64'][1].grad):.2e} oracle32-vs-f64 {rel(res['32'][1].grad, res['64'][1].grad):.2e}")
