#!/usr/bin/env python3
"""One step at ngf 64 in fp32 / bf16 operand mode with the generator's last layer as direct fp32 kernels (default) and as bf16 tap planes
(NIRGAN_NO_ENDCONV=1): prediction and generator gradient against the fp32 direct run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer
dev = "cuda:0"
def run(no_endconv, prec):
    if no_endconv: os.environ["NIRGAN_NO_ENDCONV"] = "1"
    else: os.environ.pop("NIRGAN_NO_ENDCONV", None)
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, precision=prec, lr=0.0)
    g = torch.Generator().manual_seed(5)
    rgb = (0.02 + 0.58 * torch.rand(2, 3, 128, 128, generator=g)).to(dev)
    nir = (0.05 + 0.75 * torch.rand(2, 1, 128, 128, generator=g)).to(dev)
    out = tr.step(rgb, nir).as_dict()
    torch.cuda.synchronize()
    return tr.G.pred.clone(), tr.flatG.grad.clone(), out
ref, gref, o = run(False, "fp32"); print("fp32 direct", o["loss_G_l1"], ref.abs().mean().item())
for ne in (False, True):
    for prec in ("fp32", "bf16"):
        p, g, o = run(ne, prec)
        print("no_endconv" if ne else "direct", prec, "pred rel to fp32-direct", ((p - ref).norm() / ref.norm()).item(), "grad rel", ((g - gref).norm() / gref.norm()).item(), "l1", o["loss_G_l1"])
