#!/usr/bin/env python3
"""The medium-width unforced-kink comparison of tests/test_gpu_nets.py::test_medium_width_nets_take_the_winograd_paths with the
three-term split tiles on / off: worst gradient tensors against the fp32 oracle (kink flips dominate; this prints what moved)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nirgan_oracle as O
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer
from nirgan_hip.options import OPT
DEV = "cuda:0"
WINO = sys.argv[1] if len(sys.argv) > 1 else "f6"          # OPT.winograd: f6 | f4 | off
for seed in (7, 17, 27):
    for conf in ({"split3": False}, {"split3": True, "split3_wino": False, "pair_phases": False}, {"split3": True, "split3_wino": True, "pair_phases": False},
                 {"split3": True, "split3_wino": True, "pair_phases": True}):
        OPT.reset()
        OPT.winograd = WINO
        for k, v in conf.items():
            setattr(OPT, k, v)
        torch.manual_seed(seed)
        netG = networks.define_G(3, 1, 32, "resnet_6blocks", "instance", False, "normal", 0.02)
        netD = networks.define_D(4, 32, "basic", 3, "instance", "normal", 0.02)
        G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
        D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()}
        g = torch.Generator().manual_seed(seed + 1)
        rgb, nir = torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 1, 64, 64, generator=g)
        tr = Pix2PixTrainer(netG.to(DEV), netD.to(DEV), n_blocks=6, lr=0.0)
        tr.step(rgb.to(DEV), nir.to(DEV))
        ref = O.OracleTrainer(G0, D0, 6, lr=0.0)
        ref.step(rgb, nir)
        gG = tr.flatG.grad_views()
        errs = []
        for k, v in ref.last["grads_G"].items():
            if v is not None and k not in O.shadowed_bias_keys("G", 6):
                errs.append(((gG[k].cpu() - v).norm().item() / max(v.norm().item(), 1e-20), k))
        errs.sort(reverse=True)
        pe = (tr.G.pred.cpu() - ref.last["pred"]).abs().max().item() / ref.last["pred"].abs().max().item()
        print(f"winograd={WINO} seed {seed} {conf}: pred err {pe:.2e}; worst rel L2: " + ", ".join(f"{e:.2e} {k}" for e, k in errs[:3]), flush=True)
