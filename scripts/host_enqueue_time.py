#!/usr/bin/env python3
"""Host time to ENQUEUE one fused step (ctypes calls of the plans, no synchronisation) against the device time of the step:
the margin that keeps the launch stream fed (one Python process per GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer

dev = "cuda:0"
torch.manual_seed(0)
netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
tr = Pix2PixTrainer(netG, netD, n_blocks=6)
rgb, nir = torch.rand(16, 3, 256, 256, device=dev), torch.rand(16, 1, 256, 256, device=dev)
for _ in range(3):
    tr.step(rgb, nir)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter()
    tr.step(rgb, nir)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t_all) / 20
host.sort()
print(f"enqueue per step: median {host[10] * 1e3:.2f} ms (min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f});  step {t_all * 1e3:.2f} ms on the device")
