#!/usr/bin/env python3
"""Same-process A/B of two builds of the library on the benchmark's step (configs[1]): ONE trainer, the backend switched between
timing blocks (shipped library against scripts/diag/libnirgan_ab.so from ab_build.sh); medians over alternating blocks, plus the
per-op times of the largest launches under each build."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip import lib as L
from nirgan_hip.trainer import Pix2PixTrainer
DEV = "cuda:0"
A = L.backend()
B = L._CLib(os.path.join(ROOT, "scripts", "diag", "libnirgan_ab.so"))
torch.manual_seed(0)
netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
tr = Pix2PixTrainer(netG, netD, n_blocks=6)
g = torch.Generator().manual_seed(1)
rgb = (0.02 + 0.58 * torch.rand(16, 3, 256, 256, generator=g)).to(DEV)
nir = (0.05 + 0.75 * torch.rand(16, 1, 256, 256, generator=g)).to(DEV)


def block(be, steps=20):
    L.set_backend(be)
    for _ in range(3):
        tr.step(rgb, nir)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps):
        tr.step(rgb, nir)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps


ta, tb = [], []
for _ in range(6):
    ta.append(block(A))
    tb.append(block(B))
L.set_backend(None)
ma, mb = statistics.median(ta), statistics.median(tb)
print("shipped  ms/step:", " ".join(f"{t:.3f}" for t in ta), f"  median {ma:.3f}  = {16e3 / ma:.1f} tiles/s")
print("variant  ms/step:", " ".join(f"{t:.3f}" for t in tb), f"  median {mb:.3f}  = {16e3 / mb:.1f} tiles/s   ({(ma / mb - 1) * 100:+.2f} %)")
