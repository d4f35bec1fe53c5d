#!/usr/bin/env python3
"""Same-process A/B of engine options on the benchmark's step (configs[1]): two trainers (same weights, same batch), built under the two
option sets and timed in alternating blocks.  Usage: ab_options.py name=value[,name=value...] [name=value,...]   (second set: defaults).
Python-side hooks too: X3_WGRAD_TARGET=<workgroups> overrides the split tile's weight-gradient unit target (engine.emit_wgrad)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip import geometry as G
from nirgan_hip.options import OPT
from nirgan_hip.trainer import Pix2PixTrainer
DEV = "cuda:0"


def parse(spec):
    out = {}
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        out[k] = v
    return out


def build(opts):
    OPT.reset()
    cus = G.CUS
    for k, v in opts.items():
        if k == "X3_WGRAD_TARGET":
            G.CUS = int(v)
            continue
        cur = getattr(OPT, k)
        setattr(OPT, k, (v.lower() in ("1", "true", "on")) if isinstance(cur, bool) else type(cur)(v))
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    tr = Pix2PixTrainer(netG, netD, n_blocks=6)
    g = torch.Generator().manual_seed(1)
    rgb = (0.02 + 0.58 * torch.rand(16, 3, 256, 256, generator=g)).to(DEV)
    nir = (0.05 + 0.75 * torch.rand(16, 1, 256, 256, generator=g)).to(DEV)
    for _ in range(3):
        tr.step(rgb, nir)
    G.CUS = cus
    OPT.reset()
    return tr, rgb, nir


def block(t, steps=20):
    tr, rgb, nir = t
    tr.step(rgb, nir)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps):
        tr.step(rgb, nir)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps


A = parse(sys.argv[1] if len(sys.argv) > 1 else "")
B = parse(sys.argv[2] if len(sys.argv) > 2 else "")
ta, tb = build(A), build(B)
xa, xb = [], []
for _ in range(6):
    xa.append(block(ta))
    xb.append(block(tb))
ma, mb = statistics.median(xa), statistics.median(xb)
print(f"A {A or 'defaults'}: median {ma:.3f} ms/step = {16e3 / ma:.1f} tiles/s   ({' '.join(f'{t:.3f}' for t in xa)})")
print(f"B {B or 'defaults'}: median {mb:.3f} ms/step = {16e3 / mb:.1f} tiles/s   ({' '.join(f'{t:.3f}' for t in xb)})   A vs B {(mb / ma - 1) * 100:+.2f} %")
