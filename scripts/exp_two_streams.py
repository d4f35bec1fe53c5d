#!/usr/bin/env python3
"""Experiment: do two independent half-batch train steps on two HIP streams overlap (HBM-bound instance-norm kernels of
one under the MFMA kernels of the other)?  Two trainers with bs/2 each vs one trainer with bs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = "cuda:0"


def make(b, seed):
    torch.manual_seed(seed)
    g = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(dev)
    d = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
    tr = Pix2PixTrainer(g, d, n_blocks=6, precision=prec)
    gen = torch.Generator().manual_seed(seed)
    rgb = (0.02 + 0.58 * torch.rand(b, 3, 256, 256, generator=gen)).to(dev)
    nir = (0.05 + 0.75 * torch.rand(b, 1, 256, 256, generator=gen)).to(dev)
    return tr, rgb, nir


def bench(items, streams, steps=10):
    for _ in range(3):
        for (tr, rgb, nir), s in zip(items, streams):
            with torch.cuda.stream(s):
                tr.step(rgb, nir)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (tr, rgb, nir), s in zip(items, streams):
            with torch.cuda.stream(s):
                tr.step(rgb, nir)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt


one = make(bs, 0)
dt1 = bench([one], [torch.cuda.current_stream()])
print(f"one trainer bs={bs}: {dt1 * 1e3:.2f} ms/step  {bs / dt1:.1f} tiles/s")
del one
torch.cuda.empty_cache()
two = [make(bs // 2, 1), make(bs // 2, 2)]
dts = bench(two, [torch.cuda.current_stream(), torch.cuda.current_stream()])
print(f"two trainers bs={bs // 2} same stream: {dts * 1e3:.2f} ms/pair  {bs / dts:.1f} tiles/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
dt2 = bench(two, [s1, s2])
print(f"two trainers bs={bs // 2} two streams: {dt2 * 1e3:.2f} ms/pair  {bs / dt2:.1f} tiles/s")
