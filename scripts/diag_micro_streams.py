#!/usr/bin/env python3
"""The two-part step against the single-part step (bs 4 @128, first step's prediction): per setting of the split tiles, with the
two parts on two HIP streams and with both parts on the launch stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from model import networks
from nirgan_hip import trainer as T
from nirgan_hip.options import OPT
DEV = "cuda:0"
g = torch.Generator().manual_seed(21)
rgb = (0.02 + 0.58 * torch.rand(4, 3, 128, 128, generator=g)).to(DEV)
nir = (0.05 + 0.75 * torch.rand(4, 1, 128, 128, generator=g)).to(DEV)


def run(micro, one_stream=False, repeat=1):
    torch.manual_seed(0)
    G = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    D = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    tr = T.Pix2PixTrainer(G, D, n_blocks=6, lr=0.0, micro_batches=micro)
    tr._prepare(4, 128, 128)
    if one_stream:
        tr._state.streams = [None] * len(tr._state.streams)
    out = []
    for _ in range(repeat):
        tr.step(rgb, nir)
        torch.cuda.synchronize()
        out.append((tr.pred.clone(), tr.flatG.grad.clone(), tr.flatD.grad.clone()))
    return out


for conf in ({"split3": False}, {"split3": True, "split3_wino": False}, {"split3": True, "split3_wino": True}):
    OPT.reset()
    for k, v in conf.items():
        setattr(OPT, k, v)
    ref = run(1)[0]
    for one_stream in (True, False):
        outs = run(2, one_stream, repeat=4)
        for i, o in enumerate(outs):
            e = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(o, ref)]
            where = ""
            if e[0] > 0:
                d = (o[0] - ref[0]).abs()
                bad = (d > 0.25 * d.max()).nonzero()
                where = f" pred differs in tiles {sorted(set(bad[:, 0].tolist()))} rows {bad[:, 2].min().item()}..{bad[:, 2].max().item()} cols {bad[:, 3].min().item()}..{bad[:, 3].max().item()} ({len(bad)} px)"
            print(conf, "one stream" if one_stream else "two streams", f"step {i}: pred {e[0]:.3e} gG {e[1]:.3e} gD {e[2]:.3e}{where}", flush=True)
