#!/usr/bin/env python3
"""Stress form of diag_micro_streams.py: the two-part step on two HIP streams, many times; every step's prediction bitwise against
the first step's.  For a step that differs: which of the generator engine's tensors (allocation order ~ layer order) differ and where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from model import networks
from nirgan_hip import trainer as T
from nirgan_hip.options import OPT
DEV = "cuda:0"
STEPS = int(os.environ.get("STEPS", "1500"))
CONFS = {"off": {"split3": False}, "direct": {"split3": True, "split3_wino": False}, "all": {"split3": True, "split3_wino": True}}
g = torch.Generator().manual_seed(21)
rgb = (0.02 + 0.58 * torch.rand(4, 3, 128, 128, generator=g)).to(DEV)
nir = (0.05 + 0.75 * torch.rand(4, 1, 128, 128, generator=g)).to(DEV)


def build(micro):
    torch.manual_seed(0)
    G = networks.define_G(3, 1, 64, "resnet_6blocks", "instance", False, "normal", 0.02).to(DEV)
    D = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(DEV)
    return T.Pix2PixTrainer(G, D, n_blocks=6, lr=0.0, micro_batches=micro)


def tensors(tr):
    out = []
    for mi, m in enumerate(tr._state.micros):
        seen = set()
        for i, t in enumerate(m.G.ctx.keep):
            if isinstance(t, torch.Tensor) and t.is_floating_point() and t.numel() >= 64 and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                out.append((f"m{mi}.G.keep{i}{tuple(t.shape)}{str(t.dtype)[6:]}", t))
    return out


for tag in os.environ.get("CONFS", "all,direct,off").split(","):
    OPT.reset()
    for k, v in CONFS[tag].items():
        setattr(OPT, k, v)
    tr = build(2)
    tr.step(rgb, nir)
    torch.cuda.synchronize()
    ref = tr.pred.clone()
    ts = tensors(tr)
    snap = [t.clone() for _, t in ts]
    print(tag, "tensors followed:", len(ts), flush=True)
    nbad = 0
    for i in range(1, STEPS):
        tr.step(rgb, nir)
        torch.cuda.synchronize()
        if not torch.equal(tr.pred, ref):
            nbad += 1
            d = (tr.pred - ref).abs()
            print(f"{tag} step {i}: pred max diff {d.max().item():.3e}, tiles {sorted(set((d > 0).nonzero()[:, 0].tolist()))}, {int((d > 0).sum())} px", flush=True)
            shown = 0
            for (name, t), s in zip(ts, snap):
                if not torch.equal(t, s):
                    dd = (t.float() - s.float()).abs().reshape(-1)
                    idx = (dd > 0).nonzero().reshape(-1)
                    print(f"    {name}: {len(idx)} of {dd.numel()} elements differ, max {dd.max().item():.3e} (ref max {s.float().abs().max().item():.3e}), flat index {idx.min().item()}..{idx.max().item()}", flush=True)
                    shown += 1
                    if shown >= 12:
                        break
    print(tag, f"{STEPS} steps on two streams: {nbad} differ", flush=True)
