#!/usr/bin/env python3
"""Experiment: is the generator forward launch-bound at small batches, and does a hipGraph of the forward plan help?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.nets import GeneratorEngine
from nirgan_hip.flat import FlatParams

dev = "cuda:0"
torch.manual_seed(0)
net = networks.define_G(3, 1, 64, "resnet_9blocks", "instance", False, "normal", 0.02).to(dev)
flat = FlatParams(net)
for B, S in ((1, 256), (1, 512), (4, 256), (16, 256)):
    eng = GeneratorEngine(flat.param_views(), None, 9, B, S, S, data_pad=10, need_backward=False)
    x = torch.rand(B, 3, S, S, device=dev)
    for _ in range(3):
        eng.forward(x, version=flat.version)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(x, version=flat.version)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.fwd.run()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        eng.fwd.run()
    ref = eng.pred.clone()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.rgb_in.copy_(x)
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n
    ok = torch.equal(ref, eng.pred)
    print(f"B={B} {S}x{S} (+pad 10): eager {eager * 1e3:.3f} ms  graph {graph * 1e3:.3f} ms  ({len(eng.fwd.ops)} ops, same output {ok})")
