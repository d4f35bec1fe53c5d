#!/usr/bin/env python3
"""Experiment for a later round: how fast does the existing tile run the GEMM part of a Winograd F(2x2,3x3) formulation
of the residual-block layer?  16 frequency planes x [16384 tiles x 256] x [256 x 256]: as ONE 1x1 'convolution' with
M = 16*16384 rows and K = 256 it costs the same MFMA work (34.4 GFLOP instead of 77.3 for the direct 3x3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from nirgan_hip import geometry as G, lib as L
from nirgan_hip.engine import Ctx, Halo, Plan, emit_conv

ctx = Ctx("cuda:0")
T = 16 * 32 * 32
M = 16 * T
x = Halo(ctx, 1, M // 512, 512, 256, 0)        # any dense [M][256] view
x.t.normal_()
w = ctx.zeros(256, 256)
w.normal_()
y = Halo(ctx, 1, M // 512, 512, 256, 0)
plan = Plan(ctx)
emit_conv(plan, ctx, x, G.Taps([0], [0], 256), w, None, y, N=256, OH=M // 512, OW=512)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    plan.run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
fl = 2.0 * M * 256 * 256
print(f"Winograd-domain GEMM (shared weights stand-in) M={M} K=256 N=256: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s executed; "
      f"direct 3x3 layer = 77.3 GFLOP -> equivalent {77.3 / ms:.1f} TFLOP/s before the two transforms (~4x the activation bytes each way)")
