"""Diagnostic: the fused step on a golden small net with and without the conv-epilogue statistics; per-layer mean / rstd differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "nir-gan_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np, torch
from test_gpu_nets import load, make_nets, DEV
from nirgan_hip.trainer import Pix2PixTrainer
name = sys.argv[1] if len(sys.argv) > 1 else "f1_g9_rs_pad.npz"
z = load(os.path.join(ROOT, "tests", "golden"), name)
nb = 9 if "g9" in name else 6
def run(stats):
    from nirgan_hip.options import OPT
    OPT.epilogue_stats = bool(stats)
    netG, netD = make_nets(z, nb)
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, lambda_rs=float(z["lambda_rs"]), rs_weights={"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333,
                        "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}, padding=int(z["padding"]))
    rgb, nir = torch.from_numpy(z["rgb"]).to(DEV), torch.from_numpy(z["nir"]).to(DEV)
    tr.step(rgb, nir)
    torch.cuda.synchronize()
    G = tr.G
    layers = [("L1", G.L1), ("L2", G.L2), ("L3", G.L3)] + [(f"b{i}c{j}", c) for i, (_, c1, c2) in enumerate(G.blocks) for j, c in ((1, c1), (2, c2))] + [("U1", G.U1), ("U2", G.U2)]
    out = {n: (l.stats[0].cpu().clone(), l.stats[1].cpu().clone(), l.y.t.cpu().clone()) for n, l in layers}
    ops = [n for n, _ in G.fwd.ops]
    return out, tr.G.pred.cpu().clone(), tr.flatG.grad.cpu().clone(), ops
a, pa, ga, opsa = run(True)
b, pb, gb, opsb = run(False)
print("instnorm ops with stats:", sum(1 for n, ar in zip(opsa, opsa) if n == "nirgan_instnorm_fwd"))
for n in a:
    (m1, r1, y1), (m0, r0, y0) = a[n], b[n]
    print(f"{n:6s} y diff {(y1 - y0).abs().max().item():.2e} (max {y0.abs().max().item():.2e})  mean diff {(m1 - m0).abs().max().item():.2e} (max {m0.abs().max().item():.2e})  rstd rel {((r1 - r0).abs() / r0.abs()).max().item():.2e}")
print("pred diff", (pa - pb).abs().max().item(), "grad rel", ((ga - gb).norm() / gb.norm()).item())
