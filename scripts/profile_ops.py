#!/usr/bin/env python3
"""Per-op timing of one train step (GPU box): every C-ABI call of every plan is bracketed by HIP events on the
launch stream; MFMA ops are annotated with their algorithmic FLOPs (2*M*N*K) and block counts."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from nirgan_hip.trainer import Pix2PixTrainer
if os.environ.get("NIRGAN_DIAG_LIB"):      # a diagnostic build of the library (A/B of a compile-time switch), e.g. scripts/diag/*.so
    from nirgan_hip import lib as _L
    _L.set_backend(_L._CLib(os.path.abspath(os.environ["NIRGAN_DIAG_LIB"])))
    print("library:", os.environ["NIRGAN_DIAG_LIB"])

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pad = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = "cuda:0"
torch.manual_seed(0)
netG = networks.define_G(3, 1, 64, f"resnet_{nb}blocks", "instance", False, "normal", 0.02).to(dev)
netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02).to(dev)
infer = len(sys.argv) > 5 and sys.argv[5] == "infer"      # the forward-only engine of netG.eval()(x) instead of a train step
g = torch.Generator().manual_seed(1)
rgb = (0.02 + 0.58 * torch.rand(bs, 3, 256, 256, generator=g)).to(dev)
nir = (0.05 + 0.75 * torch.rand(bs, 1, 256, 256, generator=g)).to(dev)
if infer:
    from nirgan_hip.nets import GeneratorEngine
    from nirgan_hip.flat import FlatParams
    flat = netG._flat() if hasattr(netG, "_flat") else FlatParams(netG)
    flat.ensure()
    eng = GeneratorEngine(flat.param_views(), flat.grad_views(), nb, bs, 256, 256, data_pad=pad, need_backward=False, precision=prec)

    class _Step:
        def step(self, a, b):
            eng.forward(a, version=0)
    tr = _Step()
    tr.G = eng
    plans = {"G.fwd": eng.fwd, "G.pack": eng.pack_fwd}
else:
    tr = Pix2PixTrainer(netG, netD, n_blocks=nb, padding=pad, precision=prec)
for _ in range(3):
    tr.step(rgb, nir)
if not infer:
    plans = {"G.fwd": tr.G.fwd, "G.bwd": tr.G.bwd, "D2.fwd": tr.D2.fwd, "D2.bwd": tr.D2.bwd, "D1.fwd": tr.D1.fwd, "D1.bwdP": tr.D1.bwd_pred,
             "G.pack": tr.G.pack_fwd, "G.packb": tr.G.pack_bwd}



def w6_planes(r):
    """planes of a wino6 descriptor's variant code: 0 / 3 = F(4x4,3x3): 36, 4 = F(4x4,4x4): 49, 6 = F(6x6,3x3): 64"""
    return 64 if r == 6 else (max(r, 3) + 3) ** 2


def w6_tiles(d):
    mo = 6 if d.r == 6 else 4
    return d.B * (-(-d.H // mo)) * (-(-d.W // mo))


def kname(fn, *a):
    from nirgan_hip import lib as L
    be = L.backend()
    k = getattr(be, fn)(*a) if hasattr(be, fn) else None
    k = k.decode() if k else ""
    return " [256-wide tile]" if "256" in k else ""


def describe(name, args):
    if name == "nirgan_conv_igemm":
        d = args[0]._obj
        M = d.B * d.OH * d.OW
        blocks = -(-M // 128) * (-(-d.N // 128) if d.N > 64 else 1)
        return f"conv M={M} N={d.N} K={d.ntaps}x{d.run} s{d.in_stride}/{d.out_stride} blk={blocks}" + (f" ksplit={d.ksplit}" if d.ksplit > 1 else "") + kname("nirgan_conv_kernel_name", args[0]), 2.0 * M * d.N * d.ntaps * d.run
    if name == "nirgan_conv_igemm_group":
        ds = [args[0][i].contents for i in range(args[1])]
        fl = sum(2.0 * d.B * d.OH * d.OW * d.N * d.ntaps * d.run for d in ds)
        blk = sum(-(-(d.B * d.OH * d.OW) // 128) * (-(-d.N // 128) if d.N > 64 else 1) for d in ds)
        d = ds[0]
        return f"group x{len(ds)} M={d.B * d.OH * d.OW} N={d.N} K={d.ntaps}x{d.run} s{d.in_stride}/{d.out_stride} blk={blk}", fl
    if name == "nirgan_wgrad_igemm":
        w = args[0]._obj
        M = w.B * w.OH * w.OW
        K = w.ntaps * w.run
        npl = max(w.nplanes, 1)
        blocks = (-(-w.N // 128) if w.N > 64 else 1) * (-(-K // 128)) * w.nsplit * npl
        return f"wgrad M={M} N={w.N} K={K} split={w.nsplit} planes={npl} blk={blocks}" + kname("nirgan_wgrad_kernel_name", args[0]), 2.0 * npl * M * w.N * K
    if name == "nirgan_wino6_gemm_wgrad_pair":
        d, w = args[0]._obj, args[1]._obj
        T = w6_tiles(d)
        fl = 2.0 * w6_planes(d.r) * T * d.C * d.K + 2.0 * max(w.nplanes, 1) * w.B * w.OH * w.OW * w.N * w.ntaps * w.run
        return f"wino6 pair: dgrad gemm {w6_planes(d.r)} x [T={T} x {d.C}] x [{d.K}] + wgrad {w.nplanes} planes M={w.OW} split={w.nsplit} (executed flops)", fl
    if name == "nirgan_wino6_gemm":
        d = args[0]._obj
        T = w6_tiles(d)
        return f"wino6 gemm {w6_planes(d.r)} x [T={T} x C={d.C}] x [K={d.K}] blk={w6_planes(d.r) * -(-T // 128) * -(-d.K // 128)} (executed flops)", 2.0 * w6_planes(d.r) * T * d.C * d.K
    if name in ("nirgan_wino6_input", "nirgan_wino6_input_norm", "nirgan_wino6_output", "nirgan_wino6_input_dy"):
        d = args[0]._obj
        return f"{name[7:]} B={d.B} {d.H}x{d.W} C={d.C} K={d.K}", 0.0
    if name == "nirgan_conv_wgrad_pair":
        c, w = args[0]._obj, args[1]._obj
        Mc, Mw = c.B * c.OH * c.OW, w.B * w.OH * w.OW
        cb = -(-Mc // 128) * (-(-c.N // 128))
        wb = (-(-w.N // 128)) * (-(-(w.ntaps * w.run) // 128)) * w.nsplit
        return f"pair Mc={Mc} N={c.N} K={c.ntaps*c.run} | Mw={Mw} split={w.nsplit} blk={cb}+{wb}" + kname("nirgan_conv_wgrad_pair_kernel_name", args[0], args[1]), 2.0 * Mc * c.N * c.ntaps * c.run + 2.0 * Mw * w.N * w.ntaps * w.run
    if name in ("nirgan_instnorm_fwd", "nirgan_instnorm_bwd"):
        d = args[0]._obj
        return f"{name[7:]} B={d.B} {d.H}x{d.W}x{d.C}", 0.0
    return name[7:], 0.0


for pl in plans.values():
    pl.probe_idx = {i: i for i in range(len(pl.ops))}
    pl.probe_events = []
tr.G._packed_version = tr.G._packed_bwd_version = -1   # force the pack plans once
if infer:
    eng.forward(rgb, version=1)
else:
    tr.step(rgb, nir)
torch.cuda.synchronize()
rows = []
for pname, pl in plans.items():
    for i, s, e in pl.probe_events:
        name, args = pl.ops[i]
        desc, fl = describe(name, args)
        rows.append((s.elapsed_time(e), pname, desc, fl))
    pl.probe_idx = None
tot = sum(r[0] for r in rows)
print(f"bs={bs} blocks={nb} pad={pad} precision={prec}: sum of bracketed op times {tot:.2f} ms, {len(rows)} ops")
agg = collections.OrderedDict()
for ms, pname, desc, fl in rows:
    k = desc
    a = agg.setdefault(k, [0.0, 0, 0.0])
    a[0] += ms; a[1] += 1; a[2] += fl
for k, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tf = f"{fl / (ms * 1e-3) / 1e12:7.1f} TF/s" if fl else ""
    print(f"{ms:8.3f} ms  x{n:<3d} {ms / n * 1e3:9.1f} us  {tf:14s} {k}")
