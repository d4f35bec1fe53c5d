#!/usr/bin/env python3
"""Timings of the widened rows on one MI355X: forward-only generator throughput (inference path),
image metrics, SatCLIP location encoder, histogram matching.  Event-timed, inputs resident in HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))
import torch
from model import networks
from model.satclip.location_encoder import LocationEncoder, get_neural_network, get_positional_encoding
from nirgan_hip.inference import histogram_match, predict_tiled
from utils.calculate_metrics import image_metrics_device

dev = "cuda:0"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


torch.manual_seed(0)
net = networks.define_G(3, 1, 64, "resnet_9blocks", "instance", False, "normal", 0.02).to(dev).eval()
net.data_pad = 10
x = torch.rand(16, 3, 256, 256, device=dev)
for prec in ("fp32", "bf16x3", "bf16"):
    net.precision = prec
    with torch.no_grad():
        ms = timeit(lambda: net(x))
    gf = 98.28 * (276 / 256) ** 2 * 16
    print(f"N1 inference  9-block, pad 10, bs 16 @256^2, {prec:6s}: {ms:7.3f} ms/batch  {16 / ms * 1e3:8.1f} tiles/s  {gf / ms:6.1f} TFLOP/s algorithmic")
# tiled inference of one 4096 x 4096 scene: 512-pixel tiles with 16 pixels of context per side (core 480: 9 x 9 = 81 tiles, 8 per launch)
from nirgan_hip import lib as L
net.precision = "fp32"
net.data_pad = 0
scene = torch.rand(1, 3, 4096, 4096, device=dev)
with torch.no_grad():
    ms = timeit(lambda: predict_tiled(net, scene, tile=512, margin=16, batch=8), reps=3, warm=1)
tiles = torch.empty(8, 3, 512, 512, device=dev)
pred8 = torch.rand(8, 1, 512, 512, device=dev)
outs = torch.empty(1, 1, 4096, 4096, device=dev)
st = torch.cuda.current_stream().cuda_stream
mg = timeit(lambda: L.call("nirgan_tile_gather", scene.data_ptr(), 1, 3, 4096, 4096, 512, 16, 0, 8, tiles.data_ptr(), st), reps=50)
msc = timeit(lambda: L.call("nirgan_tile_scatter", pred8.data_ptr(), 1, 1, 4096, 4096, 512, 16, 0, 8, outs.data_ptr(), st), reps=50)
print(f"N1 tiled scene 4096^2 (81 tiles of 512, margin 16, 8 per launch), 9-block fp32: {ms:8.2f} ms/scene = {4096 * 4096 / ms / 1e3:7.1f} Mpixel/s; "
      f"gather of 8 tiles {mg * 1e3:6.1f} us ({2 * tiles.numel() * 4 / mg / 1e6:6.0f} GB/s), scatter {msc * 1e3:6.1f} us")
a, b = torch.rand(16, 1, 256, 256, device=dev), torch.rand(16, 1, 256, 256, device=dev)
ms = timeit(lambda: image_metrics_device(a, b), reps=50)
print(f"N2 metrics    16 x 256^2 (L1, L2, SSIM-5): {ms * 1e3:7.1f} us  ({2 * a.numel() * 4 / ms / 1e6:6.1f} GB/s of input)")
enc = LocationEncoder(get_positional_encoding("sphericalharmonics", 10, "analytic"), get_neural_network("siren", 100, 256, 512, 2)).double().eval().to(dev)
ll = torch.stack((torch.rand(32, dtype=torch.float64) * 360 - 180, torch.rand(32, dtype=torch.float64) * 180 - 90), -1).to(dev)
ms = timeit(lambda: enc(ll), reps=50)
print(f"N3 location encoder  32 coordinates, L=10 -> 512 -> 512 -> 256, fp64: {ms * 1e3:7.1f} us")
for shape in ((16, 1, 256, 256), (2, 1, 512, 512)):
    img, ref = torch.randn(*shape, device=dev), torch.rand(*shape, device=dev)
    ms = timeit(lambda: histogram_match(img, ref), reps=10)
    print(f"N4 histogram matching {shape}: {ms:7.3f} ms  ({shape[0] / ms * 1e3:7.1f} tiles/s)")
