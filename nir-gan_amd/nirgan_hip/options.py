"""Engine options that are NOT part of the reference's API: A/B switches for measurements and for the kernel tests.

One process-wide object, read when an engine is BUILT (descriptors are static afterwards; the C library reads no environment
variables and keeps no switches of its own -- kernel choices travel in the descriptors' ``algo`` fields).  Tests and scripts set
attributes (``monkeypatch.setattr(OPT, "w6_pair", False)``); the defaults are the measured-fastest configuration.  For A/B runs from
the command line the ONE environment variable ``NIRGAN_OPTIONS`` is parsed once, at import: e.g.
``NIRGAN_OPTIONS=winograd=off,fuse_inbwd=0 python bench.py``.
"""
from __future__ import annotations

import os


class Options:
    # plane GEMMs of the Winograd layers: nirgan_wino6_desc.algo (0 = persistent workgroups on 32-k stages where they apply,
    # lib.W6_ONE_TILE / W6_PERSIST16 / W6_DIRECT_TILE)
    w6_gemm_algo: int = 0
    # weight-gradient launches of plane-matrix form: nirgan_wgrad_desc.algo (0 = persistent walk, lib.WGRAD_ONE_UNIT)
    wgrad_algo: int = 0
    # data-gradient plane GEMMs and the transform-domain weight gradient of a layer in ONE grid (nirgan_wino6_gemm_wgrad_pair)
    w6_pair: bool = True
    # Winograd for the stride-1 3x3 / 4x4 layers in exact-fp32 mode: "f6" = F(6x6,3x3) + F(4x4,4x4) (default), "f4" = F(4x4,3x3) +
    # F(4x4,4x4), "off" = the direct tiles everywhere
    winograd: str = "f6"
    # instance-norm statistics from the producing kernel's own pass (convolution epilogue / Winograd output transform) and the first
    # pass of its backward inside the kernel that produces the gradient, on maps of at least epilogue_min_pixels per sample
    epilogue_stats: bool = True
    fuse_inbwd: bool = True
    epilogue_min_pixels: int = 16384
    # bf16 operand mode: its residual trunk runs on the direct tiles (64 x 64 maps: 4096 measured 1436 -> 1453 tiles/s in round 3); 1024
    # since the tiles' bf16 epilogue moves 16 bytes per lane (round 4: the 32 x 32 maps of the 128-pixel bucket and of the PatchGAN join:
    # same-box A/B with bf16_store_min_tiles 100: 6-block +0.9 %, configs[4] mixed +2.3 %; 256 / 50 / 0 add nothing)
    epilogue_min_pixels_bf16: int = 1024
    # the second pass of a residual-block layer's instance-norm backward evaluated inside the dY Winograd transform (lane-spread kernel):
    # dY is neither written nor read (nirgan_wino6_input_dy_norm)
    fuse_dy_norm: bool = True
    # data parallel: the tail of each flat gradient goes to RCCL from inside the backward plan (two buckets per network); False = ONE blocking
    # all-reduce per network after the backward (the opt-out while multi-rank RCCL behaviour is unmeasured on hardware)
    dp_buckets: bool = True
    # bf16 operand mode: a convolution output in front of an instance norm is stored as bf16 (statistics from the fp32 accumulators)
    bf16_y: bool = True
    # bf16 operand mode: the data gradients that feed an instance-norm backward are stored as bf16 by the launch that produces them
    bf16_g: bool = True
    # ... both only for launches of at least this many output tiles (below, choose_ksplit divides K and the tiles go through a workspace;
    # the tests set 0 to run small layers through the bf16 stores)
    bf16_store_min_tiles: int = 100
    # bf16 operand mode: activations / output gradients that every reader takes from the bf16 twin are stored as bf16 only
    bf16_twin_only: bool = True
    # a ResnetBlock's first InstanceNorm + ReLU + reflect pad evaluated inside the second convolution's input transform
    fold_apply: bool = True
    # bf16 operand mode: the 256 x 256 x 64 eight-phase tiles (csrc/igemm_tile256.h) where they apply; False = the 128-row tiles (A/B)
    tile256: bool = True
    # the slab sums of a backward plan's weight gradients as one launch per flush point (nirgan_reduce_rows_batch); False = one launch per layer
    batch_reduce: bool = True
    # wanted workgroups of a stand-alone weight-gradient launch (the pixel range is split to reach it): one round of the 512 slots
    wgrad_target: int = 512
    # exact-fp32 mode: the direct convolutions and weight gradients with 32-channel runs (every stride-2 / transposed layer of both networks)
    # on the bf16 matrix pipe as THREE bf16 terms per fp32 operand and six products per fp32 product (descriptor precision 3,
    # csrc/igemm_x3.h): fp32-equivalent results (measured 0.8-1.0 x the exact tile's error against float64) at 1.6-1.8 x its rate; False =
    # the exact fp32 MFMA tiles everywhere (A/B)
    split3: bool = True
    # ... and the plane GEMMs / transform-domain weight gradients of the Winograd layers too (A/B)
    split3_wino: bool = True
    # ... and the 64-channel sub-pixel launches with short K (ConvTranspose2d(128, 64, 3, s2) and the data gradient of Conv2d(64, 128, 3, s2):
    # phases of 1 / 2 / 2 / 4 taps) as TWO problems of 128 columns -- the two phases of one output row side by side, over the union of
    # their taps (nirgan_conv_desc.out_span = 2) -- on the split tile instead of four 64-column problems on the exact fp32 tile (A/B)
    pair_phases: bool = True
    # ... and the generator's first layer, Conv2d(3, 64, 7) over the 4-channel row-packed input, with two adjacent output pixels per GEMM
    # row (128 columns, runs of 8 pixels x 4 channels = 32) on the split tile instead of 64 columns x runs of 28 on the exact fp32 tile (A/B)
    pair_pixels: bool = True
    # the split tile as ONE wave per SIMD with the activation operand fed from registers (csrc/igemm_x3r.h: conv_x3r_kernel) where it
    # applies -- N % 128 == 0, at least three K-tiles -- instead of the eight-wave tile (same bits; A/B)
    x3_r4: bool = True
    # the generator's Conv2d(64, 1, 7) + tanh as direct kernels (csrc/endconv.hip) instead of tap planes + gather
    endconv_direct: bool = True

    def reset(self):
        for k, v in vars(Options).items():
            if not k.startswith("_") and not callable(v):
                setattr(self, k, v)


    def update_from(self, text: str):
        """'name=value,name=value' with the names above; values parsed by the default's type."""
        for item in filter(None, (t.strip() for t in text.split(","))):
            k, _, v = item.partition("=")
            if not hasattr(Options, k) or k.startswith("_"):
                raise ValueError(f"NIRGAN_OPTIONS: unknown option '{k}'")
            cur = getattr(Options, k)
            setattr(self, k, (v.lower() in ("1", "true", "yes", "on")) if isinstance(cur, bool) else type(cur)(v))


OPT = Options()
if os.environ.get("NIRGAN_OPTIONS"):
    OPT.update_from(os.environ["NIRGAN_OPTIONS"])
