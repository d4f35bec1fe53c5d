"""ctypes binding of libnirgan_hip.so (the C ABI declared in include/nirgan_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  Importing this
module fails loudly when it is missing: there is no CPU or PyTorch fallback for the compute
path.  ``set_backend`` is a test seam only (tests/ install a numpy emulator of the C ABI to
exercise the host logic without a GPU); product code never calls it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libnirgan_hip.so")

MAX_TAPS = 16
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
BORDER_KEEP, BORDER_REFLECT = 0, 1
OP_CONV, OP_WGRAD, OP_IN_FWD, OP_IN_BWD = 1, 2, 3, 4

fp = C.c_void_p  # device pointers travel as integers
i32, i64, f32 = C.c_int, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("inp", fp), ("in_elems", i64), ("in_hp", i32), ("in_wp", i32), ("in_cs", i32),
                ("run", i32), ("in_stride", i32), ("in_oh", i32), ("in_ow", i32), ("ntaps", i32),
                ("tap_dh", i32 * MAX_TAPS), ("tap_dw", i32 * MAX_TAPS),
                ("w", fp), ("w_elems", i64), ("bias", fp),
                ("out", fp), ("out_elems", i64), ("out_hp", i32), ("out_wp", i32), ("out_cs", i32),
                ("out_stride", i32), ("out_oh", i32), ("out_ow", i32),
                ("B", i32), ("OH", i32), ("OW", i32), ("N", i32), ("zero_page", fp),
                ("ksplit", i32), ("split_ws", fp), ("split_ws_elems", i64), ("precision", i32), ("w_bf16", i32), ("in_bf16", i32),
                ("stats_ws", fp), ("stats_ws_elems", i64), ("stats_chunk0", i32), ("stats_chunks", i32),
                ("fuse_y", fp), ("fuse_mean", fp), ("fuse_rstd", fp), ("fuse_h", i32), ("fuse_w", i32), ("fuse_oh", i32), ("fuse_ow", i32),
                ("fuse_act", i32), ("fuse_slope", f32), ("fuse_part", fp), ("fuse_part_elems", i64), ("fuse_chunk0", i32), ("fuse_chunks", i32),
                ("out_bf16", i32), ("fuse_y_bf16", i32), ("algo", i32), ("w_x3", fp), ("w_x3_plane", i64), ("out_span", i32)]


class WgradDesc(C.Structure):
    _fields_ = [("p", fp), ("p_elems", i64), ("p_hp", i32), ("p_wp", i32), ("p_cs", i32), ("p_oh", i32), ("p_ow", i32),
                ("q", fp), ("q_elems", i64), ("q_hp", i32), ("q_wp", i32), ("q_cs", i32), ("q_stride", i32),
                ("q_oh", i32), ("q_ow", i32), ("run", i32), ("ntaps", i32),
                ("tap_dh", i32 * MAX_TAPS), ("tap_dw", i32 * MAX_TAPS),
                ("B", i32), ("OH", i32), ("OW", i32), ("N", i32),
                ("slabs", fp), ("slab_elems", i64), ("nsplit", i32), ("rows_per_split", i32), ("zero_page", fp),
                ("precision", i32), ("nplanes", i32), ("p_plane", i64), ("q_plane", i64), ("pq_bf16", i32), ("algo", i32)]


class InFwdDesc(C.Structure):
    _fields_ = [("y", fp), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("norm", i32), ("eps", f32),
                ("mean", fp), ("rstd", fp), ("act", i32), ("slope", f32),
                ("residual", fp), ("r_hp", i32), ("r_wp", i32), ("r_pad", i32),
                ("out", fp), ("o_hp", i32), ("o_wp", i32), ("o_pad", i32), ("border", i32),
                ("ws", fp), ("ws_elems", i64), ("out_bf16", fp), ("stats_chunks", i32), ("stats_shift", fp), ("y_bf16", i32)]


class InBwdDesc(C.Structure):
    _fields_ = [("g", fp), ("g_hp", i32), ("g_wp", i32), ("g_pad", i32), ("g_fold", i32), ("g2", fp),
                ("a", fp), ("a_hp", i32), ("a_wp", i32), ("a_pad", i32), ("act", i32), ("slope", f32),
                ("y", fp), ("mean", fp), ("rstd", fp), ("norm", i32),
                ("B", i32), ("H", i32), ("W", i32), ("C", i32),
                ("dy", fp), ("d_hp", i32), ("d_wp", i32), ("d_pad", i32),
                ("gsum_out", fp), ("dbias", fp), ("ws", fp), ("ws_elems", i64), ("dy_bf16", fp), ("sums_chunks", i32), ("y_bf16", i32), ("g_bf16", i32)]


class ChanDgradDesc(C.Structure):
    _fields_ = [("dy", fp), ("dy_hp", i32), ("dy_wp", i32), ("dy_pad", i32), ("C", i32),
                ("w", fp), ("cin", i32), ("k", i32), ("stride", i32), ("pad", i32), ("channel", i32),
                ("B", i32), ("H", i32), ("W", i32), ("out", fp)]


class TapGatherDesc(C.Structure):
    _fields_ = [("q", fp), ("q_hp", i32), ("q_wp", i32), ("q_cs", i32), ("ntaps", i32),
                ("tap_dh", i32 * 64), ("tap_dw", i32 * 64), ("bias", fp), ("act", i32),
                ("B", i32), ("OH", i32), ("OW", i32), ("crop", i32), ("dst", fp)]


class TapScatterDesc(C.Structure):
    _fields_ = [("dout", fp), ("out", fp), ("act", i32), ("B", i32), ("OH", i32), ("OW", i32), ("crop", i32),
                ("ntaps", i32), ("tap_dh", i32 * 64), ("tap_dw", i32 * 64),
                ("dq", fp), ("q_hp", i32), ("q_wp", i32), ("q_cs", i32), ("dbias", fp)]


PIX_LOSS_WS_ELEMS = 8192        # include/nirgan_hip.h: NIRGAN_PIX_LOSS_WS_ELEMS


class PixLossDesc(C.Structure):
    _fields_ = [("rgb", fp), ("nir", fp), ("pred", fp), ("B", i32), ("H", i32), ("W", i32),
                ("w_l1", f32), ("w_ndvi", f32), ("w_ndwi", f32), ("w_gndvi", f32), ("w_savi", f32),
                ("w_msavi", f32), ("w_evi", f32), ("criterion", i32), ("log_all", i32),
                ("extra", fp), ("extra_cs", i32), ("extra_c", i32), ("extra_scale", f32),
                ("sums", fp), ("grad_pred", fp), ("ws", fp), ("ws_elems", i64)]


class InjectFwdDesc(C.Structure):
    _fields_ = [("z", fp), ("e", fp), ("scale", fp), ("style", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32),
                ("out", fp), ("o_hp", i32), ("o_wp", i32), ("o_pad", i32)]


class InjectBwdDesc(C.Structure):
    _fields_ = [("g", fp), ("a", fp), ("a_hp", i32), ("a_wp", i32), ("a_pad", i32), ("z", fp), ("e", fp),
                ("scale", fp), ("style", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32),
                ("dz", fp), ("de", fp), ("dscale", fp), ("ws", fp), ("ws_elems", i64)]


class MetricsDesc(C.Structure):
    _fields_ = [("pred", fp), ("target", fp), ("planes", i32), ("H", i32), ("W", i32),
                ("window", i32), ("sigma", f32), ("max_val", f32), ("eps", f32),
                ("ws", fp), ("ws_elems", i64), ("means", fp)]


class SsimLossDesc(C.Structure):
    _fields_ = [("pred", fp), ("target", fp), ("planes", i32), ("H", i32), ("W", i32), ("window", i32), ("sigma", f32), ("max_val", f32),
                ("eps", f32), ("weight", f32), ("ws", fp), ("ws_elems", i64), ("loss", fp), ("value", fp), ("grad_pred", fp)]


class EmdLossDesc(C.Structure):
    _fields_ = [("pred", fp), ("target", fp), ("B", i32), ("N", i64), ("weight", f32), ("ws", fp), ("ws_bytes", i64),
                ("loss", fp), ("value", fp), ("grad_pred", fp)]


class LocEncDesc(C.Structure):
    _fields_ = [("lonlat", fp), ("B", i32), ("L", i32), ("sh_norm", fp), ("nlayers", i32),
                ("weights", C.POINTER(fp)), ("biases", C.POINTER(fp)), ("dims", C.POINTER(i32)),
                ("w0", C.POINTER(C.c_double)), ("out", fp), ("features", fp)]


class HistMatchDesc(C.Structure):
    _fields_ = [("image", fp), ("reference", fp), ("B", i32), ("N", i32), ("ws", fp), ("ws_bytes", i64), ("out", fp)]


class WinoDyDesc(C.Structure):
    _fields_ = [("dy", fp), ("dy_hp", i32), ("dy_wp", i32), ("dy_pad", i32), ("B", i32), ("H", i32), ("W", i32), ("K", i32),
                ("Yt", fp), ("Yt_elems", i64), ("r", i32)]


class Wino6Desc(C.Structure):
    _fields_ = [("x", fp), ("x_hp", i32), ("x_wp", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("K", i32),
                ("U", fp), ("bias", fp), ("V", fp), ("V_elems", i64), ("M", fp), ("M_elems", i64), ("y", fp), ("zero_page", fp), ("r", i32), ("stats_ws", fp), ("stats_ws_elems", i64),
                ("fuse_y", fp), ("fuse_mean", fp), ("fuse_rstd", fp), ("fuse_g2", fp), ("fuse_gz", fp), ("fuse_part", fp), ("fuse_part_elems", i64),
                ("fuse_act", i32), ("fuse_slope", f32), ("algo", i32), ("U3", fp)]


W6_ONE_TILE, W6_PERSIST16, W6_DIRECT_TILE, W6_TILE256, W6_X3_R4 = 1, 2, 3, 4, 5      # nirgan_wino6_desc.algo
W6_PATCH_PER_THREAD, W6_PATCH_PER_LANES = 16, 17                                 # nirgan_wino6_desc.algo for the input transforms (A/B)
WGRAD_ONE_UNIT, WGRAD_TILE128, WGRAD_RING10 = 1, 2, 3      # nirgan_wgrad_desc.algo
CONV_TILE128, CONV_TILE256, CONV_X3_BN64, CONV_X3_R4 = 1, 2, 3, 4            # nirgan_conv_desc.algo


class EndConvDesc(C.Structure):
    _fields_ = [("x", fp), ("x_hp", i32), ("x_wp", i32), ("B", i32), ("OH", i32), ("OW", i32), ("crop", i32), ("C", i32), ("k", i32),
                ("w", fp), ("bias", fp), ("act", i32), ("out", fp), ("dout", fp), ("dz", fp), ("dz_elems", i64), ("gx", fp),
                ("gw", fp), ("gbias", fp), ("ws", fp), ("ws_elems", i64)]


class PlanEntry(C.Structure):
    _fields_ = [("op", i32), ("desc", fp)]


# name -> (restype, argtypes); every symbol include/nirgan_hip.h declares
PROTOTYPES = {
    "nirgan_version": (i32, []),
    "nirgan_last_error": (C.c_char_p, []),
    "nirgan_conv_igemm": (i32, [C.POINTER(ConvDesc), fp]),
    "nirgan_wgrad_igemm": (i32, [C.POINTER(WgradDesc), fp]),
    "nirgan_conv_igemm_group": (i32, [C.POINTER(C.POINTER(ConvDesc)), i32, fp]),
    "nirgan_conv_wgrad_pair": (i32, [C.POINTER(ConvDesc), C.POINTER(WgradDesc), fp]),
    "nirgan_conv_kernel_name": (C.c_char_p, [C.POINTER(ConvDesc)]),
    "nirgan_wgrad_kernel_name": (C.c_char_p, [C.POINTER(WgradDesc)]),
    "nirgan_conv_wgrad_pair_kernel_name": (C.c_char_p, [C.POINTER(ConvDesc), C.POINTER(WgradDesc)]),
    "nirgan_reduce_rows": (i32, [fp, i32, i32, i32, fp, fp, i64, i32, i32, fp]),
    "nirgan_reduce_rows_part": (i32, [fp, i32, i32, i32, i32, i32, fp, fp, i64, i32, i32, fp]),
    "nirgan_reduce_rows_batch": (i32, [fp, i32, i32, fp]),
    "nirgan_pack_rows": (i32, [fp, i64, i32, fp, fp, i32, i32, fp]),
    "nirgan_pack_rows_bf16": (i32, [fp, i64, i32, fp, fp, i32, i32, fp]),
    "nirgan_pack_rows_batch": (i32, [fp, i32, i32, fp]),
    "nirgan_split3": (i32, [fp, fp, i64, i64, fp]),
    "nirgan_location_encoder": (i32, [C.POINTER(LocEncDesc), fp]),
    "nirgan_tile_count": (i64, [i32, i32, i32, i32, i32]),
    "nirgan_tile_gather": (i32, [fp, i32, i32, i32, i32, i32, i32, i32, i32, fp, fp]),
    "nirgan_tile_scatter": (i32, [fp, i32, i32, i32, i32, i32, i32, i32, i32, fp, fp]),
    "nirgan_wino6_tiles": (i64, [i32, i32, i32]),
    "nirgan_wino6_tiles_r": (i64, [i32, i32, i32, i32]),
    "nirgan_wino6_weights": (i32, [fp, i32, i32, i32, fp, fp]),
    "nirgan_wino6_weights_r": (i32, [fp, i32, i32, i32, i32, fp, fp]),
    "nirgan_wino6_weights_x3": (i32, [fp, i32, i32, i32, i32, fp, fp, fp]),
    "nirgan_wino6_weights_batch": (i32, [fp, i32, i32, fp]),
    "nirgan_wino6_input": (i32, [C.POINTER(Wino6Desc), fp]),
    "nirgan_wino6_input_norm": (i32, [C.POINTER(Wino6Desc), fp, fp, fp, i32, f32, fp]),
    "nirgan_wino6_input_dy_norm": (i32, [C.POINTER(Wino6Desc), C.POINTER(WinoDyDesc), C.POINTER(InBwdDesc), fp]),
    "nirgan_wino6_gemm": (i32, [C.POINTER(Wino6Desc), fp]),
    "nirgan_wino6_gemm_wgrad_pair": (i32, [C.POINTER(Wino6Desc), C.POINTER(WgradDesc), fp]),
    "nirgan_wino6_gemm_kernel_name": (C.c_char_p, [C.POINTER(Wino6Desc)]),
    "nirgan_wino6_pair_kernel_name": (C.c_char_p, [C.POINTER(Wino6Desc), C.POINTER(WgradDesc)]),
    "nirgan_wino6_output": (i32, [C.POINTER(Wino6Desc), fp]),
    "nirgan_wino6_conv3x3": (i32, [C.POINTER(Wino6Desc), fp]),
    "nirgan_wino6_dy": (i32, [C.POINTER(WinoDyDesc), fp]),
    "nirgan_wino6_input_dy": (i32, [C.POINTER(Wino6Desc), C.POINTER(WinoDyDesc), fp]),
    "nirgan_wino6_wgrad_finish": (i32, [fp, i32, i32, i32, fp, i32, fp]),
    "nirgan_wino6_wgrad_finish_r": (i32, [fp, i32, i32, i32, i32, fp, i32, fp]),
    "nirgan_wino6_wgrad_finish_batch": (i32, [fp, fp, i32, i32, i32, i32, i32, i32, fp]),
    "nirgan_hist_match_ws_bytes": (i64, [i32, i32]),
    "nirgan_hist_match": (i32, [C.POINTER(HistMatchDesc), fp]),
    "nirgan_image_metrics_ws_elems": (i64, [i32, i32, i32]),
    "nirgan_image_metrics": (i32, [C.POINTER(MetricsDesc), fp]),
    "nirgan_ssim_loss_ws_elems": (i64, [i32, i32, i32, i32]),
    "nirgan_ssim_loss": (i32, [C.POINTER(SsimLossDesc), fp]),
    "nirgan_emd_loss_ws_bytes": (i64, [i32, i64, i32]),
    "nirgan_emd_loss": (i32, [C.POINTER(EmdLossDesc), fp]),
    "nirgan_instnorm_ws_elems": (i64, [i32, i32, i32, i32]),
    "nirgan_instnorm_fwd": (i32, [C.POINTER(InFwdDesc), fp]),
    "nirgan_instnorm_bwd": (i32, [C.POINTER(InBwdDesc), fp]),
    "nirgan_nchw_to_halo": (i32, [fp, i32, i32, i32, i32, fp, i32, i32, i32, i32, i32, fp]),
    "nirgan_conv_channel_dgrad": (i32, [C.POINTER(ChanDgradDesc), fp]),
    "nirgan_tap_gather": (i32, [C.POINTER(TapGatherDesc), fp]),
    "nirgan_tap_scatter": (i32, [C.POINTER(TapScatterDesc), fp]),
    "nirgan_endconv_dz_elems": (i64, [i32, i32, i32]),
    "nirgan_endconv_ws_elems": (i64, [i32, i32, i32]),
    "nirgan_endconv_fwd": (i32, [C.POINTER(EndConvDesc), fp]),
    "nirgan_endconv_dz": (i32, [C.POINTER(EndConvDesc), fp]),
    "nirgan_endconv_dgrad": (i32, [C.POINTER(EndConvDesc), fp]),
    "nirgan_endconv_wgrad": (i32, [C.POINTER(EndConvDesc), fp]),
    "nirgan_lsgan": (i32, [fp, i64, f32, f32, fp, fp, fp]),
    "nirgan_pix_loss": (i32, [C.POINTER(PixLossDesc), fp]),
    "nirgan_adam": (i32, [fp, fp, fp, fp, i64, f32, f32, f32, f32, i32, fp]),
    "nirgan_bilinear_fwd": (i32, [fp, i32, i32, i32, fp, i32, i32, fp]),
    "nirgan_bilinear_bwd": (i32, [fp, i32, i32, i32, fp, i32, i32, fp]),
    "nirgan_inject_fwd": (i32, [C.POINTER(InjectFwdDesc), fp]),
    "nirgan_inject_bwd": (i32, [C.POINTER(InjectBwdDesc), fp]),
    "nirgan_param_scale_fwd": (i32, [fp, fp, fp, i64, fp]),
    "nirgan_param_scale_bwd": (i32, [fp, fp, fp, fp, fp, fp, i64, i64, fp]),
    "nirgan_colsum": (i32, [fp, i64, i32, fp, i32, fp]),
    "nirgan_fill": (i32, [fp, i64, f32, fp]),
    "nirgan_axpy": (i32, [fp, fp, i64, f32, fp]),
    "nirgan_run_plan": (i32, [C.POINTER(PlanEntry), i32, fp]),
}


class _CLib:
    """The real backend: the shared library, prototypes attached."""

    is_emulator = False

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build the HIP extension first (python __graft_entry__.py). "
                "The NIR-GAN MI355X path has no CPU fallback.")
        # torch first: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so).  If libnirgan_hip.so is loaded before it, the
        # loader binds our kernels to /opt/rocm's copy, torch then brings a second runtime, and launches on torch's streams fail with
        # "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process)
        import torch  # noqa: F401
        self._dll = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(self._dll, name)       # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


_real = _CLib(LIB_PATH)
_backend = _real


def check_exports() -> None:
    """Every symbol the header declares is exported (done at import; kept for build())."""
    for name in PROTOTYPES:
        getattr(_real, name)
    assert _real.nirgan_version() >= 100


def backend():
    return _backend


def set_backend(obj) -> None:
    """TEST SEAM: install an object implementing the C ABI (or None to restore the library)."""
    global _backend
    _backend = _real if obj is None else obj


def is_emulated() -> bool:
    return _backend is not _real


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = _backend.nirgan_last_error()
        if isinstance(msg, bytes):
            msg = msg.decode()
        raise RuntimeError(f"libnirgan_hip {what} failed ({rc}): {msg}")


def call(name: str, *args) -> None:
    check(getattr(_backend, name)(*args), name)
