"""Launch plans for the generator and the PatchGAN on libnirgan_hip.

An engine owns, for one (batch, tile size) shape, every device buffer of a network (halo'd
NHWC activations, saved statistics, gradient scratch, packed weights, split-K slabs) and the
pre-built descriptor lists ("plans") for forward and backward.  Running a plan is a tight
loop of C-ABI calls on the current HIP stream: no allocation, no synchronisation.  Buffers
are sized for residency (288 GB HBM3E): nothing is recomputed or re-laid-out between forward
and backward.

Reference structure being reproduced: ResnetGenerator / ResnetBlock (model/networks.py:316-434),
ResnetGenerator_inject.forward (model/generator_inject.py:105-135), NLayerDiscriminator
(model/networks.py:539-584), Px2Px_PL.forward's pad/crop (model/pix2pix.py:88-110).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import geometry as G
from . import lib as L
from .options import OPT

IN_EPS = 1e-5


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2}


def precision_code(precision) -> int:
    """'fp32' (default, the parity path) | 'bf16' (operands rounded to bf16 on the matrix pipe, fp32 accumulate) |
    'bf16x3' (fp32 operands as two bf16 terms, three bf16 products)."""
    if precision in PRECISIONS.values():
        return int(precision)
    if precision not in PRECISIONS:
        raise NotImplementedError(f"precision [{precision}] is not implemented ('fp32', 'bf16', 'bf16x3')")
    return PRECISIONS[precision]


class Ctx:
    """Device context of an engine: allocator, zero page, stream, operand precision of the contractions."""

    def __init__(self, device, precision="fp32"):
        self.device = torch.device(device)
        self.precision = precision_code(precision)
        if self.device.type != "cuda" and not L.is_emulated():
            raise RuntimeError("nirgan_hip runs on MI355X (cuda device) only; there is no CPU path")
        if self.device.type == "cuda":
            G.CUS = torch.cuda.get_device_properties(self.device).multi_processor_count     # the split plans balance against the CUs that exist
        self.zero_page = torch.zeros(64, dtype=torch.float32, device=self.device)
        self.keep: list = []
        self.bytes = 0

    def zeros(self, *shape) -> torch.Tensor:
        t = torch.zeros(*shape, dtype=torch.float32, device=self.device)
        self.bytes += t.numel() * 4
        return t

    def i32(self, arr: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)
        self.keep.append(t)
        return t

    def stream(self):
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return None


class Halo:
    """[B][H+2p][W+2p][C] fp32 buffer; the interior starts at (p, p)."""

    def __init__(self, ctx: Ctx, B, H, W, C, pad=0, tensor: Optional[torch.Tensor] = None, twin: bool = False, bf16: bool = False):
        self.B, self.H, self.W, self.C, self.pad = B, H, W, C, pad
        self.hp, self.wp = H + 2 * pad, W + 2 * pad
        if bf16:        # stored as bf16 only (a convolution output in front of its instance norm, bf16 operand mode: ConvIN.y)
            assert tensor is None and not twin
            tensor = torch.zeros(B, self.hp, self.wp, C, dtype=torch.bfloat16, device=ctx.device)
            ctx.bytes += tensor.numel() * 2
        self.is16 = bf16
        self.t = tensor if tensor is not None else ctx.zeros(B, self.hp, self.wp, C)
        assert self.t.numel() == B * self.hp * self.wp * C
        ctx.keep.append(self.t)          # descriptors hold raw pointers: the context owns every buffer
        # bf16 operand mode: an instance-norm producer mirrors every store into a bf16 twin of the same geometry (rounded
        # once, to nearest even); convolutions read the twin.  RULE: a buffer with a twin is written by emit_in_fwd (as
        # `out`) or emit_in_bwd (as `dy`) ONLY -- they are the kernels that keep the twin in step.
        self.t16 = None
        if twin and ctx.precision == 1 and C % 8 == 0:
            self.t16 = torch.zeros(B, self.hp, self.wp, C, dtype=torch.bfloat16, device=ctx.device)
            ctx.bytes += self.t16.numel() * 2
            ctx.keep.append(self.t16)

        # bf16 operand mode: who reads the fp32 tensor?  Launches that may take the twin instead register their descriptor in `readers`
        # (emit_conv / emit_wgrad, through operand_ptr()); every other use goes through `.ptr` and pins the fp32 tensor.  A twinned buffer
        # that is never pinned and whose readers all take the twin is stored as bf16 only (drop_dead_fp32_stores).
        self.readers, self.pinned, self.writers, self.fp32_dead = [], False, [], False
        # set by drop_dead_fp32_stores: the storage decision of this buffer (fp32 + twin, twin only, fp32 only) is taken from the readers
        # seen so far, so a launch emitted afterwards must not register as a new reader (it would read a tensor nobody maintains)
        self.finalised = False

    @property
    def ptr(self):
        if self.is16:             # only the launches that know the storage rule may touch it (ptr_any)
            raise RuntimeError("this buffer holds bf16 elements: a launch site has to take it through ptr_any() and pass the *_bf16 flag")
        if self.fp32_dead:        # a launch emitted after the engine was finalised would read a tensor nobody writes any more
            raise RuntimeError("this buffer is stored as bf16 only (drop_dead_fp32_stores ran): its fp32 tensor is not maintained")
        self.pinned = True
        return self.t.data_ptr()

    def ptr_any(self):
        """address of the tensor whatever its element type (fp32, or bf16 when is16): for the launch sites that pass the *_bf16 flag"""
        if not self.is16:
            return self.ptr
        return self.t.data_ptr()

    def operand_ptr(self, reader=None):
        """fp32 address for a launch that is twin-aware: `reader` (its descriptor) says later whether it took the twin (reads_twin)."""
        if self.finalised and (self.fp32_dead or self.t16 is None or reader is None):
            # after the engine's storage decisions: only a twin-aware reader of a buffer that still has BOTH tensors may come late
            raise RuntimeError("this buffer's storage was finalised (drop_dead_fp32_stores ran): emit every launch before the engine is "
                               "finalised, or read it through the tensor that is still maintained")
        if reader is not None:
            self.readers.append(reader)
        return self.t.data_ptr()

    @property
    def elems(self):
        return self.t.numel()

    def interior(self) -> torch.Tensor:
        if self.fp32_dead:
            raise RuntimeError("this buffer is stored as bf16 only (drop_dead_fp32_stores ran): its fp32 tensor is not maintained")
        p = self.pad
        return self.t[:, p:p + self.H, p:p + self.W, :]


HOOK = "__hook__"


class Plan:
    def __init__(self, ctx: Ctx):
        self.ctx = ctx
        self.ops: list = []
        self.probe_idx = None        # bench.py: {op index: kernel label} to bracket with HIP events on the launch stream
        self.probe_events: list = []

    def add(self, name: str, *args):
        self.ops.append((name, args))

    def insert_hook(self, index: int, fn) -> None:
        """Host callback between two launches (data parallel: start a gradient bucket's all-reduce as soon as the launches
        that complete it are on the stream).  ``index`` counts C-ABI ops; earlier hooks do not shift it."""
        pos, seen = len(self.ops), 0
        for j, (n, _) in enumerate(self.ops):
            if n is not HOOK:
                if seen == index:
                    pos = j
                    break
                seen += 1
        self.ops.insert(pos, (HOOK, (fn,)))

    def extend(self, other: "Plan"):
        self.ops.extend(other.ops)

    def fuse_packs(self):
        """Replace the nirgan_pack_rows ops of this plan by ONE nirgan_pack_rows_batch launch (job table in device memory); a pack that is
        followed by its nirgan_split3 (precision 3: three bf16 planes of the packed weights) carries the planes in its job."""
        packs = [(a, n == "nirgan_pack_rows_bf16") for n, a in self.ops if n in ("nirgan_pack_rows", "nirgan_pack_rows_bf16")]
        if len(packs) < 2 or len(packs) > 256:
            return
        twins = {a[0]: (a[1], a[3]) for n, a in self.ops if n == "nirgan_split3"}       # packed buffer -> (planes, plane stride)
        rows, first = [], 0
        for (src, src_elems, stride, imap, dst, N, K), bf16 in packs:
            tw, plane = twins.pop(dst, (0, 0))
            rows.append([src, dst, imap, src_elems, N, K, stride | ((1 << 32) if bf16 else 0), first, tw, plane])
            first += N * ((K + 1023) // 1024)
        assert not twins, "a nirgan_split3 op without its pack"
        table = torch.tensor(rows, dtype=torch.int64).to(self.ctx.device)
        self.ctx.keep.append(table)
        self.ops = [(n, a) for n, a in self.ops if n not in ("nirgan_pack_rows", "nirgan_pack_rows_bf16", "nirgan_split3")]
        self.ops.append(("nirgan_pack_rows_batch", (table.data_ptr(), len(rows), first)))

    def fuse_wino6_weights(self):
        """The same for the Winograd weight transforms (csrc/wino6.hip: two per residual-block convolution and step, forward and flipped filter)."""
        names = ("nirgan_wino6_weights_r", "nirgan_wino6_weights_x3")
        jobs = [(a + (0,) if n == names[0] else a) for n, a in self.ops if n in names]
        if len(jobs) < 2 or len(jobs) > 256:
            return
        rows, first = [], 0
        for w, K, Cc, r, flip, U, U3 in jobs:
            rows.append([w, U or 0, K, Cc, flip, first, r, U3])
            first += (K * Cc + 255) // 256
        table = torch.tensor(rows, dtype=torch.int64).to(self.ctx.device)
        self.ctx.keep.append(table)
        self.ops = [(n, a) for n, a in self.ops if n not in names]
        self.ops.append(("nirgan_wino6_weights_batch", (table.data_ptr(), len(rows), first)))

    def run(self):
        be = L.backend()
        st = self.ctx.stream()
        if self.probe_idx:
            return self._run_probed(be, st)
        for name, args in self.ops:
            if name is HOOK:
                args[0]()
                continue
            rc = getattr(be, name)(*args, st)
            if rc != 0:
                L.check(rc, name)

    def _run_probed(self, be, st):
        """bench.py: the same loop with HIP events around the ops listed in probe_idx (on the launch stream)."""
        main = torch.cuda.current_stream(self.ctx.device) if self.ctx.device.type == "cuda" else None
        probes = self.probe_idx or {}
        for i, (name, args) in enumerate(self.ops):
            if name is HOOK:
                args[0]()
                continue
            if i in probes and main is not None:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(main)
                rc = getattr(be, name)(*args, st)
                e.record(main)
                self.probe_events.append((probes[i], s, e))
            else:
                rc = getattr(be, name)(*args, st)
            if rc != 0:
                L.check(rc, name)


def _set_taps(desc, dh, dw):
    assert len(dh) == len(dw) <= L.MAX_TAPS
    desc.ntaps = len(dh)
    for i, (a, b) in enumerate(zip(dh, dw)):
        desc.tap_dh[i] = a
        desc.tap_dw[i] = b


class Weights:
    """Packed copies of one parameter tensor, refreshed by a pack plan."""

    def __init__(self, ctx: Ctx):
        self.ctx = ctx
        self.cache: Dict[tuple, torch.Tensor] = {}

    def packed(self, plan: Plan, param: torch.Tensor, spec: G.PackSpec, rows_alloc: Optional[int] = None, fp32: bool = False) -> torch.Tensor:
        key = (param.data_ptr(), spec.key, fp32)
        if key in self.cache:
            return self.cache[key]
        rows = max(spec.N, rows_alloc or 0)
        imap = self.ctx.i32(spec.index_map)
        # bf16 operand mode: the packed copy is STORED as bf16 (rounded once here instead of at every fragment read) when
        # its rows stay 16-byte aligned; consumers recognise it by the tensor's dtype (emit_conv)
        # (fp32=True: a consumer that computes in fp32 in every mode -- the direct last-layer kernels)
        if self.ctx.precision == 1 and spec.run > 0 and spec.run % 8 == 0 and spec.K % 8 == 0 and not fp32:
            buf = torch.zeros(rows, spec.K, dtype=torch.bfloat16, device=self.ctx.device)
            self.ctx.bytes += buf.numel() * 2
            plan.add("nirgan_pack_rows_bf16", param.data_ptr(), param.numel(), spec.row_stride, imap.data_ptr(),
                     buf.data_ptr(), spec.N, spec.K)
        else:
            buf = self.ctx.zeros(rows, spec.K)
            plan.add("nirgan_pack_rows", param.data_ptr(), param.numel(), spec.row_stride, imap.data_ptr(),
                     buf.data_ptr(), spec.N, spec.K)
            if self.ctx.precision == 0 and OPT.split3 and spec.run > 0 and spec.run % 32 == 0 and spec.N % 64 == 0 and not fp32:
                # precision 3 (csrc/igemm_x3.h): the packed weights once more as three bf16 planes h, m, l (w = h + m + l exactly),
                # refreshed with the pack; emit_conv finds them on the tensor
                plane = rows * spec.K
                tw = torch.zeros(3 * plane, dtype=torch.bfloat16, device=self.ctx.device)
                self.ctx.bytes += tw.numel() * 2
                self.ctx.keep.append(tw)
                plan.add("nirgan_split3", buf.data_ptr(), tw.data_ptr(), spec.N * spec.K, plane)
                buf.x3 = (tw, plane)
        self.ctx.keep.append(buf)
        self.cache[key] = buf
        return buf

    def packed_pair(self, plan: Plan, param: torch.Tensor, specs: list) -> torch.Tensor:
        """Two pack specs of equal shape stacked into one [2 N][K] matrix with its three bf16 planes (precision 3): the weights of a
        paired sub-pixel problem (nirgan_conv_desc.out_span = 2) -- rows 0..N-1 the first pixel's, N..2N-1 the second's."""
        a, b = specs
        assert (a.N, a.K, a.run) == (b.N, b.K, b.run) and a.run % 32 == 0 and self.ctx.precision == 0
        key = (param.data_ptr(), "pair", a.key, b.key)
        if key in self.cache:
            return self.cache[key]
        buf = self.ctx.zeros(2 * a.N, a.K)
        plane = 2 * a.N * a.K
        tw = torch.zeros(3 * plane, dtype=torch.bfloat16, device=self.ctx.device)
        self.ctx.bytes += tw.numel() * 2
        for i, spec in enumerate(specs):
            imap = self.ctx.i32(spec.index_map)
            half = buf.data_ptr() + i * a.N * a.K * 4
            plan.add("nirgan_pack_rows", param.data_ptr(), param.numel(), spec.row_stride, imap.data_ptr(), half, spec.N, spec.K)
            plan.add("nirgan_split3", half, tw.data_ptr() + i * a.N * a.K * 2, spec.N * spec.K, plane)
        buf.x3 = (tw, plane)
        self.ctx.keep += [buf, tw]
        self.cache[key] = buf
        return buf

    def doubled(self, plan: Plan, vec: torch.Tensor) -> torch.Tensor:
        """[v, v]: the bias of a paired sub-pixel problem (one value per GEMM column), refreshed with the packs."""
        key = (vec.data_ptr(), "doubled")
        if key in self.cache:
            return self.cache[key]
        n = vec.numel()
        buf = self.ctx.zeros(2 * n)
        imap = self.ctx.i32(np.concatenate([np.arange(n), np.arange(n)]))
        plan.add("nirgan_pack_rows", vec.data_ptr(), n, 0, imap.data_ptr(), buf.data_ptr(), 1, 2 * n)
        self.ctx.keep.append(buf)
        self.cache[key] = buf
        return buf


def channels_of(d) -> int:
    """Channels per output pixel of a convolution descriptor (N, or N / 2 with out_span = 2)."""
    return d.N // max(1, d.out_span)


# ---------------------------------------------------------------------------------------------
# descriptor emitters
# ---------------------------------------------------------------------------------------------
class SplitPool:
    """Workspace for split-K partial tiles, shared by the launches of a context (they run serially)."""

    def __init__(self, ctx: "Ctx"):
        self.ctx, self.buf = ctx, None

    def get(self, n: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < n:
            self.buf = self.ctx.zeros(n)
            self.ctx.keep.append(self.buf)
        return self.buf


def choose_ksplit(tiles: int, nk: int) -> int:
    """Few output tiles (< ~0.8 per CU slot) and a long K loop: divide K so that ~512-1024 blocks exist."""
    if tiles >= 400 or nk < 32:
        return 1
    k = max(1, min(8, 1024 // max(tiles, 1), nk // 16))
    return k if k > 1 else 1


def emit_conv(plan: Plan, ctx: Ctx, inp: Halo, taps: G.Taps, w: torch.Tensor, bias, out: Halo, *, N, OH, OW,
              in_stride=1, in_oh=0, in_ow=0, out_stride=1, out_oh=0, out_ow=0, in_hw=None, allow_split=True, out_span=1,
              in_cs=None, out_hw=None, out_cs=None):
    """in_hw / in_cs, out_hw / out_cs: another VIEW of the same dense buffers (two adjacent pixels as one pixel of twice the channels)."""
    d = L.ConvDesc()
    d.inp, d.in_elems = inp.operand_ptr(d), inp.elems
    if inp.t16 is not None and w.dtype == torch.bfloat16 and taps.run % 8 == 0:
        d.inp, d.in_bf16 = inp.t16.data_ptr(), 1          # both operands as stored (bf16 twin of the producer, bf16 weights)
    d.in_hp, d.in_wp, d.in_cs = (in_hw or (inp.hp, inp.wp)) + (in_cs or inp.C,)
    d.run, d.in_stride, d.in_oh, d.in_ow = taps.run, in_stride, in_oh, in_ow
    _set_taps(d, taps.dh, taps.dw)
    d.w, d.w_elems, d.bias = w.data_ptr(), w.numel(), _ptr(bias)
    d.w_bf16 = 1 if w.dtype == torch.bfloat16 else 0
    if d.w_bf16 and taps.run % 8:
        raise ValueError(f"bf16-stored weights need run % 8 == 0 (run={taps.run})")
    d.out_bf16 = 1 if getattr(out, "is16", False) else 0
    d.out, d.out_elems = (out.ptr_any() if d.out_bf16 else out.ptr), out.elems
    if d.out_bf16:
        allow_split = False          # a bf16 output is stored by the tile's own epilogue (no split-K workspace pass)
    d.out_hp, d.out_wp, d.out_cs = (out_hw or (out.hp, out.wp)) + (out_cs or out.C,)
    d.out_stride, d.out_oh, d.out_ow = out_stride, out_oh, out_ow
    d.B, d.OH, d.OW, d.N = inp.B, OH, OW, N
    d.zero_page = ctx.zero_page.data_ptr()
    d.precision = ctx.precision
    d.algo = 0 if OPT.tile256 else L.CONV_TILE128
    x3 = getattr(w, "x3", None)
    # (N = 64: the 256 x 64 tile converts as many activation rows per MFMA as the 128-column tile does for two -- it pays from K = 512 on:
    # PatchGAN sub-pixel phases 307 -> 247 us, the generator's K = 128 .. 512 phases on the 256 x 256 maps 397 -> 508)
    if (x3 is not None and ctx.precision == 0 and OPT.split3 and taps.run % 32 == 0 and not d.out_bf16 and out.C % 4 == 0
            and (N % 128 == 0 or (N % 64 == 0 and taps.n * taps.run >= 512))):
        # exact-fp32 mode: this contraction on the bf16 pipe as three bf16 terms per operand, six products (fp32-equivalent; no split-K form)
        d.precision, d.w_x3, d.w_x3_plane = 3, x3[0].data_ptr(), x3[1]
        if OPT.x3_r4:
            d.algo = L.CONV_X3_R4        # (the library takes it where the four-wave tile applies: igemm_x3r.h::conv_x3r_ok)
        allow_split = False
    if out_span > 1:
        # two adjacent output pixels per GEMM row (the split tile only): N = 2 C columns into a dense C-channel tensor
        assert d.precision == 3 and out.C * out_span == N and (out_stride >= out_span or d.out_cs == N), "out_span needs the three-term split tile and a dense output"
        d.out_span = out_span
    ctx.keep.append(d)
    if plan is not None:
        M = inp.B * OH * OW
        tiles = -(-M // 128) * (-(-N // 128) if N > 64 else 1)
        ks = choose_ksplit(tiles, taps.n * (-(-taps.run // 32))) if (allow_split and N % 4 == 0) else 1
        if ks > 1:
            if not hasattr(ctx, "split_pool"):
                ctx.split_pool = SplitPool(ctx)
            ws = ctx.split_pool.get(ks * M * N)
            d.ksplit, d.split_ws, d.split_ws_elems = ks, ws.data_ptr(), ws.numel()
        plan.add("nirgan_conv_igemm", C.byref(d))
    return d


def emit_wgrad(plan: Plan, ctx: Ctx, p: Halo, q: Halo, taps: G.Taps, spec: G.PackSpec, grad: torch.Tensor, *,
               N, OH, OW, p_oh, p_ow, q_stride=1, q_oh=0, q_ow=0, accumulate=False, slabs_pool=None, pair_with=None):
    """pair_with: a ConvDesc (built with plan=None) launched in the same grid (nirgan_conv_wgrad_pair)."""
    K = taps.n * taps.run
    assert K == spec.K, (K, spec.K)
    tiles = (-(-N // 128) if N > 64 else 1) * (-(-K // 128))
    M = p.B * OH * OW
    target = OPT.wgrad_target         # blocks of a stand-alone launch: ONE round of the 512 slots (256 / 384 / 512 / 1024 / 2048: 765 / 759 / 789 / 782 / 771 tiles/s)
    if pair_with is not None:
        c = pair_with
        conv_blocks = -(-(c.B * c.OH * c.OW) // 128) * (-(-c.N // 128) if c.N > 64 else 1)
        total = 512 * max(1, round((conv_blocks + 1024) / 512))      # (bf16 mode, round 3: 256 / 512 / 1024 / 1536 wanted blocks: 13.46 / 13.50 / 13.45 / 13.74 ms)
        target = max(total - conv_blocks, 512)
    twins = (ctx.precision == 1 and p.t16 is not None and q.t16 is not None and N > 64 and N % 8 == 0 and taps.run % 8 == 0
             and (pair_with is None or pair_with.in_bf16))
    nsplit, rows = G.wgrad_split(M, tiles, target, 64 if twins else 32)
    # exact-fp32 mode: the three-term split tile (csrc/igemm_x3.h::wgrad_tile_x3: persistent, one workgroup per CU, units of 128 or 256 rows
    # x 128 columns x one split) where its scalar pixel walk applies -- mirrors wgrad_x3_ok
    x3 = (ctx.precision == 0 and OPT.split3 and N % 128 == 0 and taps.run % 8 == 0 and OW % 32 == 0 and M % 32 == 0
          and all(a * q.wp * q.C + b * q.C >= 0 for a, b in zip(taps.dh, taps.dw)))
    if x3:
        nsplit, rows = G.wgrad_split(M, (N // (256 if N % 256 == 0 else 128)) * (-(-K // 128)), G.CUS, 32)
    if twins and OPT.tile256 and G.wgrad256_ok(M, OH, OW, N, K, taps.run):
        # the 256-wide persistent tiles (csrc/igemm_tile256.h): splits sized so that the data-gradient and weight-gradient workgroups
        # of the fused launch finish together (the library divides the CUs with the same cost figure)
        c = pair_with
        fused = c is not None and c.in_bf16 and c.N % 256 == 0 and c.run % 64 == 0 and -(-(c.B * c.OH * c.OW) // 256) * (c.N // 256) >= 128
        nsplit, rows = (G.pair256_plan(M, (N // 256) * (K // 256), -(-(c.B * c.OH * c.OW) // 256) * (c.N // 256), c.ntaps * (c.run // 64)) if fused
                        else G.pair256_plan(M, (N // 256) * (K // 256)))
    need = nsplit * N * K
    # deferred slab sums (ctx.rr_deferred is a list while a network's backward plan is being built): the layer keeps its own slabs and all
    # of them are summed by ONE launch at the plan's next flush point (emit_deferred_reduce_rows: where a data-parallel bucket is final,
    # and at the end of the plan)
    deferred = getattr(ctx, "rr_deferred", None)
    slabs = ctx.zeros(need) if deferred is not None else (slabs_pool.get(need) if slabs_pool is not None else ctx.zeros(need))
    ctx.keep.append(slabs)
    d = L.WgradDesc()
    d.p, d.p_elems, d.p_hp, d.p_wp, d.p_cs, d.p_oh, d.p_ow = p.operand_ptr(d), p.elems, p.hp, p.wp, p.C, p_oh, p_ow
    d.q, d.q_elems, d.q_hp, d.q_wp, d.q_cs = q.operand_ptr(d), q.elems, q.hp, q.wp, q.C
    if twins:
        d.p, d.q, d.pq_bf16 = p.t16.data_ptr(), q.t16.data_ptr(), 1     # both operands from the producers' bf16 twins
    d.q_stride, d.q_oh, d.q_ow = q_stride, q_oh, q_ow
    d.run = taps.run
    _set_taps(d, taps.dh, taps.dw)
    d.B, d.OH, d.OW, d.N = p.B, OH, OW, N
    d.slabs, d.slab_elems, d.nsplit, d.rows_per_split = slabs.data_ptr(), slabs.numel(), nsplit, rows
    d.zero_page = ctx.zero_page.data_ptr()
    d.precision = 3 if x3 else ctx.precision
    if not OPT.tile256:
        d.algo = L.WGRAD_TILE128
    ctx.keep.append(d)
    imap = ctx.i32(spec.index_map)
    if pair_with is not None:
        plan.add("nirgan_conv_wgrad_pair", C.byref(pair_with), C.byref(d))
    else:
        plan.add("nirgan_wgrad_igemm", C.byref(d))
    if deferred is not None:
        # the Conv2d weight's own layout: packed column t * Cin + c <-> gradient element c * T + t -- the batch launch then writes whole runs
        T = taps.n
        conv_layout = (spec.key[:1] == ("cf",) and taps.run % 64 == 0 and T <= 16 and spec.row_stride == taps.run * T and K == T * taps.run)
        deferred.append((slabs, nsplit, N, K, imap, grad, spec.row_stride, 1 if accumulate else 0, T if conv_layout else 0))
        return d
    plan.add("nirgan_reduce_rows", slabs.data_ptr(), nsplit, N, K, imap.data_ptr(), grad.data_ptr(), grad.numel(),
             spec.row_stride, 1 if accumulate else 0)
    return d


def emit_deferred_reduce_rows(plan: Plan, ctx: Ctx, last: bool = True):
    """ONE nirgan_reduce_rows_batch for the weight gradients collected in ctx.rr_deferred since the last flush (up to 64 per launch): a
    single layer's slab sum is 12-14 us of launch latency for up to 31 MB; a dozen in one grid run at the memory rate.  `last` ends the
    collection (the end of the plan); otherwise the list stays open for the layers behind this flush point."""
    if getattr(ctx, "rr_deferred", None) is None:
        return
    items = ctx.rr_deferred
    ctx.rr_deferred = None if last else []
    for i0 in range(0, len(items), 64):
        part = items[i0:i0 + 64]
        if len(part) == 1:
            slabs, nsplit, N, K, imap, grad, stride, acc, _ = part[0]
            plan.add("nirgan_reduce_rows", slabs.data_ptr(), nsplit, N, K, imap.data_ptr(), grad.data_ptr(), grad.numel(), stride, acc)
            continue
        rows, first = [], 0
        for slabs, nsplit, N, K, imap, grad, stride, acc, T in part:
            rows.append([slabs.data_ptr(), grad.data_ptr(), imap.data_ptr(), nsplit, N, K, grad.numel(), stride | (acc << 32), first, T])
            first += N * ((K // T // 64) if T else ((K + 255) // 256))
        table = torch.tensor(rows, dtype=torch.int64).to(ctx.device)
        ctx.keep.append(table)
        plan.add("nirgan_reduce_rows_batch", table.data_ptr(), len(rows), first)


def wino_applicable(ctx: Ctx, inp: Halo, k, s, p, cout, OH, OW) -> bool:
    """Winograd forward (csrc/wino6.hip) instead of the direct tile: exact-fp32 mode, stride-1 3x3 / 4x4 with padding 1 over a halo of
    exactly 1, channel counts the plane GEMMs support.  In this network: the two convolutions of every ResnetBlock (64 % of the direct
    FLOPs) as F(6x6,3x3) -- F(4x4,3x3) with OPT.winograd = 'f4' -- and the stride-1 4x4 layer of the PatchGAN as F(4x4,4x4).
    OPT.winograd = 'off' keeps the direct tiles everywhere (A/B)."""
    return (ctx.precision == 0 and OPT.winograd != "off" and k in (3, 4) and s == 1 and p == 1 and inp.pad == 1
            and inp.C % 32 == 0 and cout % 128 == 0 and OH == inp.H + 3 - k and OW == inp.W + 3 - k and OH > 1 and OW > 1)


def wino_dgrad_applicable(ctx: Ctx, k, s, dgrad_out: Halo, cout, cin) -> bool:
    return ctx.precision == 0 and OPT.winograd != "off" and k in (3, 4) and s == 1 and cout % 32 == 0 and cin % 128 == 0


class _FullExtent:
    """A halo'd buffer seen as the dense [B][hp][wp][C] tensor it is in memory (the data gradient covers the padded extent)."""

    def __init__(self, h: Halo):
        self.B, self.hp, self.wp, self.C, self.ptr = h.B, h.hp, h.wp, h.C, h.ptr


def wino6_variant(r: int) -> int:
    """The `r` code of the wino6 descriptors for a filter size: 3 = F(4x4,3x3), 4 = F(4x4,4x4), 6 = F(6x6,3x3).  3x3 filters take
    F(6x6,3x3) -- 64 products per 36 outputs instead of 36 per 16, in the plane GEMMs AND in the bytes of the transform-domain tensors --
    unless OPT.winograd = 'f4' (A/B)."""
    return 6 if (r == 3 and OPT.winograd != "f4") else r


def _w6_geo(v: int):
    """(outputs per tile and dimension, planes) of a variant."""
    mo, filt = (6, 3) if v == 6 else (4, v)
    return mo, (mo + filt - 1) ** 2


def _w6_tiles(B, H, W, v: int = 3) -> int:
    mo = _w6_geo(v)[0]
    return B * (-(-H // mo)) * (-(-W // mo))


def emit_wino6(plan: Optional[Plan], pack: Plan, ctx: Ctx, x: Halo, weight: torch.Tensor, bias, y, *, H, W, cin, cout, flip=False,
               own_V: bool = False, x_norm=None, r: int = 3, stats_ws: Optional[torch.Tensor] = None):
    """U = G g G^T in the pack plan; input transform, (r+3)^2 plane GEMMs, output transform in `plan` (None: the caller places them).
    x: [B][H+r-1][W+r-1][cin] buffer, y: dense [B][H][W][cout].  flip: data gradient (x = dY with a zero halo of r-1, H x W = padded input)."""
    assert x.hp == H + r - 1 and x.wp == W + r - 1 and x.C == cin and y.hp == H and y.wp == W and y.C == cout, (x.hp, x.wp, H, W, y.hp, y.wp)
    B = x.B
    v = wino6_variant(r)
    T = _w6_tiles(B, H, W, v)
    NP = _w6_geo(v)[1]
    U = U3 = None
    want_x3 = ctx.precision == 0 and OPT.split3 and OPT.split3_wino and cin % 32 == 0 and cout % 64 == 0
    if want_x3 and not L.is_emulated():
        # the library decides (32-bit offsets, plane sizes: wino6.hip::w6_x3): ask it with the planes pretended present -- where it says
        # no, U is allocated and the exact-fp32 tile takes the layer instead of a launch failing with "U is required"
        probe = L.Wino6Desc()
        probe.r, probe.B, probe.H, probe.W, probe.C, probe.K = v, B, H, W, cin, cout
        probe.U3, probe.V, probe.V_elems, probe.M, probe.M_elems = ctx.zero_page.data_ptr(), ctx.zero_page.data_ptr(), NP * T * cin, ctx.zero_page.data_ptr(), NP * T * cout
        probe.zero_page = ctx.zero_page.data_ptr()
        want_x3 = (L.backend().nirgan_wino6_gemm_kernel_name(C.byref(probe)) or b"").startswith(b"conv_x3")
    if want_x3:
        # precision 3 for the plane GEMMs (csrc/igemm_x3.h): U as three bf16 planes ONLY -- the split tile reads nothing else and the
        # transform-domain weight gradient needs no U (6 instead of 10 bytes per transformed weight and step)
        U3 = torch.zeros(3 * NP * cout * cin, dtype=torch.bfloat16, device=ctx.device)
        ctx.bytes += U3.numel() * 2
        ctx.keep.append(U3)
        pack.add("nirgan_wino6_weights_x3", weight.data_ptr(), cout, cin, v, 1 if flip else 0, None, U3.data_ptr())
    else:
        U = ctx.zeros(NP * cout * cin)
        ctx.keep.append(U)
        pack.add("nirgan_wino6_weights_r", weight.data_ptr(), cout, cin, v, 1 if flip else 0, U.data_ptr())
    for name in ("wino6_pool_v", "wino6_pool_m"):
        if not hasattr(ctx, name):
            setattr(ctx, name, SplitPool(ctx))          # one layer at a time (launches run serially)
    if own_V:                                          # kept for the layer's weight gradient (same x): 2.25 x the input bytes, resident
        V = ctx.zeros(NP * T * cin)
        ctx.keep.append(V)
    else:
        V = ctx.wino6_pool_v.get(NP * T * cin)
    M = ctx.wino6_pool_m.get(NP * T * cout)
    d = L.Wino6Desc()
    d.r = v
    d.x, d.x_hp, d.x_wp = x.ptr, x.hp, x.wp
    d.B, d.H, d.W, d.C, d.K = B, H, W, cin, cout
    d.U, d.bias, d.V, d.V_elems, d.M, d.M_elems, d.y = _ptr(U), _ptr(bias), V.data_ptr(), V.numel(), M.data_ptr(), M.numel(), y.ptr
    d.zero_page = ctx.zero_page.data_ptr()
    d.algo = OPT.w6_gemm_algo
    if U3 is not None and OPT.x3_r4 and d.algo == 0:
        d.algo = L.W6_X3_R4
    d.U3 = _ptr(U3)
    if stats_ws is not None:        # the output transform leaves the instance norm's partial sums (one chunk per tile): no statistics pass over y
        assert stats_ws.numel() >= T * 4 * cout
        d.stats_ws, d.stats_ws_elems = stats_ws.data_ptr(), stats_ws.numel()
    ctx.keep.append(d)
    if plan is not None:
        if x_norm is not None:             # (y, (mean, rstd), act) of the producer: its apply pass is folded into this transform
            yh, st, act = x_norm
            assert r == 3 and yh.pad == 0 and yh.H == H and yh.W == W and yh.C == cin
            plan.add("nirgan_wino6_input_norm", C.byref(d), yh.ptr, st[0].data_ptr(), st[1].data_ptr(), act, 0.2)
        else:
            plan.add("nirgan_wino6_input", C.byref(d))
        plan.add("nirgan_wino6_gemm", C.byref(d))
        plan.add("nirgan_wino6_output", C.byref(d))
    return d


def emit_wino6_backward(plan: Plan, ctx: Ctx, dy: Halo, inp: Halo, grad: torch.Tensor, *, OH, OW, cin, cout, slabs_pool,
                        dgrad: "L.Wino6Desc", V_fwd: Optional["L.Wino6Desc"] = None, accumulate=False, r: int = 3, norm_desc=None):
    """Backward of an F(4x4,3x3) layer: dY -> (V of dY for the data gradient, Yt = A dY A^T for the weight gradient) in one pass, the
    data gradient's 36 plane GEMMs + output transform, the 36 transform-domain weight-gradient problems dU[f] = Yt[f]^T V[f] (V of the
    forward input, kept by the forward) as one weight-gradient launch, dW = G^T dU G."""
    B = inp.B
    assert inp.pad == 1 and inp.H == OH + r - 3 and inp.W == OW + r - 3 and inp.C == cin and dy.C == cout and dy.pad == r - 1 and dgrad.x == dy.ptr
    v = wino6_variant(r)
    assert dgrad.r == v
    T = _w6_tiles(B, OH, OW, v)
    NP = _w6_geo(v)[1]
    for name in ("wino6_pool_x", "wino6_pool_y", "wino6_slabs"):
        if not hasattr(ctx, name):
            setattr(ctx, name, SplitPool(ctx))
    # (round 3, measured: the weight gradient on an auxiliary stream as a one-workgroup-per-CU launch under the HBM-bound chain that
    # follows the data gradient overlaps zero-sum -- profiles/r03_side_stream_1wg_per_cu_experiment.txt; the option is gone)
    Yt = ctx.wino6_pool_y.get(NP * T * cout)
    if V_fwd is not None:
        slabs_pool = ctx.wino6_slabs
    vin = None
    if V_fwd is not None:
        V_ptr, V_elems = V_fwd.V, V_fwd.V_elems
    else:
        V = ctx.wino6_pool_x.get(NP * T * cin)
        vin = L.Wino6Desc()
        vin.r = v
        vin.x, vin.x_hp, vin.x_wp, vin.B, vin.H, vin.W, vin.C, vin.K = inp.ptr, inp.hp, inp.wp, B, OH, OW, cin, cout
        vin.V, vin.V_elems = V.data_ptr(), V.numel()
        V_ptr, V_elems = V.data_ptr(), V.numel()
    ydesc = L.WinoDyDesc()
    ydesc.dy, ydesc.dy_hp, ydesc.dy_wp, ydesc.dy_pad = dy.ptr, dy.hp, dy.wp, dy.pad
    ydesc.B, ydesc.H, ydesc.W, ydesc.K = B, OH, OW, cout
    ydesc.Yt, ydesc.Yt_elems, ydesc.r = Yt.data_ptr(), Yt.numel(), v
    tiles = (-(-cout // 128)) * (-(-cin // 128)) * NP
    # 256 output channels: the pair launch is 512 persistent workgroups that walk the weight-gradient units before their GEMM tiles
    # (csrc/wino6.hip::wino6_pair16p_kernel) -- one unit each (two splits per output tile) instead of the two rounds of short blocks
    # the one-tile-per-workgroup launch wants: half the slab traffic, 743 -> 750 tiles/s
    persistent = cout == 256 and OPT.w6_gemm_algo == 0 and OPT.wgrad_algo == 0 and OPT.w6_pair
    nsplit, rows = G.wgrad_split(T, tiles, 512 if persistent else 1024)
    # precision 3 (the layer's plane GEMMs run on the split tile: dgrad.U3): the transform-domain weight gradient as wgrad_tile_x3 units
    # (256 or 128 rows x 128 columns x one split per plane), one round of the CUs where the split count allows
    x3 = bool(dgrad.U3) and cout % 128 == 0 and cin % 8 == 0
    if x3:
        units = (cout // (256 if cout % 256 == 0 else 128)) * (-(-cin // 128)) * NP
        nsplit, rows = G.wgrad_split(T, units, max(G.CUS, units), 32)
    # deferred finish (ctx.w6_deferred is a list while a network collects the layers of one trunk): the layer keeps its own slabs and the
    # inverse transforms of all of them run as ONE launch behind the trunk (emit_w6_deferred_finishes)
    deferred = getattr(ctx, "w6_deferred", None) if persistent else None
    slabs = ctx.zeros(NP * nsplit * cout * cin) if deferred is not None else slabs_pool.get(NP * nsplit * cout * cin)
    d = L.WgradDesc()
    d.p, d.p_elems, d.p_hp, d.p_wp, d.p_cs, d.p_oh, d.p_ow = Yt.data_ptr(), NP * T * cout, 1, T, cout, 0, 0
    assert V_elems >= NP * T * cin
    d.q, d.q_elems, d.q_hp, d.q_wp, d.q_cs = V_ptr, NP * T * cin, 1, T, cin
    d.q_stride, d.q_oh, d.q_ow = 1, 0, 0
    d.run = cin
    _set_taps(d, [0], [0])
    d.B, d.OH, d.OW, d.N = 1, 1, T, cout
    d.slabs, d.slab_elems, d.nsplit, d.rows_per_split = slabs.data_ptr(), slabs.numel(), nsplit, rows
    d.zero_page = ctx.zero_page.data_ptr()
    d.precision = 3 if x3 else 0
    d.nplanes, d.p_plane, d.q_plane = NP, T * cout, T * cin
    d.algo = OPT.wgrad_algo
    ctx.keep.extend([vin, ydesc, d, slabs])
    if norm_desc is not None:    # dY evaluated on the fly from the instance-norm backward's sums: its buffer is neither written nor read
        plan.add("nirgan_wino6_input_dy_norm", C.byref(dgrad), C.byref(ydesc), C.byref(norm_desc))
    else:
        plan.add("nirgan_wino6_input_dy", C.byref(dgrad), C.byref(ydesc))      # one read of dY for both transforms
    if vin is not None:
        plan.add("nirgan_wino6_input", C.byref(vin))
    if OPT.w6_pair and not x3:                     # 24.87 -> 24.64 ms per step
        plan.add("nirgan_wino6_gemm_wgrad_pair", C.byref(dgrad), C.byref(d))      # one grid: weight-gradient blocks first, GEMM blocks behind
    else:
        plan.add("nirgan_wino6_gemm", C.byref(dgrad))
        plan.add("nirgan_wgrad_igemm", C.byref(d))
    plan.add("nirgan_wino6_output", C.byref(dgrad))
    if deferred is not None:
        deferred.append((slabs, nsplit, cout, cin, v, grad, 1 if accumulate else 0))
    else:
        plan.add("nirgan_wino6_wgrad_finish_r", slabs.data_ptr(), nsplit, cout, cin, v, grad.data_ptr(), 1 if accumulate else 0)
    return d


def emit_w6_deferred_finishes(plan: Plan, ctx: Ctx):
    """One nirgan_wino6_wgrad_finish_batch per geometry for the layers collected in ctx.w6_deferred (at most 16 per launch): a single
    layer's inverse transform is 17 us of launch latency for 33 MB of slabs, twelve in one grid run at the memory rate."""
    items, ctx.w6_deferred = (getattr(ctx, "w6_deferred", None) or []), None
    groups = {}
    for slabs, nsplit, K, Cc, v, grad, acc in items:
        groups.setdefault((nsplit, K, Cc, v, acc), []).append((slabs, grad))
    for (nsplit, K, Cc, v, acc), lst in groups.items():
        for i in range(0, len(lst), 16):
            part = lst[i:i + 16]
            if len(part) == 1:
                plan.add("nirgan_wino6_wgrad_finish_r", part[0][0].data_ptr(), nsplit, K, Cc, v, part[0][1].data_ptr(), acc)
                continue
            sp = (C.c_void_p * len(part))(*[a.data_ptr() for a, _ in part])
            gp = (C.c_void_p * len(part))(*[g.data_ptr() for _, g in part])
            ctx.keep.extend([sp, gp])
            plan.add("nirgan_wino6_wgrad_finish_batch", sp, gp, len(part), nsplit, K, Cc, v, acc)


def attach_conv_stats(ctx: Ctx, descs: list, bias) -> Optional[tuple]:
    """Let the convolution launches behind `descs` (one problem, or the sub-pixel phases of a transposed convolution writing one
    output) leave the partial sums of the instance norm that follows (csrc/igemm_tiles.h, conv_tile epilogue): returns
    (chunks per sample, shift, workspace) for emit_in_fwd(pre_stats=...), or None when a problem does not qualify."""
    if not OPT.epilogue_stats or any(d.ksplit > 1 or (d.OH * d.OW) % 128 for d in descs):
        return None
    # worth it from ~16 K pixels per sample in fp32 (the 128x128 and 256x256 layers save 16 / 54 us of statistics pass each; the trunk is
    # Winograd there) and from 4 K in the bf16 operand mode (its 64x64 trunk maps run on the direct tiles: 1436 -> 1453 tiles/s); below, the pass
    # costs 2-4 us and the layer keeps it (OPT.epilogue_min_pixels: the kernel tests run small layers through it)
    # (a paired launch -- out_span = 2 -- writes two output pixels per GEMM row: counted as the pixels they are)
    if sum(d.OH * d.OW * max(1, d.out_span) for d in descs) < (OPT.epilogue_min_pixels if ctx.precision == 0 else min(OPT.epilogue_min_pixels, OPT.epilogue_min_pixels_bf16)):
        return None
    B, N = descs[0].B, channels_of(descs[0])
    recs = [d.OH * d.OW // 64 * max(1, d.out_span) for d in descs]       # (a paired problem leaves two records per 64 rows)
    total = sum(recs)
    if not hasattr(ctx, "conv_pool_stats"):
        ctx.conv_pool_stats = SplitPool(ctx)
    ws = ctx.conv_pool_stats.get(B * total * 4 * N)
    first = 0
    for d, n in zip(descs, recs):
        d.stats_ws, d.stats_ws_elems, d.stats_chunk0, d.stats_chunks = ws.data_ptr(), ws.numel(), first, total
        first += n
    return total, bias, ws


def emit_wgrad_pixel_pairs(plan: Plan, ctx: Ctx, dy: Halo, inp: Halo, grad: torch.Tensor, *, k, p, cin, cout, OH, OW, accumulate=False) -> bool:
    """Weight gradient of the row-packed first convolution on the split tile, two adjacent output pixels per GEMM row (as its forward,
    ConvIN.emit_fwd): P = dY viewed [B][OH][OW/2][2 cout], Q = the 4-channel input viewed in pixel pairs, one tap per kernel row over
    (k + 1) pixels x cs channels = a run of 32.  Slab rows are (pixel parity q, output channel); column (kh, px, c) of parity q is
    weight element (c, kh, px - q): two nirgan_reduce_rows_part calls fold the two bands into the gradient.  False: does not apply."""
    cs = inp.C
    N, run, K = 2 * cout, (k + 1) * cs, k * (k + 1) * cs
    OWp = OW // 2
    M = inp.B * OH * OWp
    if not (ctx.precision == 0 and OPT.split3 and OPT.pair_pixels and cout == 64 and run == 32 and OW % 2 == 0 and inp.wp % 2 == 0
            and dy.wp % 2 == 0 and dy.pad % 2 == 0 and (inp.pad - p) % 2 == 0 and dy.t16 is None and OWp % 32 == 0 and M % 32 == 0):
        return False
    nsplit, rows = G.wgrad_split(M, (N // 128) * (-(-K // 128)), G.CUS, 32)
    slabs = ctx.zeros(nsplit * N * K)
    ctx.keep.append(slabs)
    d = L.WgradDesc()
    d.p, d.p_elems, d.p_hp, d.p_wp, d.p_cs, d.p_oh, d.p_ow = dy.operand_ptr(d), dy.elems, dy.hp, dy.wp // 2, N, dy.pad, dy.pad // 2
    d.q, d.q_elems, d.q_hp, d.q_wp, d.q_cs = inp.operand_ptr(d), inp.elems, inp.hp, inp.wp // 2, 2 * cs
    d.q_stride, d.q_oh, d.q_ow = 1, inp.pad - p, (inp.pad - p) // 2
    d.run = run
    _set_taps(d, list(range(k)), [0] * k)
    d.B, d.OH, d.OW, d.N = inp.B, OH, OWp, N
    d.slabs, d.slab_elems, d.nsplit, d.rows_per_split = slabs.data_ptr(), slabs.numel(), nsplit, rows
    d.zero_page = ctx.zero_page.data_ptr()
    d.precision = 3
    ctx.keep.append(d)
    plan.add("nirgan_wgrad_igemm", C.byref(d))
    for q in (0, 1):
        spec = G.conv_rowpacked_pair_pack(cout, cin, k, cs, q)
        imap = ctx.i32(spec.index_map)
        plan.add("nirgan_reduce_rows_part", slabs.data_ptr(), nsplit, N, q * cout, cout, K, imap.data_ptr(), grad.data_ptr(), grad.numel(),
                 spec.row_stride, 1 if (accumulate or q) else 0)
    return True


def want_phase_pairs(ctx: Ctx, phases: list, run: int, N: int, out) -> bool:
    """A 64-channel sub-pixel launch whose four phases would not all take the split tile (taps x run < 512 somewhere: 1 / 2 / 2 / 4 taps of a
    3 x 3 kernel) runs as two paired problems of 128 columns on it (nirgan_conv_desc.out_span = 2, geometry.pair_row_phases): measured
    412 / 349 us -> see DESIGN 3.1 for ConvTranspose2d(128, 64, 3, s2) at bs 16 / the data gradient of Conv2d(64, 128, 3, s2)."""
    return (ctx.precision == 0 and OPT.split3 and OPT.pair_phases and N == 64 and run % 32 == 0 and out.C == N and out.C % 4 == 0
            and not getattr(out, "is16", False) and len(phases) == 4 and min(len(ph.dh) for ph in phases) * run < 512)


def emit_phase_pairs(eng, pack: Plan, ctx: Ctx, inp: Halo, phases: list, spec_fn, weight, bias, out: Halo, *, N, in_off, out_off):
    """The two paired problems (output-row parities) of a stride-2 gather, or None when the phases do not pair up.  spec_fn(taps_hw) ->
    PackSpec of one phase; in_off / out_off: what the caller adds to a phase's input / output origin."""
    pairs = G.pair_row_phases(phases)
    if pairs is None:
        return None
    descs = []
    for pr in pairs:
        w = eng.weights.packed_pair(pack, weight, [G.masked_pack(spec_fn, hw) for hw in pr.taps_hw])
        b2 = eng.weights.doubled(pack, bias) if bias is not None else None
        d = emit_conv(None, ctx, inp, G.Taps(pr.dh, pr.dw, inp.C), w, b2, out, N=2 * N, OH=pr.n_h, OW=pr.n_w,
                      in_oh=pr.in_oh + in_off, in_ow=pr.in_ow + in_off, out_stride=2, out_oh=pr.out_oh + out_off, out_ow=pr.out_ow + out_off,
                      out_span=2)
        descs.append(d)          # (bench.py prices the zero blocks: 9 of the 12 tap blocks of a 3 x 3 kernel's two problems are algorithmic work)
    return descs


def emit_conv_group(plan: Plan, ctx: Ctx, descs: list):
    """One launch for up to 4 conv descriptors (sub-pixel phases)."""
    for i in range(0, len(descs), 4):
        grp = descs[i:i + 4]
        if any(d.precision != grp[0].precision for d in grp):      # the three-term split tile for every phase of a launch, or for none
            for d in grp:
                if d.precision == 3:
                    d.precision = 0
        arr = (C.POINTER(L.ConvDesc) * len(grp))(*[C.pointer(d) for d in grp])
        ctx.keep.append(arr)
        plan.add("nirgan_conv_igemm_group", arr, len(grp))


class SlabPool:
    """Scratch shared by the weight-gradient launches of an engine (they run serially on one stream)."""

    DEFAULT = 20 * 1024 * 1024   # floats; covers ~1024 blocks x 128x128 partial tiles

    def __init__(self, ctx: Ctx):
        self.ctx, self.buf = ctx, None

    def get(self, n: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < n:
            floor = 0 if (self.buf is not None or L.is_emulated()) else self.DEFAULT
            self.buf = self.ctx.zeros(max(n, floor))
            self.ctx.keep.append(self.buf)
        return self.buf


def emit_in_fwd(plan: Plan, ctx: Ctx, y: Halo, out: Halo, *, norm=True, act=L.ACT_NONE, slope=0.2, residual: Optional[Halo] = None,
                border=L.BORDER_KEEP, stats=None, ws=None, stats_only=False, pre_stats=None):
    """stats_only: mean / rstd only -- the layer's single consumer normalises on the fly (nirgan_wino_input_norm), `out` stays unwritten.
    pre_stats = (chunks per sample, shift tensor or None): the producer of y left the partial sums in `ws` (nirgan_wino6_output)."""
    assert y.pad == 0 and out.H == y.H and out.W == y.W and out.C == y.C
    assert not stats_only or (norm and residual is None)
    d = L.InFwdDesc()
    d.y, d.B, d.H, d.W, d.C = y.ptr_any(), y.B, y.H, y.W, y.C
    d.y_bf16 = 1 if y.is16 else 0
    d.norm, d.eps = (1 if norm else 0), IN_EPS
    if norm:
        d.mean, d.rstd = stats[0].data_ptr(), stats[1].data_ptr()
        d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
        if pre_stats is not None:
            d.stats_chunks, d.stats_shift = pre_stats[0], _ptr(pre_stats[1])
    d.act, d.slope = act, slope
    if residual is not None:
        d.residual, d.r_hp, d.r_wp, d.r_pad = residual.ptr, residual.hp, residual.wp, residual.pad
    if not stats_only:
        d.out, d.o_hp, d.o_wp, d.o_pad, d.border = out.operand_ptr(), out.hp, out.wp, out.pad, border
        if out.t16 is not None:
            d.out_bf16 = out.t16.data_ptr()
            out.writers.append(d)     # the fp32 store is dropped when every reader takes the twin (drop_dead_fp32_stores)
    ctx.keep.append(d)
    plan.add("nirgan_instnorm_fwd", C.byref(d))
    return d


def reads_twin(desc) -> bool:
    """Does this launch descriptor read its activation operand(s) from the producers' bf16 twins?"""
    if isinstance(desc, L.ConvDesc):
        return bool(desc.in_bf16)
    if isinstance(desc, L.WgradDesc):
        return bool(desc.pq_bf16)
    return False


def one_pass_tiles(M: int, N: int) -> bool:
    """Does a convolution launch of M output pixels x N channels store its tiles itself (choose_ksplit: no split-K from 400 tiles on)?
    Only such a launch can store bf16 (out_bf16)."""
    return -(-M // 128) * (-(-N // 128) if N > 64 else 1) >= OPT.bf16_store_min_tiles


def grad_halo(ctx: Ctx, B, H, W, Cc, pad, phases: int = 1) -> Halo:
    """The buffer a data-gradient launch writes and the consumer layer's instance-norm backward reads (with the reflect fold when pad > 0).
    bf16 operand mode: stored as bf16 by the launch's epilogue (out_bf16) -- the gradient is rounded once more before it meets the next
    contraction anyway (dy's twin); the sums of the fused first pass still come from the fp32 accumulators.  phases = 4 when the producer is
    the four sub-pixel problems of a stride-2 layer.  Small problems (split-K) and OPT.bf16_g = False keep fp32."""
    is16 = ctx.precision == 1 and OPT.bf16_g and Cc % 4 == 0 and one_pass_tiles(B * H * W // phases, Cc)
    return Halo(ctx, B, H, W, Cc, pad, bf16=is16)


def drop_dead_fp32_stores(buffers) -> int:
    """bf16 operand mode: a twinned buffer whose fp32 tensor nobody reads -- never pinned through `.ptr`, every registered reader takes
    the twin (reads_twin) -- is stored as bf16 only: its instance-norm writers get out / dy = NULL (168 -> 101 MB per forward launch of a
    64 x 64 x 256 map at bs 16, 235 -> 168 MB per backward launch).  Called once per engine, after all of its plans are built."""
    n = 0
    buffers = list(buffers)
    for h in buffers:
        h.finalised = True
    if not OPT.bf16_twin_only:
        return n
    for h in buffers:
        if h.t16 is not None and h.writers and not any(reads_twin(r) for r in h.readers):
            # the other way round: nobody reads the TWIN (the last up-convolution's output feeds the fp32 direct kernels of the last layer,
            # the first layer's dY an N = 64 weight gradient that takes fp32): the mirrored store goes
            for w in h.writers:
                if isinstance(w, L.InFwdDesc):
                    w.out_bf16 = None
                else:
                    w.dy_bf16 = None
            h.t16 = None              # nobody maintains the twin any more: a later emit_conv / emit_wgrad must not pick it up
            n += 1
            continue
        if h.t16 is None or h.pinned or not h.readers or not h.writers or not all(reads_twin(r) for r in h.readers):
            continue
        if not all(isinstance(w, L.InFwdDesc) or (isinstance(w, L.InBwdDesc) and w.norm) for w in h.writers):
            continue                  # (a backward launch without norm writes dy in its first pass: keeps fp32)
        for w in h.writers:
            if isinstance(w, L.InFwdDesc):
                w.out = None
            else:
                w.dy = None
            n += 1
        h.fp32_dead = True
    return n


def emit_in_bwd(plan: Plan, ctx: Ctx, *, g: Optional[Halo], g_fold=False, g2: Optional[Halo] = None, a: Optional[Halo] = None,
                act=L.ACT_NONE, slope=0.2, y: Optional[Halo] = None, stats=None, norm=True, dy: Halo, gsum: Optional[Halo] = None,
                dbias: Optional[torch.Tensor] = None, ws=None, shape=None, sums_only: bool = False, pre_sums: int = 0):
    """sums_only (with norm): the two reduction passes only; dy is then evaluated on the fly by the consumer (nirgan_wino6_input_dy_norm)
    and its buffer is neither written nor read.  pre_sums = chunks per sample: the producer of g (the data gradient's output transform in
    its fused mode) already left the folded gradient in `gsum` and the first pass's partial sums in `ws`."""
    B, H, W, Cc = shape
    d = L.InBwdDesc()
    if g is not None:
        assert g.H == H and g.W == W and g.C == Cc
        d.g, d.g_hp, d.g_wp, d.g_pad, d.g_fold = g.ptr_any(), g.hp, g.wp, g.pad, (1 if g_fold else 0)
        d.g_bf16 = 1 if g.is16 else 0
    if g2 is not None:
        assert g2.pad == 0
        d.g2 = g2.ptr
    # (`a`, the activated output, is not read by any kernel: the mask is the sign of z recomputed from y; the descriptor field stays 0)
    d.act, d.slope = act, slope
    d.norm = 1 if norm else 0
    if y is not None:
        d.y, d.y_bf16 = y.ptr_any(), (1 if y.is16 else 0)
    if norm:
        d.mean, d.rstd = stats[0].data_ptr(), stats[1].data_ptr()
    if norm or dbias is not None:       # partial sums of the two reduction passes / of the live bias gradient (one row per block)
        d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    d.B, d.H, d.W, d.C = B, H, W, Cc
    assert not sums_only or norm
    d.dy, d.d_hp, d.d_wp, d.d_pad = (None if sums_only else dy.operand_ptr()), dy.hp, dy.wp, dy.pad
    if dy.t16 is not None and not sums_only:
        d.dy_bf16 = dy.t16.data_ptr()
        dy.writers.append(d)
    if gsum is not None:
        d.gsum_out = gsum.ptr
    if pre_sums:
        assert norm and (gsum is not None or (g is not None and not g_fold and g2 is None))
        d.sums_chunks = pre_sums
    if dbias is not None:
        d.dbias = dbias.data_ptr()
    ctx.keep.append(d)
    plan.add("nirgan_instnorm_bwd", C.byref(d))
    return d


class _Scratch:
    """Workspace for instance-norm partial sums, sized for the largest user."""

    def __init__(self, ctx: Ctx):
        self.ctx, self.n = ctx, 0
        self.t: Optional[torch.Tensor] = None

    def want(self, B, H, W, Cc):
        self.n = max(self.n, int(L.backend().nirgan_instnorm_ws_elems(B, H, W, Cc)))

    def get(self) -> torch.Tensor:
        if self.t is None or self.t.numel() < self.n:
            self.t = self.ctx.zeros(max(self.n, 4))
            self.ctx.keep.append(self.t)      # descriptors emitted against a smaller, earlier workspace keep their (still valid) pointer
        return self.t


# ---------------------------------------------------------------------------------------------
# layer bundles: each keeps its buffers and emits forward / backward ops
# ---------------------------------------------------------------------------------------------
class ConvIN:
    """conv (Conv2d / ConvTranspose2d / row-packed first conv) -> [InstanceNorm] -> act -> halo'd output.

    kind: 'conv' (Conv2d k,s,p over an NHWC halo buffer), 'rowpacked' (Conv2d over a 4-channel
    buffer, one tap per kernel row), 'convT' (ConvTranspose2d k3 s2 p1 op1).
    """

    def __init__(self, eng, name, kind, inp: Halo, weight, bias, *, k, s, p, cout, norm=True, act=L.ACT_RELU,
                 out_pad=0, out_border=L.BORDER_KEEP, residual: Optional[Halo] = None, cin_real=None, keep_z=False):
        self.eng, self.name, self.kind = eng, name, kind
        self.inp, self.weight, self.bias = inp, weight, bias
        self.k, self.s, self.p, self.cout = k, s, p, cout
        self.norm, self.act, self.residual = norm, act, residual
        self.cin = inp.C if kind != "rowpacked" else cin_real
        ctx = eng.ctx
        if kind == "convT":
            self.OH, self.OW = inp.H * 2, inp.W * 2
        else:
            self.OH, self.OW = G.conv_out(inp.H, k, s, p), G.conv_out(inp.W, k, s, p)
        assert inp.pad >= (p if kind != "convT" else 1), (name, inp.pad, p)
        B = inp.B
        # bf16 operand mode: y is stored as bf16 when the convolution's epilogue takes the statistics from its fp32 accumulators (predicted
        # here by the rule of attach_conv_stats; emit_fwd falls back to fp32 when the launch did not qualify): every pass over y -- the
        # convolution's store, the norm's apply, both passes of its backward -- moves half the bytes
        phase_px = self.OH * self.OW // (4 if kind == "convT" else 1)
        y16 = (ctx.precision == 1 and norm and OPT.bf16_y and OPT.epilogue_stats and cout % 4 == 0 and phase_px % 128 == 0
               and self.OH * self.OW >= min(OPT.epilogue_min_pixels, OPT.epilogue_min_pixels_bf16) and one_pass_tiles(B * phase_px, cout))
        self.y = Halo(ctx, B, self.OH, self.OW, cout, 0, bf16=y16)
        self.out = Halo(ctx, B, self.OH, self.OW, cout, out_pad, twin=True)
        eng.__dict__.setdefault("twinned", []).append(self.out)          # see drop_dead_fp32_stores
        self.out_border = out_border
        self.stats = (ctx.zeros(B, cout), ctx.zeros(B, cout)) if norm else None
        eng.scratch.want(B, self.OH, self.OW, cout)
        self.keep_z = keep_z   # inject: `out` holds z (no act); the modulation produces the activated tensor

    # ---- forward
    def emit_fwd(self, plan: Plan, pack: Plan):
        eng, ctx, inp = self.eng, self.eng.ctx, self.inp
        k, s, p = self.k, self.s, self.p
        pre = None          # (chunks per sample, shift, workspace) when the convolution's last kernel leaves the instance norm's partial sums
        if self.kind == "conv" and wino_applicable(ctx, inp, k, s, p, self.cout, self.OH, self.OW):
            keep = bool(getattr(eng, "need_backward", False))     # V = B^T x B stays resident for the layer's transform-domain weight gradient
            prod = getattr(self, "producer", None)
            xn = (prod.y, prod.stats, prod.act) if prod is not None and getattr(prod, "defer_apply", False) else None
            self.wino6 = True
            sws = None
            if self.norm and OPT.epilogue_stats:
                # instance-norm statistics from the output transform's own pass (one chunk of partial sums per tile)
                T = _w6_tiles(inp.B, self.OH, self.OW, wino6_variant(k))
                if not hasattr(ctx, "wino6_pool_stats"):
                    ctx.wino6_pool_stats = SplitPool(ctx)
                sws = ctx.wino6_pool_stats.get(T * 4 * self.cout)
                pre = (T // inp.B, self.bias, sws)
            self.wino_fwd = emit_wino6(plan, pack, ctx, inp, self.weight, self.bias, self.y, H=self.OH, W=self.OW, cin=inp.C,
                                       cout=self.cout, own_V=keep, x_norm=xn, r=k, stats_ws=sws)
            self.wino_fwd_keeps_V = keep
        elif self.kind == "conv":
            taps = G.conv_fwd_taps(k, inp.C)
            w = eng.weights.packed(pack, self.weight, G.conv_fwd_pack(self.cout, inp.C, k))
            cd = emit_conv(plan, ctx, inp, taps, w, self.bias, self.y, N=self.cout, OH=self.OH, OW=self.OW,
                           in_stride=s, in_oh=inp.pad - p, in_ow=inp.pad - p)
            if self.norm:
                pre = attach_conv_stats(ctx, [cd], self.bias)
            self._y_fallback(pre, [cd])
        elif (self.kind == "rowpacked" and s == 1 and ctx.precision == 0 and OPT.split3 and OPT.pair_pixels and self.cout == 64
              and (k + 1) * inp.C == 32 and self.OW % 2 == 0 and inp.wp % 2 == 0 and (inp.pad - p) % 2 == 0 and not self.y.is16):
            # Conv2d(3, 64, 7) on the split tile: TWO adjacent output pixels per GEMM row -- 128 columns, one tap per kernel row over the
            # 8 pixels x 4 channels both windows cover (a run of 32), the buffers viewed as pixel pairs (measured: DESIGN 3.1)
            cs = inp.C
            w = eng.weights.packed_pair(pack, self.weight, [G.conv_rowpacked_pair_pack(self.cout, self.cin, k, cs, q) for q in (0, 1)])
            b2 = eng.weights.doubled(pack, self.bias) if self.bias is not None else None
            cd = emit_conv(plan, ctx, inp, G.Taps(list(range(k)), [0] * k, (k + 1) * cs), w, b2, self.y, N=2 * self.cout, OH=self.OH, OW=self.OW // 2,
                           in_oh=inp.pad - p, in_ow=(inp.pad - p) // 2, in_hw=(inp.hp, inp.wp // 2), in_cs=2 * cs,
                           out_hw=(self.y.hp, self.y.wp // 2), out_cs=2 * self.cout, out_span=2)
            if self.norm:
                pre = attach_conv_stats(ctx, [cd], self.bias)
        elif self.kind == "rowpacked":
            taps = G.conv_rowpacked_taps(k, inp.C)
            w = eng.weights.packed(pack, self.weight, G.conv_rowpacked_pack(self.cout, self.cin, k, inp.C))
            cd = emit_conv(plan, ctx, inp, taps, w, self.bias, self.y, N=self.cout, OH=self.OH, OW=self.OW,
                           in_stride=s, in_oh=inp.pad - p, in_ow=inp.pad - p)
            if self.norm:
                pre = attach_conv_stats(ctx, [cd], self.bias)
            self._y_fallback(pre, [cd])
        else:  # convT: 4 sub-pixel phases over the zero-halo-1 input, one launch
            descs = None
            phases = G.convT_fwd_phases(inp.H, inp.W, k, p)
            if want_phase_pairs(ctx, phases, inp.C, self.cout, self.y):
                descs = emit_phase_pairs(eng, pack, ctx, inp, phases, lambda hw: G.convT_fwd_pack(inp.C, self.cout, k, hw), self.weight,
                                         self.bias, self.y, N=self.cout, in_off=inp.pad - 1, out_off=0)
            for ph in (phases if descs is None else []):
                descs = descs or []
                taps = G.Taps(ph.dh, ph.dw, inp.C)
                w = eng.weights.packed(pack, self.weight, G.convT_fwd_pack(inp.C, self.cout, k, ph.taps_hw))
                descs.append(emit_conv(None, ctx, inp, taps, w, self.bias, self.y, N=self.cout, OH=ph.n_h, OW=ph.n_w,
                                       in_oh=ph.in_oh + inp.pad - 1, in_ow=ph.in_ow + inp.pad - 1,
                                       out_stride=2, out_oh=ph.out_oh, out_ow=ph.out_ow))
            if self.norm:
                pre = attach_conv_stats(ctx, descs, self.bias)
            self._y_fallback(pre, descs)
            emit_conv_group(plan, ctx, descs)
        emit_in_fwd(plan, ctx, self.y, self.out, norm=self.norm, act=(L.ACT_NONE if self.keep_z else self.act),
                    residual=self.residual, border=self.out_border, stats=self.stats, ws=(pre[2] if pre is not None else eng.scratch.get()),
                    stats_only=getattr(self, "defer_apply", False), pre_stats=(pre[:2] if pre is not None else None))

    def _y_fallback(self, pre, descs):
        """y was allocated as bf16 on the prediction that the launch leaves the statistics; it did not (split-K): back to fp32."""
        if self.y.is16 and pre is None:
            self.y = Halo(self.eng.ctx, self.inp.B, self.OH, self.OW, self.cout, 0)
            for d in descs:
                d.out, d.out_bf16 = self.y.ptr, 0

    # ---- backward: g (+g2) is the gradient wrt `out`; produces dy (zero halo) then weight / data gradients
    def alloc_bwd(self, need_dgrad: bool):
        ctx, k = self.eng.ctx, self.k
        if self.kind == "convT":
            zpad = 1
        elif self.s == 1:
            zpad = k - 1 if need_dgrad else 0
        else:
            zpad = 1 if need_dgrad else 0
        self.dy = Halo(ctx, self.inp.B, self.OH, self.OW, self.cout, zpad, twin=True)
        self.eng.__dict__.setdefault("twinned", []).append(self.dy)

    def emit_bwd(self, plan: Plan, pack: Plan, *, g: Optional[Halo], g_fold=False, g2: Optional[Halo] = None,
                 gsum: Optional[Halo] = None, gw: Optional[torch.Tensor], gb: Optional[torch.Tensor], dgrad_out: Optional[Halo],
                 mask: Optional[Halo] = None, act=None):
        eng, ctx, inp = self.eng, self.eng.ctx, self.inp
        k, s, p = self.k, self.s, self.p
        act = self.act if act is None else act
        dy = self.dy
        # Winograd backward (below).  (Round 2 measured the second pass of the instance-norm backward evaluated INSIDE the per-thread dY
        # transform at 161 us against 84 + 31 per layer -- every dY element re-derived by the 2.25 patches that contain it, at 240 VGPRs.
        # The lane-spread transform of round 3 has the registers for it: OPT.fuse_dy_norm.)
        w6_bwd = (self.kind == "conv" and s == 1 and gw is not None and dgrad_out is not None and dgrad_out.pad == p and dy.pad == k - 1
                  and wino_dgrad_applicable(ctx, k, s, dgrad_out, self.cout, inp.C) and inp.C > 64
                  and inp.pad == 1 and p == 1 and self.cout % 128 == 0)
        fuse_dy = (w6_bwd and k == 3 and self.norm and dy.t16 is None and OPT.fuse_dy_norm and wino6_variant(3) == 6
                   and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU))
        # The gradient arrives from an F(6x6,3x3) data gradient over the padded extent (reflect halo of 1 to fold): that launch's output
        # transform is switched to its fused mode -- it folds the halo in registers (lane pairs exchange half tiles so that the per-pixel
        # phase moves 16 bytes per lane), adds the skip gradient, stores the folded gradient dense and leaves the partial sums of this
        # layer's first backward pass; the halo'd buffer `g` is then never written or read.  The first pass disappears (94 -> 45 us per
        # layer) for an output transform of 67 us instead of 39: 21.16 -> 20.87 ms per step.  OPT.fuse_inbwd = False keeps the two passes.
        pre_sums, ws = 0, eng.scratch.get()
        od = getattr(g, "w6_out_desc", None) if g is not None else None
        if od is not None and od.r == 6 and g_fold and self.norm and g.pad == 1 and dy.t16 is None and OPT.fuse_inbwd:
            mo = _w6_geo(od.r if od.r else 3)[0]
            Hp, Wp = od.H, od.W
            if (Hp == self.OH + 2 and Wp == self.OW + 2 and od.K == self.cout and od.B == inp.B and min(Hp, Wp) >= 6
                    and (Hp - 3) // mo == (Hp - 1) // mo and (Wp - 3) // mo == (Wp - 1) // mo):
                chunks = (-(-Hp // mo)) * (-(-Wp // mo))
                if gsum is None:                   # no skip path: the folded gradient lives in a pooled dense buffer until the second pass
                    key = ("inbwd_gz", inp.B, self.OH, self.OW, self.cout)
                    pool = ctx.__dict__.setdefault("inbwd_pool", {})
                    if key not in pool:
                        pool[key] = Halo(ctx, inp.B, self.OH, self.OW, self.cout, 0)
                    gsum = pool[key]
                if not hasattr(ctx, "inbwd_part"):
                    ctx.inbwd_part = SplitPool(ctx)
                ws = ctx.inbwd_part.get(inp.B * chunks * 2 * self.cout + inp.B * 2 * self.cout)
                od.fuse_y, od.fuse_mean, od.fuse_rstd = self.y.ptr, self.stats[0].data_ptr(), self.stats[1].data_ptr()
                od.fuse_g2 = g2.ptr if g2 is not None else None
                od.fuse_gz, od.fuse_part, od.fuse_part_elems = gsum.ptr, ws.data_ptr(), ws.numel()
                od.fuse_act, od.fuse_slope = act, 0.2
                pre_sums = chunks
        # The gradient arrives from direct-tile data-gradient launches (the sub-pixel phases of a stride-2 convolution, or a transposed
        # convolution's strided one): their epilogue takes this layer's first backward pass next to the store (nirgan_conv_desc.fuse_*).
        # Worth it on the large maps only (threshold as for the forward statistics); OPT.fuse_inbwd = False keeps the separate pass.
        cds = getattr(g, "conv_out_descs", None) if g is not None else None
        if (cds and not pre_sums and self.norm and not g_fold and g2 is None and gsum is None and self.cout % 4 == 0
                and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) and OPT.fuse_inbwd
                and all(c.ksplit <= 1 and (c.OH * c.OW) % 128 == 0 and channels_of(c) == self.cout and not c.bias for c in cds)
                and sum(c.OH * c.OW * max(1, c.out_span) for c in cds) == self.OH * self.OW
                and self.OH * self.OW >= (OPT.epilogue_min_pixels if ctx.precision == 0 else min(OPT.epilogue_min_pixels, OPT.epilogue_min_pixels_bf16))):
            chunks = sum(c.OH * c.OW // 128 * max(1, c.out_span) for c in cds)
            if not hasattr(ctx, "inbwd_part"):
                ctx.inbwd_part = SplitPool(ctx)
            ws = ctx.inbwd_part.get(inp.B * chunks * 2 * self.cout + inp.B * 2 * self.cout)
            first = 0
            for c in cds:
                c.fuse_y, c.fuse_mean, c.fuse_rstd = self.y.ptr_any(), self.stats[0].data_ptr(), self.stats[1].data_ptr()
                c.fuse_y_bf16 = 1 if self.y.is16 else 0
                c.fuse_h, c.fuse_w = self.OH, self.OW
                c.fuse_oh, c.fuse_ow = c.out_oh - g.pad, c.out_ow - g.pad
                c.fuse_act, c.fuse_slope = act, 0.2
                c.fuse_part, c.fuse_part_elems, c.fuse_chunk0, c.fuse_chunks = ws.data_ptr(), ws.numel(), first, chunks
                first += c.OH * c.OW // 128 * max(1, c.out_span)
            pre_sums = chunks
        nd = emit_in_bwd(plan, ctx, g=g, g_fold=g_fold, g2=g2, act=act,
                         y=self.y, stats=self.stats, norm=self.norm, dy=self.dy, gsum=gsum,
                         dbias=(None if self.norm else gb),   # a bias in front of InstanceNorm has gradient exactly 0: left at 0
                         ws=ws,
                         shape=(inp.B, self.OH, self.OW, self.cout), pre_sums=pre_sums, sums_only=fuse_dy)
        # stride-1 convolutions that need both gradients
        if w6_bwd:
            # Winograd: data gradient and transform-domain weight gradient, dY read once for both of its transforms
            wd6 = emit_wino6(None, pack, ctx, dy, self.weight, None, _FullExtent(dgrad_out), H=dgrad_out.hp, W=dgrad_out.wp,
                             cin=self.cout, cout=inp.C, flip=True, r=k)
            keepV = getattr(self, "wino_fwd_keeps_V", False) and getattr(self, "wino6", False)
            emit_wino6_backward(plan, ctx, dy, inp, gw, OH=self.OH, OW=self.OW, cin=inp.C, cout=self.cout, slabs_pool=eng.slabs,
                                dgrad=wd6, V_fwd=(self.wino_fwd if keepV else None), r=k, norm_desc=(nd if fuse_dy else None))
            if k == 3:
                dgrad_out.w6_out_desc = wd6          # the consumer of this gradient may switch the output transform to its fused mode
            return
        # direct tiles: one fused launch (data-gradient tiles + weight-gradient tiles)
        if self.kind == "conv" and s == 1 and gw is not None and dgrad_out is not None:
            assert dgrad_out.H == inp.H and dgrad_out.pad == p and dy.pad == k - 1
            hw = [(kh, kw) for kh in range(k) for kw in range(k)]
            wd = eng.weights.packed(pack, self.weight, G.conv_dgrad_pack(self.cout, inp.C, k, hw))
            cdesc = emit_conv(None, ctx, dy, G.conv_dgrad_s1_taps(k, self.cout), wd, None, dgrad_out, N=inp.C,
                              OH=dgrad_out.hp, OW=dgrad_out.wp)
            emit_wgrad(plan, ctx, dy, inp, G.conv_fwd_taps(k, inp.C), G.conv_fwd_pack(self.cout, inp.C, k), gw,
                       N=self.cout, OH=self.OH, OW=self.OW, p_oh=dy.pad, p_ow=dy.pad, q_stride=s,
                       q_oh=inp.pad - p, q_ow=inp.pad - p, slabs_pool=eng.slabs, pair_with=cdesc)
            return
        # weight gradient (skipped when the parameters are frozen: gw is None)
        if gw is None:
            pass
        elif self.kind == "conv":
            emit_wgrad(plan, ctx, dy, inp, G.conv_fwd_taps(k, inp.C), G.conv_fwd_pack(self.cout, inp.C, k), gw,
                       N=self.cout, OH=self.OH, OW=self.OW, p_oh=dy.pad, p_ow=dy.pad, q_stride=s,
                       q_oh=inp.pad - p, q_ow=inp.pad - p, slabs_pool=eng.slabs)
        elif self.kind == "rowpacked" and s == 1 and emit_wgrad_pixel_pairs(plan, ctx, dy, inp, gw, k=k, p=p, cin=self.cin, cout=self.cout,
                                                                            OH=self.OH, OW=self.OW):
            pass
        elif self.kind == "rowpacked":
            emit_wgrad(plan, ctx, dy, inp, G.conv_rowpacked_taps(k, inp.C), G.conv_rowpacked_pack(self.cout, self.cin, k, inp.C), gw,
                       N=self.cout, OH=self.OH, OW=self.OW, p_oh=dy.pad, p_ow=dy.pad, q_stride=s,
                       q_oh=inp.pad - p, q_ow=inp.pad - p, slabs_pool=eng.slabs)
        else:  # convT: rows = input channels, gathered side = dY with stride 2
            emit_wgrad(plan, ctx, inp, dy, G.convT_dgrad_taps(k, self.cout), G.convT_dgrad_pack(inp.C, self.cout, k), gw,
                       N=inp.C, OH=inp.H, OW=inp.W, p_oh=inp.pad, p_ow=inp.pad, q_stride=2,
                       q_oh=dy.pad - p, q_ow=dy.pad - p, slabs_pool=eng.slabs)
        # data gradient
        if dgrad_out is None:
            return
        if self.kind == "conv" and s == 1:
            # full correlation: gradient wrt the halo'd input (size H+2p), folded / cropped by the consumer
            assert dgrad_out.H == inp.H and dgrad_out.pad == p and dy.pad == k - 1
            if gw is None and wino_dgrad_applicable(ctx, k, s, dgrad_out, self.cout, inp.C) and inp.C > 64:
                # frozen parameters (the discriminator inside the generator step): the data gradient alone, as a Winograd convolution
                emit_wino6(plan, pack, ctx, dy, self.weight, None, _FullExtent(dgrad_out), H=dgrad_out.hp, W=dgrad_out.wp,
                           cin=self.cout, cout=inp.C, flip=True, r=k)
                return
            taps = G.conv_dgrad_s1_taps(k, self.cout)
            hw = [(kh, kw) for kh in range(k) for kw in range(k)]
            w = eng.weights.packed(pack, self.weight, G.conv_dgrad_pack(self.cout, inp.C, k, hw))
            emit_conv(plan, ctx, dy, taps, w, None, dgrad_out, N=inp.C, OH=dgrad_out.hp, OW=dgrad_out.wp)
        elif self.kind in ("conv", "rowpacked") and s == 2:
            assert dgrad_out.H == inp.H and dy.pad == 1
            cin_buf = inp.C
            descs = None
            phases = G.conv_dgrad_s2_phases(inp.H, inp.W, k, p)
            if self.kind == "conv" and want_phase_pairs(ctx, phases, self.cout, inp.C, dgrad_out):
                descs = emit_phase_pairs(eng, pack, ctx, dy, phases, lambda hw: G.conv_dgrad_pack(self.cout, inp.C, k, hw), self.weight, None,
                                         dgrad_out, N=inp.C, in_off=0, out_off=dgrad_out.pad)
            for ph in (phases if descs is None else []):
                descs = descs or []
                if self.kind == "conv":
                    spec = G.conv_dgrad_pack(self.cout, inp.C, k, ph.taps_hw)
                else:
                    spec = G.conv_dgrad_pack(self.cout, self.cin, k, ph.taps_hw)
                    cin_buf = self.cin
                w = eng.weights.packed(pack, self.weight, spec)
                descs.append(emit_conv(None, ctx, dy, G.Taps(ph.dh, ph.dw, self.cout), w, None, dgrad_out, N=cin_buf,
                                       OH=ph.n_h, OW=ph.n_w, in_oh=ph.in_oh, in_ow=ph.in_ow, out_stride=2,
                                       out_oh=ph.out_oh + dgrad_out.pad, out_ow=ph.out_ow + dgrad_out.pad))
            emit_conv_group(plan, ctx, descs)
            dgrad_out.conv_out_descs = descs             # the consumer of this gradient may have the epilogues take its first backward pass
        elif self.kind == "convT":
            assert dgrad_out.H == inp.H and dy.pad == 1
            w = eng.weights.packed(pack, self.weight, G.convT_dgrad_pack(inp.C, self.cout, k))
            cd = emit_conv(plan, ctx, dy, G.convT_dgrad_taps(k, self.cout), w, None, dgrad_out, N=inp.C, OH=inp.H, OW=inp.W,
                           in_stride=2, in_oh=dy.pad - p, in_ow=dy.pad - p, out_oh=dgrad_out.pad, out_ow=dgrad_out.pad)
            dgrad_out.conv_out_descs = [cd]
        else:
            raise NotImplementedError(self.kind)


class TapPlaneConv:
    """Conv2d(C, 1, k, padding=p) (+bias, +tanh, +crop) (model/networks.py:367-368 and :579).

    Conv2d(64, 1, 7), the generator's last layer, runs as direct kernels with the 64 channels on the 64 lanes of a wave
    (csrc/endconv.hip); other widths as a 1x1 product into k*k tap planes followed by a shifted gather-sum."""

    def __init__(self, eng, name, inp: Halo, weight, bias, *, k, p, act=L.ACT_NONE, crop=0):
        self.eng, self.name, self.inp, self.weight, self.bias = eng, name, inp, weight, bias
        self.k, self.p, self.act, self.crop = k, p, act, crop
        assert inp.pad == p
        ctx = eng.ctx
        self.nt = k * k
        self.qcs = -(-self.nt // 4) * 4
        self.OH, self.OW = inp.hp - k + 1, inp.wp - k + 1
        self.direct = inp.C == 64 and k == 7 and OPT.endconv_direct      # exact fp32 kernels in every precision mode
        if not self.direct:
            self.q = Halo(ctx, inp.B, inp.hp, inp.wp, self.qcs, 0)
        self.whole_in = Halo(ctx, inp.B, inp.hp, inp.wp, inp.C, 0, tensor=inp.t)     # same memory, halo as image
        inp.pinned = True       # read as fp32 through the alias above (and by the direct kernels): never a twin-only buffer
        self.dst = ctx.zeros(inp.B, 1, self.OH - 2 * crop, self.OW - 2 * crop)

    def _direct_desc(self, w):
        inp = self.inp
        d = L.EndConvDesc()
        d.x, d.x_hp, d.x_wp = inp.ptr, inp.hp, inp.wp
        d.B, d.OH, d.OW, d.crop, d.C, d.k = inp.B, self.OH, self.OW, self.crop, inp.C, self.k
        d.w, d.bias, d.act, d.out = w.data_ptr(), _ptr(self.bias), self.act, self.dst.data_ptr()
        self.eng.ctx.keep.append(d)
        return d

    def emit_fwd(self, plan: Plan, pack: Plan):
        eng, ctx = self.eng, self.eng.ctx
        w = eng.weights.packed(pack, self.weight, G.tapplane_fwd_pack(self.inp.C, self.k), rows_alloc=self.qcs, fp32=self.direct)
        if self.direct:
            plan.add("nirgan_endconv_fwd", C.byref(self._direct_desc(w)))
            return
        emit_conv(plan, ctx, self.whole_in, G.Taps([0], [0], self.inp.C), w, None, self.q, N=self.qcs,
                  OH=self.inp.hp, OW=self.inp.wp)
        d = L.TapGatherDesc()
        d.q, d.q_hp, d.q_wp, d.q_cs, d.ntaps = self.q.ptr, self.q.hp, self.q.wp, self.qcs, self.nt
        for t in range(self.nt):
            d.tap_dh[t], d.tap_dw[t] = t // self.k, t % self.k
        d.bias, d.act = _ptr(self.bias), self.act
        d.B, d.OH, d.OW, d.crop, d.dst = self.inp.B, self.OH, self.OW, self.crop, self.dst.data_ptr()
        ctx.keep.append(d)
        plan.add("nirgan_tap_gather", C.byref(d))

    def alloc_bwd(self):
        ctx, inp = self.eng.ctx, self.inp
        if self.direct:
            be = L.backend()
            self.dz = ctx.zeros(be.nirgan_endconv_dz_elems(inp.B, self.OH, self.OW))
            self.ws = ctx.zeros(be.nirgan_endconv_ws_elems(inp.B, self.OH, self.OW))
        else:
            self.dq = Halo(ctx, inp.B, inp.hp, inp.wp, self.qcs, 0)
        self.gin = Halo(ctx, inp.B, inp.H, inp.W, inp.C, inp.pad)     # gradient wrt the halo'd input
        self.gin_whole = Halo(ctx, inp.B, inp.hp, inp.wp, inp.C, 0, tensor=self.gin.t)
        self.dout = ctx.zeros(inp.B, 1, self.OH - 2 * self.crop, self.OW - 2 * self.crop)

    def emit_bwd(self, plan: Plan, pack: Plan, gw: Optional[torch.Tensor], gb: Optional[torch.Tensor]):
        eng, ctx, inp = self.eng, self.eng.ctx, self.inp
        if self.direct:
            w = eng.weights.packed(pack, self.weight, G.tapplane_fwd_pack(inp.C, self.k), rows_alloc=self.qcs, fp32=True)
            d = self._direct_desc(w)
            d.dout, d.dz, d.dz_elems = self.dout.data_ptr(), self.dz.data_ptr(), self.dz.numel()
            d.gx, d.gw, d.gbias = self.gin.ptr, _ptr(gw), _ptr(gb)
            d.ws, d.ws_elems = self.ws.data_ptr(), self.ws.numel()
            plan.add("nirgan_endconv_dz", C.byref(d))
            if gw is not None:
                plan.add("nirgan_endconv_wgrad", C.byref(d))
            plan.add("nirgan_endconv_dgrad", C.byref(d))
            return
        d = L.TapScatterDesc()
        d.dout, d.out, d.act = self.dout.data_ptr(), self.dst.data_ptr(), self.act
        d.B, d.OH, d.OW, d.crop, d.ntaps = inp.B, self.OH, self.OW, self.crop, self.nt
        for t in range(self.nt):
            d.tap_dh[t], d.tap_dw[t] = t // self.k, t % self.k
        d.dq, d.q_hp, d.q_wp, d.q_cs = self.dq.ptr, self.dq.hp, self.dq.wp, self.qcs
        d.dbias = _ptr(gb)
        ctx.keep.append(d)
        self.scatter_desc = d
        plan.add("nirgan_tap_scatter", C.byref(d))
        if gw is not None:
            emit_wgrad(plan, ctx, self.dq, self.whole_in, G.Taps([0], [0], inp.C), G.tapplane_fwd_pack(inp.C, self.k), gw,
                       N=self.nt, OH=inp.hp, OW=inp.wp, p_oh=0, p_ow=0, slabs_pool=eng.slabs)
        w = eng.weights.packed(pack, self.weight, G.tapplane_dgrad_pack(inp.C, self.k, self.qcs))
        emit_conv(plan, ctx, self.dq, G.Taps([0], [0], self.qcs), w, None, self.gin_whole, N=inp.C, OH=inp.hp, OW=inp.wp)
