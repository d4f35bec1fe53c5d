"""numpy emulator of the libnirgan_hip C ABI -- TEST INFRASTRUCTURE ONLY.

Implements every entry point of include/nirgan_hip.h on HOST pointers with the documented
semantics (halo'd NHWC buffers, tap lists, split slabs ...).  Installed through the
``nirgan_hip.lib.set_backend`` test seam so that the host logic (descriptor geometry, buffer
plumbing, plan ordering, flat parameters, trainer, data parallel reducer) can be exercised
against the oracle WITHOUT a GPU.  It is never used by the product path, smoke() or bench.py.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F


def arr(ptr, n, dtype=np.float32):
    if ptr is None or ptr == 0:
        return None
    if hasattr(ptr, "value"):
        ptr = ptr.value
    ct = {np.float32: C.c_float, np.float64: C.c_double}.get(dtype, C.c_int32)
    return np.ctypeslib.as_array((ct * int(n)).from_address(int(ptr)))


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16 -> fp32 (what v_cvt_pk_bf16_f32 does), finite inputs."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + (0x7FFF + ((u >> 16) & 1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def contract(A: np.ndarray, B: np.ndarray, precision: int) -> np.ndarray:
    """A @ B with the operand treatment of the descriptor's `precision` (accumulation in float64 here)."""
    if precision == 0:
        return A.astype(np.float64) @ B.astype(np.float64)
    ah, bh = bf16_round(A), bf16_round(B)
    if precision == 1:
        return ah.astype(np.float64) @ bh.astype(np.float64)
    am, bm = bf16_round(A - ah), bf16_round(B - bh)
    if precision == 3:        # three bf16 terms per operand, the six products down to 2^-16 of the leading one (csrc/igemm_x3.h)
        al, bl = bf16_round(A - ah - am), bf16_round(B - bh - bm)
        ah, bh, am, bm, al, bl = (v.astype(np.float64) for v in (ah, bh, am, bm, al, bl))
        return al @ bh + ah @ bl + am @ bm + am @ bh + ah @ bm + ah @ bh
    ah, bh, am, bm = (v.astype(np.float64) for v in (ah, bh, am, bm))
    return ah @ bh + ah @ bm + am @ bh


def split3_planes(x: np.ndarray):
    """x = h + m + l, each term the nearest-even bf16 of what the previous ones left (nirgan_split3), as uint16 bit patterns"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    h = bf16_round(x)
    m = bf16_round(x - h)
    l = bf16_round(x - h - m)
    return [(v.view(np.uint32) >> 16).astype(np.uint16) for v in (h, m, l)]


def mirror_twin(twin_ptr, buf: np.ndarray) -> None:
    """bf16 twin of a buffer: same geometry, every value rounded to nearest even (what the producers' mirrored stores leave)."""
    if twin_ptr:
        t = np.ctypeslib.as_array((C.c_uint16 * buf.size).from_address(int(twin_ptr)))
        t[:] = (bf16_round(buf.reshape(-1)).view(np.uint32) >> 16).astype(np.uint16)


def obj(ref):
    return ref._obj if hasattr(ref, "_obj") else ref


def arr16(ptr, n) -> np.ndarray:
    """n bf16 elements at ptr as a uint16 view (writable)."""
    return np.ctypeslib.as_array((C.c_uint16 * int(n)).from_address(int(ptr)))


def load_y(ptr, n, is_bf16) -> np.ndarray:
    """a convolution output in front of an instance norm: fp32, or (bf16 operand mode, y_bf16 / fuse_y_bf16) stored as bf16"""
    if is_bf16:
        return (arr16(ptr, n).astype(np.uint32) << 16).view(np.float32)
    return arr(ptr, n)


def _in_nchunk(B, HW, Cc):
    q4 = Cc // 4
    nrg = 1 if q4 >= 256 else 256 // q4
    want = max(1, 2048 // B)
    cap = max(1, HW // (nrg * 8))
    return min(want, cap)


def halo_images(h, H, P):
    out = [h + P]
    if 1 <= h <= P:
        out.append(P - h)
    if H - 1 - P <= h <= H - 2:
        out.append(P + 2 * (H - 1) - h)
    return out


def reflect(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


class EmuBackend:
    is_emulator = True

    def __init__(self):
        self.err = b""
        self.calls = []

    # ------------------------------------------------------------------ housekeeping
    def nirgan_version(self):
        return 100

    def nirgan_last_error(self):
        return self.err

    def _fail(self, msg):
        self.err = msg.encode()
        return -1

    # ------------------------------------------------------------------ conv
    def nirgan_conv_igemm(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("conv")
        if d.run % 4 or d.in_cs % 4:
            return self._fail("conv: run/in_cs not multiple of 4")
        K = d.ntaps * d.run
        if d.in_bf16:
            i16 = np.ctypeslib.as_array((C.c_uint16 * int(d.in_elems)).from_address(int(d.inp)))
            inp = (i16.astype(np.uint32) << 16).view(np.float32)
        else:
            inp = arr(d.inp, d.in_elems)
        if d.w_bf16:
            w16 = np.ctypeslib.as_array((C.c_uint16 * int(d.w_elems)).from_address(int(d.w)))[: d.N * K]
            w = (w16.astype(np.uint32) << 16).view(np.float32).reshape(d.N, K)
        else:
            w = arr(d.w, d.w_elems)[: d.N * K].reshape(d.N, K)
        prec = d.precision
        if prec == 3:
            # the split tile's coverage (csrc/igemm_x3.h::conv_x3_ok); anything else runs as exact fp32
            if d.w_x3 and d.run % 32 == 0 and d.N % 64 == 0 and d.ksplit <= 1 and not (d.in_bf16 or d.w_bf16 or d.out_bf16 or d.fuse_y_bf16):
                if d.w_x3_plane < d.N * K or d.w_x3_plane % 8:
                    return self._fail("conv: w_x3_plane too small")
                planes = [(arr16(d.w_x3 + 2 * t * d.w_x3_plane, d.N * K).astype(np.uint32) << 16).view(np.float32).astype(np.float64) for t in range(3)]
                w3 = (planes[0] + planes[1] + planes[2]).reshape(d.N, K)
                if not np.array_equal(w3, w.astype(np.float64)):
                    return self._fail("conv: the w_x3 planes are not the three-term split of w (stale planes?)")
            else:
                prec = 0
        if d.out_bf16 and (d.N % 4 or d.out_cs % 4 or d.ksplit > 1):
            return self._fail("conv: a bf16 output needs N % 4 == 0, out_cs % 4 == 0 and no split-K")
        out = arr16(d.out, d.out_elems) if d.out_bf16 else arr(d.out, d.out_elems)
        bias = arr(d.bias, d.N)
        in_row, out_row = d.in_wp * d.in_cs, d.out_wp * d.out_cs
        in_img, out_img = d.in_hp * in_row, d.out_hp * out_row
        oh, ow = np.arange(d.OH)[:, None], np.arange(d.OW)[None, :]
        base = (oh * d.in_stride + d.in_oh) * in_row + (ow * d.in_stride + d.in_ow) * d.in_cs
        obase = (oh * d.out_stride + d.out_oh) * out_row + (ow * d.out_stride + d.out_ow) * d.out_cs
        rr = np.arange(d.run)
        for t in range(d.ntaps):
            r0, r1 = (d.OH - 1) * d.in_stride + d.in_oh + d.tap_dh[t], d.in_oh + d.tap_dh[t]
            if min(r0, r1) < 0 or max(r0, r1) >= d.in_hp:
                return self._fail("conv: input rows out of range")
            c0 = d.in_ow + d.tap_dw[t]
            c1 = (d.OW - 1) * d.in_stride + d.in_ow + d.tap_dw[t]
            if c0 < 0 or c1 * d.in_cs + d.run > d.in_wp * d.in_cs:
                return self._fail("conv: input cols out of range")
        span = d.out_span if d.out_span > 1 else 1
        ch = d.N // span                                  # channels per output pixel
        span_view = span == 2 and d.out_cs == d.N        # the caller describes the output in pixel pairs already
        if span > 1 and not (span == 2 and d.precision == 3 and prec == 3 and d.N % 8 == 0
                             and ((not d.fuse_y) if span_view else (d.out_cs == ch and d.out_stride >= 2))
                             and not (d.out_bf16 or d.fuse_y_bf16) and d.ksplit <= 1):
            return self._fail("conv: out_span needs precision 3 with its weight planes on a problem the split tile covers, out_cs == N / 2 and out_stride >= 2 (or out_cs == N)")
        if ch > d.out_cs or (d.OH - 1) * d.out_stride + d.out_oh >= d.out_hp or (d.OW - 1) * d.out_stride + d.out_ow + (0 if span_view else span - 1) >= d.out_wp:
            return self._fail("conv: output out of range")
        if d.fuse_y:
            if (d.ksplit > 1 or (d.OH * d.OW) % 128 or d.N % 4 or d.bias or not (d.fuse_mean and d.fuse_rstd and d.fuse_part)
                    or (d.OH - 1) * d.out_stride + d.fuse_oh >= d.fuse_h or (d.OW - 1) * d.out_stride + d.fuse_ow + span - 1 >= d.fuse_w
                    or d.fuse_chunk0 < 0 or d.fuse_chunk0 + d.OH * d.OW // 128 * span > d.fuse_chunks or d.fuse_part_elems < d.B * d.fuse_chunks * 2 * ch):
                return self._fail("conv: the fused instance-norm backward sums need OH*OW % 128 == 0, N % 4 == 0, no split-K, no bias, a window inside y and a large enough fuse_part")
            self.calls.append("conv_inbwd")
        stats = None
        if d.stats_ws:
            ohw = d.OH * d.OW
            if d.ksplit > 1 or ohw % 128 or d.stats_chunk0 < 0 or d.stats_chunk0 + ohw // 64 * span > d.stats_chunks or d.stats_ws_elems < d.B * d.stats_chunks * 4 * ch:
                return self._fail("conv: the instance-norm partial sums need OH*OW % 128 == 0, no split-K and a large enough stats_ws")
            stats = arr(d.stats_ws, d.B * d.stats_chunks * 4 * ch).reshape(d.B, d.stats_chunks, 4, ch)
            self.calls.append("conv_stats")
        for b in range(d.B):
            acc = np.zeros((d.OH, d.OW, d.N), dtype=np.float64)
            for t in range(d.ntaps):
                off = (d.tap_dh[t] * d.in_wp + d.tap_dw[t]) * d.in_cs
                A = inp[b * in_img + base[..., None] + off + rr]
                acc += contract(A.reshape(-1, d.run), np.ascontiguousarray(w[:, t * d.run:(t + 1) * d.run].T), prec).reshape(d.OH, d.OW, d.N)
            if stats is not None:                      # per 64 rows: {k = first row, sum (v - k), sum (v - k)^2, 64}, without the bias
                # (out_span = 2: a row holds two pixels -- two records of `ch` columns per 64 rows, the first pixel's first)
                rows = acc.reshape(-1, 64, span, ch).transpose(0, 2, 1, 3).reshape(-1, 64, ch)
                k = rows[:, 0]
                sl = slice(d.stats_chunk0, d.stats_chunk0 + rows.shape[0])
                stats[b, sl, 0] = k
                stats[b, sl, 1] = (rows - k[:, None]).sum(1)
                stats[b, sl, 2] = ((rows - k[:, None]) ** 2).sum(1)
                stats[b, sl, 3] = 64.0
            if bias is not None:
                acc += bias
            idx = b * out_img + obase[..., None] + np.arange(d.N)
            if d.out_bf16:
                out[idx] = (bf16_round(acc.astype(np.float32)).view(np.uint32) >> 16).astype(np.uint16)
            else:
                out[idx] = acc.astype(np.float32)
            if d.fuse_y:                               # first pass of the consumer layer's instance-norm backward, per 128-pixel tile
                yv = load_y(d.fuse_y, d.B * d.fuse_h * d.fuse_w * ch, d.fuse_y_bf16).reshape(d.B, d.fuse_h, d.fuse_w, ch)
                ys = np.stack([yv[b, d.fuse_oh:d.fuse_oh + (d.OH - 1) * d.out_stride + 1:d.out_stride,
                                  d.fuse_ow + q:d.fuse_ow + q + (d.OW - 1) * d.out_stride + 1:d.out_stride] for q in range(span)], axis=2)     # [OH][OW][span][ch]
                z = ((ys - arr(d.fuse_mean, d.B * ch).reshape(d.B, ch)[b]) * arr(d.fuse_rstd, d.B * ch).reshape(d.B, ch)[b]).astype(np.float32)
                gv = acc.astype(np.float32).astype(np.float64).reshape(d.OH, d.OW, span, ch)
                neg = 0.0 if d.fuse_act == 1 else (d.fuse_slope if d.fuse_act == 2 else 1.0)
                gz = np.where(z > 0, gv, gv * neg)
                part = arr(d.fuse_part, d.B * d.fuse_chunks * 2 * ch).reshape(d.B, d.fuse_chunks, 2, ch)
                nch = d.OH * d.OW // 128
                # per 128 rows: one record per pixel of the row (span of them), the first pixel's first
                part[b, d.fuse_chunk0:d.fuse_chunk0 + nch * span, 0] = gz.reshape(nch, 128, span, ch).sum(1).reshape(nch * span, ch)
                part[b, d.fuse_chunk0:d.fuse_chunk0 + nch * span, 1] = (gz * z).reshape(nch, 128, span, ch).sum(1).reshape(nch * span, ch)
        return 0

    def nirgan_wgrad_igemm(self, ref, stream=None):
        d = obj(ref)
        if d.nplanes > 1:                      # independent problems of identical geometry: run them one by one
            import copy
            K = d.ntaps * d.run
            for i in range(d.nplanes):
                one = type(d)()
                C.memmove(C.byref(one), C.byref(d), C.sizeof(d))
                one.nplanes = 1
                one.p, one.q = int(d.p) + 4 * i * d.p_plane, int(d.q) + 4 * i * d.q_plane
                one.p_elems, one.q_elems = d.p_elems - i * d.p_plane, d.q_elems - i * d.q_plane
                one.slabs = int(d.slabs) + 4 * i * d.nsplit * d.N * K
                one.slab_elems = d.nsplit * d.N * K
                rc = self.nirgan_wgrad_igemm(one)
                if rc:
                    return rc
            return 0
        self.calls.append("wgrad")
        K = d.ntaps * d.run
        if d.rows_per_split % 32 or d.nsplit * d.rows_per_split < d.B * d.OH * d.OW:
            return self._fail("wgrad: bad split")
        if ((d.N + 3) & ~3) > d.p_cs:
            return self._fail("wgrad: N exceeds p_cs")
        if d.pq_bf16:
            rd = lambda ptr, n: (np.ctypeslib.as_array((C.c_uint16 * int(n)).from_address(int(ptr))).astype(np.uint32) << 16).view(np.float32)   # noqa: E731
            p, q = rd(d.p, d.p_elems), rd(d.q, d.q_elems)
        else:
            p, q = arr(d.p, d.p_elems), arr(d.q, d.q_elems)
        slabs = arr(d.slabs, d.slab_elems)[: d.nsplit * d.N * K].reshape(d.nsplit, d.N, K)
        p_row, q_row = d.p_wp * d.p_cs, d.q_wp * d.q_cs
        oh, ow = np.arange(d.OH)[:, None], np.arange(d.OW)[None, :]
        pbase = (oh + d.p_oh) * p_row + (ow + d.p_ow) * d.p_cs
        qbase = (oh * d.q_stride + d.q_oh) * q_row + (ow * d.q_stride + d.q_ow) * d.q_cs
        for t in range(d.ntaps):
            r0, r1 = d.q_oh + d.tap_dh[t], (d.OH - 1) * d.q_stride + d.q_oh + d.tap_dh[t]
            c0, c1 = d.q_ow + d.tap_dw[t], (d.OW - 1) * d.q_stride + d.q_ow + d.tap_dw[t]
            if r0 < 0 or r1 >= d.q_hp or c0 < 0 or c1 * d.q_cs + d.run > d.q_wp * d.q_cs:
                return self._fail("wgrad: q window out of range")
        if d.OH - 1 + d.p_oh >= d.p_hp or d.OW - 1 + d.p_ow >= d.p_wp:
            return self._fail("wgrad: p window out of range")
        P_all, Q_all = [], []
        for b in range(d.B):
            P_all.append(p[b * d.p_hp * p_row + pbase[..., None] + np.arange(d.N)].reshape(-1, d.N))
            qs = []
            for t in range(d.ntaps):
                off = (d.tap_dh[t] * d.q_wp + d.tap_dw[t]) * d.q_cs
                qs.append(q[b * d.q_hp * q_row + qbase[..., None] + off + np.arange(d.run)].reshape(-1, d.run))
            Q_all.append(np.concatenate(qs, axis=1))
        P_all, Q_all = np.concatenate(P_all), np.concatenate(Q_all)
        M = P_all.shape[0]
        for s in range(d.nsplit):
            a, e = s * d.rows_per_split, min((s + 1) * d.rows_per_split, M)
            slabs[s] = contract(np.ascontiguousarray(P_all[a:e].T), Q_all[a:e], d.precision).astype(np.float32) if e > a else 0.0
        return 0

    def nirgan_conv_igemm_group(self, descs, n, stream=None):
        for i in range(n):
            rc = self.nirgan_conv_igemm(descs[i].contents)
            if rc:
                return rc
        return 0

    def nirgan_conv_wgrad_pair(self, cref, wref, stream=None):
        rc = self.nirgan_conv_igemm(cref)
        return rc if rc else self.nirgan_wgrad_igemm(wref)

    def nirgan_reduce_rows(self, slabs, nsplit, N, K, imap, dst, dst_elems, stride, accumulate, stream=None):
        self.calls.append("reduce")
        s = arr(slabs, nsplit * N * K).reshape(nsplit, N, K).sum(0)
        m = arr(imap, K, np.int32)
        o = arr(dst, dst_elems)
        ok = m >= 0
        idx = (np.arange(N)[:, None] * stride + m[None, :])[:, ok]
        if idx.max() >= dst_elems:
            return self._fail("reduce_rows: index out of range")
        if accumulate:
            o[idx] += s[:, ok]
        else:
            o[idx] = s[:, ok]
        return 0

    def nirgan_reduce_rows_part(self, slabs, nsplit, N, row0, rows, K, imap, dst, dst_elems, stride, accumulate, stream=None):
        self.calls.append("reduce_part")
        if row0 < 0 or rows <= 0 or row0 + rows > N:
            return self._fail("reduce_rows_part: rows outside the slab")
        s = arr(slabs, nsplit * N * K).reshape(nsplit, N, K).sum(0)[row0:row0 + rows]
        m = arr(imap, K, np.int32)
        o = arr(dst, dst_elems)
        ok = m >= 0
        idx = (np.arange(rows)[:, None] * stride + m[None, :])[:, ok]
        if idx.max() >= dst_elems:
            return self._fail("reduce_rows_part: index out of range")
        if accumulate:
            o[idx] += s[:, ok]
        else:
            o[idx] = s[:, ok]
        return 0

    def nirgan_pack_rows_bf16(self, src, src_elems, stride, imap, dst, N, K, stream=None):
        return self.nirgan_pack_rows(src, src_elems, stride, imap, dst, N, K, bf16=True)

    def nirgan_pack_rows(self, src, src_elems, stride, imap, dst, N, K, stream=None, bf16=False):
        self.calls.append("pack")
        if bf16:
            tmp = np.zeros((N, K), dtype=np.float32)
            rc = self.nirgan_pack_rows(src, src_elems, stride, imap, tmp.ctypes.data, N, K)
            if rc:
                return rc
            o16 = np.ctypeslib.as_array((C.c_uint16 * (N * K)).from_address(int(dst if not hasattr(dst, "value") else dst.value)))
            o16[:] = (bf16_round(tmp).view(np.uint32) >> 16).astype(np.uint16).reshape(-1)
            return 0
        s, m, o = arr(src, src_elems), arr(imap, K, np.int32), arr(dst, N * K).reshape(N, K)
        ok = m >= 0
        idx = np.arange(N)[:, None] * stride + np.where(ok, m, 0)[None, :]
        if idx.max() >= src_elems:
            return self._fail("pack_rows: index out of range")
        o[:] = np.where(ok[None, :], s[idx], 0.0)
        return 0

    def nirgan_reduce_rows_batch(self, jobs, njobs, total_blocks, stream=None):
        J = np.ctypeslib.as_array((C.c_int64 * (njobs * 10)).from_address(int(jobs))).reshape(njobs, 10)
        blocks = 0
        for slabs, dst, imap, nsplit, N, K, dst_elems, stride, first, taps in J:
            if first != blocks:
                return self._fail("reduce_rows_batch: first_block mismatch")
            if taps:
                cin = int(K) // int(taps)
                m = arr(int(imap), int(K), np.int32)
                if int(K) % int(taps) or cin % 64 or taps > 16 or not np.array_equal(m.reshape(int(taps), cin), np.arange(cin)[None, :] * int(taps) + np.arange(int(taps))[:, None]):
                    return self._fail("reduce_rows_batch: taps given but the map is not (t, c) -> c * taps + t")
            rc = self.nirgan_reduce_rows(int(slabs), int(nsplit), int(N), int(K), int(imap), int(dst), int(dst_elems), int(stride) & 0xffffffff, int(stride) >> 32)
            if rc:
                return rc
            blocks += int(N) * ((int(K) // int(taps) // 64) if taps else ((int(K) + 255) // 256))
        return 0 if blocks == total_blocks else self._fail("reduce_rows_batch: total_blocks mismatch")

    def nirgan_split3(self, src, dst, n, plane, stream=None):
        self.calls.append("split3")
        if n <= 0 or n % 8 or plane < n or plane % 8:
            return self._fail("split3: n and plane must be positive multiples of 8, plane >= n")
        for t, v in enumerate(split3_planes(arr(src, n))):
            arr16(int(dst) + 2 * t * int(plane), n)[:] = v
        return 0

    def nirgan_pack_rows_batch(self, jobs, njobs, total_blocks, stream=None):
        J = np.ctypeslib.as_array((C.c_int64 * (njobs * 10)).from_address(int(jobs))).reshape(njobs, 10)
        blocks = 0
        for src, dst, imap, src_elems, N, K, stride, first, w3, w3_plane in J:
            if first != blocks:
                return self._fail("pack_rows_batch: first_block mismatch")
            rc = self.nirgan_pack_rows(int(src), int(src_elems), int(stride) & 0xffffffff, int(imap), int(dst), int(N), int(K),
                                       bf16=bool(int(stride) >> 32))
            if rc:
                return rc
            if w3:
                if int(stride) >> 32:
                    return self._fail("pack_rows_batch: w_x3 planes come with an fp32 destination")
                rc = self.nirgan_split3(int(dst), int(w3), int(N) * int(K), int(w3_plane))
                if rc:
                    return rc
            blocks += int(N) * ((int(K) + 1023) // 1024)
        return 0 if blocks == total_blocks else self._fail("pack_rows_batch: total_blocks mismatch")

    # ------------------------------------------------------------------ SatCLIP location encoder
    def nirgan_location_encoder(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("locenc")
        nf = d.L * d.L
        if d.dims[0] != nf or d.nlayers < 1 or d.nlayers > 8:
            return self._fail("location_encoder: bad layer table")
        ll = arr(d.lonlat, d.B * 2, np.float64).reshape(d.B, 2)
        K = arr(d.sh_norm, nf, np.float64)
        phi, theta = np.deg2rad(ll[:, 0] + 180.0), np.deg2rad(ll[:, 1] + 90.0)
        x = np.cos(theta)
        Y = np.zeros((d.B, nf))
        for l in range(d.L):
            for m in range(-l, l + 1):
                am = abs(m)
                pmm = np.ones_like(x)
                if am > 0:
                    somx2, fact = np.sqrt((1 - x) * (1 + x)), 1.0
                    for _ in range(am):
                        pmm = pmm * (-fact) * somx2
                        fact += 2.0
                if l == am:
                    P = pmm
                else:
                    pmmp1 = x * (2.0 * am + 1.0) * pmm
                    P = pmmp1
                    for q in range(am + 2, l + 1):
                        P = ((2.0 * q - 1.0) * x * pmmp1 - (q + am - 1.0) * pmm) / (q - am)
                        pmm, pmmp1 = pmmp1, P
                f = l * l + l + m
                Y[:, f] = K[f] * P if m == 0 else (K[f] * np.cos(m * phi) * P if m > 0 else K[f] * np.sin(am * phi) * P)
        if d.features:
            arr(d.features, d.B * nf, np.float64)[:] = Y.reshape(-1)
        h = Y
        for i in range(d.nlayers):
            W = arr(d.weights[i], d.dims[i + 1] * d.dims[i], np.float64).reshape(d.dims[i + 1], d.dims[i])
            h = h @ W.T
            if d.biases[i]:
                h = h + arr(d.biases[i], d.dims[i + 1], np.float64)
            if d.w0[i] != 0.0:
                h = np.sin(d.w0[i] * h)
        arr(d.out, d.B * d.dims[d.nlayers], np.float64)[:] = h.reshape(-1)
        return 0

    # ------------------------------------------------------------------ Winograd F(4x4, 3x3) / F(4x4, 4x4) (csrc/wino6.hip)
    _W6 = {
        3: (np.array([[1 / 2, 0, 0], [1 / 6, 1 / 6, 1 / 6], [1 / 6, -1 / 6, 1 / 6], [1 / 30, 1 / 15, 2 / 15], [16 / 15, -8 / 15, 4 / 15], [0, 0, 1 / 2]]),
            np.array([[2, 3, -4, -3, 2, 0], [0, 2, 5, 1, -2, 0], [0, 2, 1, -5, 2, 0], [0, -1, -2, 1, 2, 0], [0, -2, 1, 2, -1, 0], [0, 2, 3, -4, -3, 2]], dtype=np.float64),
            np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -1 / 2, 0], [0, 1, 1, 4, 1 / 4, 0], [0, 1, -1, 8, -1 / 8, 1]], dtype=np.float64)),
        # Cook-Toom over 0, 1, -1, 2, -2, 1/2, inf (csrc/wino6.hip w7_G / w7_BT / w7_AT)
        4: (np.array([[1 / 4, 0, 0, 0], [1 / 6, 1 / 6, 1 / 6, 1 / 6], [1 / 18, -1 / 18, 1 / 18, -1 / 18], [1 / 72, 1 / 36, 1 / 18, 1 / 9],
                      [1 / 120, -1 / 60, 1 / 30, -1 / 15], [32 / 45, 16 / 45, 8 / 45, 4 / 45], [0, 0, 0, 1 / 2]]),
            np.array([[4, -8, -5, 10, 1, -2, 0], [0, -4, 4, 9, -1, -2, 0], [0, -4, 12, -7, -3, 2, 0], [0, 2, -3, -4, 3, 2, 0],
                      [0, 2, -5, 0, 5, -2, 0], [0, 4, 0, -5, 0, 1, 0], [0, -4, 8, 5, -10, -1, 2]], dtype=np.float64),
            np.array([[1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 0], [0, 1, -1, 8, -8, 1 / 8, 1]], dtype=np.float64)),
        # F(6x6,3x3): Cook-Toom over 0, 1, -1, 2, -2, 1/2, -1/2, inf (csrc/wino6.hip w8_G / w8_BT / w8_AT)
        6: (np.array([[1 / 4, 0, 0], [1 / 18, 1 / 18, 1 / 18], [1 / 18, -1 / 18, 1 / 18], [1 / 360, 1 / 180, 1 / 90], [1 / 360, -1 / 180, 1 / 90],
                      [16 / 45, 8 / 45, 4 / 45], [16 / 45, -8 / 45, 4 / 45], [0, 0, 1 / 4]]),
            np.array([[4, 0, -21, 0, 21, 0, -4, 0], [0, -4, -4, 17, 17, -4, -4, 0], [0, 4, -4, -17, 17, 4, -4, 0], [0, 2, 1, -10, -5, 8, 4, 0],
                      [0, -2, 1, 10, -5, -8, 4, 0], [0, 4, 8, -5, -10, 1, 2, 0], [0, -4, 8, 5, -10, -1, 2, 0], [0, -4, 0, 21, 0, -21, 0, 4]], dtype=np.float64),
            np.array([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0], [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 0],
                      [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]], dtype=np.float64)),
    }

    @staticmethod
    def _r6(r):
        return 3 if r == 0 else r

    @staticmethod
    def _geo6(v):
        """variant code -> (filter size, outputs per tile and dimension, points per dimension)"""
        filt, mo = (3, 6) if v == 6 else (v, 4)
        return filt, mo, mo + filt - 1

    def nirgan_wino6_tiles(self, B, H, W):
        return self.nirgan_wino6_tiles_r(B, H, W, 3)

    def nirgan_wino6_tiles_r(self, B, H, W, r):
        v = self._r6(r)
        if v not in self._W6 or min(B, H, W) <= 0:
            return 0
        mo = self._geo6(v)[1]
        return B * (-(-H // mo)) * (-(-W // mo))

    def nirgan_wino6_weights(self, w, K, Cc, flip, U, stream=None):
        return self.nirgan_wino6_weights_r(w, K, Cc, 3, flip, U)

    def nirgan_wino6_weights_r(self, w, K, Cc, r, flip, U, stream=None, U3=None):
        self.calls.append("wino6_w")
        r = self._r6(r)
        if r not in self._W6:
            return self._fail("wino6_weights: filter size")
        Gm = self._W6[r][0]
        variant = r
        r, _, n = self._geo6(r)
        if flip:
            g = arr(w, K * Cc * r * r).reshape(Cc, K, r, r).astype(np.float64).transpose(1, 0, 2, 3)[:, :, ::-1, ::-1]
        else:
            g = arr(w, K * Cc * r * r).reshape(K, Cc, r, r).astype(np.float64)
        u = np.einsum("ai,kcij,bj->abkc", Gm, g, Gm).reshape(-1).astype(np.float32)
        if not U and not U3:
            return self._fail("wino6_weights: neither U nor its planes")
        if U:
            arr(U, n * n * K * Cc)[:] = u
        if U3:            # the same values as three bf16 planes (nirgan_split3's rule), for the three-term split plane GEMMs
            for t, v in enumerate(split3_planes(u)):
                arr16(int(U3) + 2 * t * n * n * K * Cc, n * n * K * Cc)[:] = v
            # (U may be omitted: the plane GEMM then checks its planes against a fresh transform of THESE weights)
            if not hasattr(self, "_u3_src"):
                self._u3_src = {}
            self._u3_src[int(U3)] = (int(w), K, Cc, variant, int(flip))
        return 0

    def nirgan_wino6_weights_x3(self, w, K, Cc, r, flip, U, U3, stream=None):
        if not U3:
            return self._fail("wino6_weights_x3: null planes")
        return self.nirgan_wino6_weights_r(w, K, Cc, r, flip, U, U3=U3)

    def nirgan_wino6_weights_batch(self, jobs, njobs, total_blocks, stream=None):
        J = np.ctypeslib.as_array((C.c_int64 * (njobs * 8)).from_address(int(jobs))).reshape(njobs, 8)
        blocks = 0
        for w, U, K, Cc, flip, first, r, U3 in J:
            if first != blocks:
                return self._fail("wino6_weights_batch: first_block mismatch")
            rc = self.nirgan_wino6_weights_r(int(w), int(K), int(Cc), int(r), int(flip), int(U), U3=int(U3))
            if rc:
                return rc
            blocks += (int(K) * int(Cc) + 255) // 256
        return 0 if blocks == total_blocks else self._fail("wino6_weights_batch: total_blocks mismatch")

    def _wino6_V(self, x, B, H, W, Cc, v=3):
        """x: [B][H+r-1][W+r-1][C] float64 -> V [n][n][B][TH][TW][C]"""
        r, mo, n = self._geo6(v)
        TH, TW = -(-H // mo), -(-W // mo)
        xp = np.zeros((B, mo * TH + r - 1, mo * TW + r - 1, Cc))
        xp[:, :H + r - 1, :W + r - 1] = x
        tiles = np.stack([np.stack([xp[:, i:i + mo * TH:mo, j:j + mo * TW:mo] for j in range(n)], 0) for i in range(n)], 0)
        BT = self._W6[v][1]
        return np.einsum("ai,ijbyxc,lj->albyxc", BT, tiles, BT)

    def nirgan_wino6_input(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("wino6_in")
        v = self._r6(d.r)
        if v not in self._W6:
            return self._fail("wino6_input: variant")
        r, mo, n = self._geo6(v)
        if d.x_hp != d.H + r - 1 or d.x_wp != d.W + r - 1 or d.C % 4:
            return self._fail("wino6_input: bad geometry")
        B, H, W, Cc = d.B, d.H, d.W, d.C
        if d.V_elems < n * n * self.nirgan_wino6_tiles_r(B, H, W, v) * Cc:
            return self._fail("wino6_input: V workspace too small")
        x = arr(d.x, B * d.x_hp * d.x_wp * Cc).reshape(B, d.x_hp, d.x_wp, Cc).astype(np.float64)
        V = self._wino6_V(x, B, H, W, Cc, v)
        arr(d.V, V.size)[:] = V.reshape(-1).astype(np.float32)
        return 0

    def nirgan_wino6_input_norm(self, ref, y, mean, rstd, act, slope, stream=None):
        d = obj(ref)
        self.calls.append("wino6_in_norm")
        v = self._r6(d.r)
        if v not in (3, 6):
            return self._fail("wino6_input_norm: 3x3 filters only")
        B, H, W, Cc = d.B, d.H, d.W, d.C
        yv = arr(y, B * H * W * Cc).reshape(B, H, W, Cc)
        m, r = arr(mean, B * Cc).reshape(B, 1, 1, Cc), arr(rstd, B * Cc).reshape(B, 1, 1, Cc)
        a = self._act(((yv - m) * r).astype(np.float32).astype(np.float64), act, slope).astype(np.float32)       # in_apply's fp32 arithmetic
        hh, ww = reflect(np.arange(H + 2) - 1, H), reflect(np.arange(W + 2) - 1, W)
        V = self._wino6_V(a[:, hh][:, :, ww].astype(np.float64), B, H, W, Cc, v)
        arr(d.V, V.size)[:] = V.reshape(-1).astype(np.float32)
        return 0

    def nirgan_wino6_gemm(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("wino6_gemm")
        T = self.nirgan_wino6_tiles_r(d.B, d.H, d.W, d.r)
        nplanes = self._geo6(self._r6(d.r))[2] ** 2
        if d.K <= 64 or d.K % 4 or d.C % 4 or d.V_elems < nplanes * T * d.C or d.M_elems < nplanes * T * d.K:
            return self._fail("wino6_gemm: bad geometry / workspace")
        V = arr(d.V, nplanes * T * d.C).reshape(nplanes, T, d.C).astype(np.float64)
        if d.U3 and d.C % 32 == 0 and d.K % 64 == 0:
            # precision 3 (csrc/igemm_x3.h): the planes must be the three-term split of U -- of the U given, or (U omitted) of a fresh
            # transform of the weights the planes were written from; the products follow the six-product rule
            n = nplanes * d.K * d.C
            pl = [(arr16(int(d.U3) + 2 * t * n, n).astype(np.uint32) << 16).view(np.float32).astype(np.float64) for t in range(3)]
            if d.U:
                U = arr(d.U, n).reshape(nplanes, d.K, d.C).astype(np.float64)
            else:
                src = getattr(self, "_u3_src", {}).get(int(d.U3))
                if src is None:
                    return self._fail("wino6_gemm: U3 planes nobody wrote")
                w_, K_, C_, r_, flip_ = src
                Gm = self._W6[r_][0]
                rr = self._geo6(r_)[0]
                gw = arr(w_, K_ * C_ * rr * rr)
                gw = gw.reshape(C_, K_, rr, rr).astype(np.float64).transpose(1, 0, 2, 3)[:, :, ::-1, ::-1] if flip_ else gw.reshape(K_, C_, rr, rr).astype(np.float64)
                U = np.einsum("ai,kcij,bj->abkc", Gm, gw, Gm).astype(np.float32).astype(np.float64).reshape(nplanes, d.K, d.C)
            if not np.array_equal((pl[0] + pl[1] + pl[2]).reshape(U.shape), U):
                return self._fail("wino6_gemm: the U3 planes are not the three-term split of U (stale planes?)")
            Mo = np.stack([contract(V[f].astype(np.float32), np.ascontiguousarray(U[f].T).astype(np.float32), 3) for f in range(nplanes)])
            arr(d.M, nplanes * T * d.K)[:] = Mo.reshape(-1).astype(np.float32)
            return 0
        if not d.U:
            return self._fail("wino6_gemm: U is required unless the split tile takes the launch")
        U = arr(d.U, nplanes * d.K * d.C).reshape(nplanes, d.K, d.C).astype(np.float64)
        arr(d.M, nplanes * T * d.K)[:] = np.einsum("ftc,fkc->ftk", V, U).reshape(-1).astype(np.float32)
        return 0

    def nirgan_conv_kernel_name(self, ref):
        return b"emulated_conv"

    def nirgan_wgrad_kernel_name(self, ref):
        return b"emulated_wgrad"

    def nirgan_conv_wgrad_pair_kernel_name(self, cref, wref):
        return b"emulated_pair"

    def nirgan_wino6_gemm_kernel_name(self, ref):
        return b"emulated_wino6_gemm"

    def nirgan_wino6_pair_kernel_name(self, cref, wref):
        return b"emulated_wino6_pair"

    def nirgan_wino6_gemm_wgrad_pair(self, cref, wref, stream=None):
        rc = self.nirgan_wino6_gemm(cref)
        return rc if rc else self.nirgan_wgrad_igemm(wref)

    def nirgan_wino6_output(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("wino6_out")
        B, H, W, K = d.B, d.H, d.W, d.K
        v = self._r6(d.r)
        r, mo, n = self._geo6(v)
        TH, TW = -(-H // mo), -(-W // mo)
        M = arr(d.M, n * n * B * TH * TW * K).reshape(n, n, B, TH, TW, K).astype(np.float64)
        AT = self._W6[v][2]
        Y = np.einsum("pa,albyxk,ql->bypxqk", AT, M, AT).reshape(B, mo * TH, mo * TW, K)
        if d.fuse_gz:
            # data gradient over the padded extent + the first pass of the consumer's instance-norm backward: fold the reflect halo
            # (rows, then columns), add the skip gradient, store g_a dense, leave the per-tile partial sums of g_z and g_z * z
            if v != 6 or d.bias or H < 6 or W < 6 or (H - 3) // mo != (H - 1) // mo or (W - 3) // mo != (W - 1) // mo:
                return self._fail("wino6_output: fused instance-norm backward: geometry")
            if d.fuse_part_elems < B * TH * TW * 2 * K:
                return self._fail("wino6_output: fuse_part too small")
            self.calls.append("wino6_out_inbwd")
            R = Y[:, :H, :W].copy()
            R[:, 2] += R[:, 0]
            R[:, H - 3] += R[:, H - 1]
            R[:, :, 2] += R[:, :, 0]
            R[:, :, W - 3] += R[:, :, W - 1]
            Hi, Wi = H - 2, W - 2
            ga = R[:, 1:H - 1, 1:W - 1]
            if d.fuse_g2:
                ga = ga + arr(d.fuse_g2, B * Hi * Wi * K).reshape(B, Hi, Wi, K)
            arr(d.fuse_gz, B * Hi * Wi * K).reshape(B, Hi, Wi, K)[:] = ga.astype(np.float32)
            yv = arr(d.fuse_y, B * Hi * Wi * K).reshape(B, Hi, Wi, K).astype(np.float64)
            z = (yv - arr(d.fuse_mean, B * K).reshape(B, 1, 1, K)) * arr(d.fuse_rstd, B * K).reshape(B, 1, 1, K)
            gzv = ga
            if d.fuse_act in (1, 2):
                gzv = np.where(z > 0, ga, ga * (0.0 if d.fuse_act == 1 else d.fuse_slope))
            grid = np.zeros((2, B, mo * TH, mo * TW, K))
            grid[0, :, 1:H - 1, 1:W - 1] = gzv
            grid[1, :, 1:H - 1, 1:W - 1] = gzv * z
            part = grid.reshape(2, B, TH, mo, TW, mo, K).sum((3, 5))                 # [2][B][TH][TW][K]
            arr(d.fuse_part, B * TH * TW * 2 * K).reshape(B, TH, TW, 2, K)[:] = part.transpose(1, 2, 3, 0, 4)
            return 0
        if d.stats_ws:
            # per tile {k = its first output, sum (o - k), sum (o - k)^2, count} over the stored outputs, without the bias
            if d.stats_ws_elems < B * TH * TW * 4 * K:
                return self._fail("wino6_output: stats_ws too small")
            valid = np.zeros((TH * mo, TW * mo), dtype=bool)
            valid[:H, :W] = True
            vt = valid.reshape(TH, mo, TW, mo)
            tiles = Y.reshape(B, TH, mo, TW, mo, K)
            k = tiles[:, :, 0, :, 0, :]                                              # [B][TH][TW][K]
            dl = (tiles - k[:, :, None, :, None, :]) * vt[None, :, :, :, :, None]
            st = arr(d.stats_ws, B * TH * TW * 4 * K).reshape(B, TH, TW, 4, K)
            st[:, :, :, 0] = k
            st[:, :, :, 1] = dl.sum((2, 4))
            st[:, :, :, 2] = (dl ** 2).sum((2, 4))
            st[:, :, :, 3] = vt.sum((1, 3))[None, :, :, None]
        bias = arr(d.bias, K)
        if bias is not None:
            Y = Y + bias
        arr(d.y, B * H * W * K).reshape(B, H, W, K)[:] = Y[:, :H, :W].astype(np.float32)
        return 0

    def nirgan_wino6_conv3x3(self, ref, stream=None):
        for fn in (self.nirgan_wino6_input, self.nirgan_wino6_gemm, self.nirgan_wino6_output):
            rc = fn(ref)
            if rc:
                return rc
        return 0

    def nirgan_wino6_dy(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("wino6_dy")
        B, H, W, K = d.B, d.H, d.W, d.K
        v = self._r6(d.r)
        if v not in self._W6:
            return self._fail("wino6_dy: variant")
        r, mo, n = self._geo6(v)
        TH, TW = -(-H // mo), -(-W // mo)
        if d.Yt_elems < n * n * B * TH * TW * K or d.dy_hp != H + 2 * d.dy_pad:
            return self._fail("wino6_dy: bad geometry / workspace")
        dy = arr(d.dy, B * d.dy_hp * d.dy_wp * K).reshape(B, d.dy_hp, d.dy_wp, K).astype(np.float64)
        z = np.zeros((B, mo * TH, mo * TW, K))
        z[:, :H, :W] = dy[:, d.dy_pad:d.dy_pad + H, d.dy_pad:d.dy_pad + W]
        tiles = np.stack([np.stack([z[:, i::mo, j::mo] for j in range(mo)], 0) for i in range(mo)], 0)   # [mo][mo][B][TH][TW][K]
        A = self._W6[v][2].T                                                                              # n x mo
        Yt = np.einsum("ia,abnyxk,jb->ijnyxk", A, tiles, A)
        arr(d.Yt, n * n * B * TH * TW * K)[:] = Yt.reshape(-1).astype(np.float32)
        return 0

    def nirgan_wino6_input_dy(self, cref, yref, stream=None):
        c, y = obj(cref), obj(yref)
        v = self._r6(c.r)
        r = self._geo6(v)[0]
        if c.x != y.dy or y.dy_pad != r - 1 or c.H != y.H + r - 1 or c.W != y.W + r - 1 or c.C != y.K or self._r6(y.r) != v:
            return self._fail("wino6_input_dy: the two descriptors do not describe the same output-gradient buffer")
        rc = self.nirgan_wino6_input(cref)
        return rc if rc else self.nirgan_wino6_dy(yref)

    def nirgan_wino6_input_dy_norm(self, cref, yref, nref, stream=None):
        """dY evaluated from the instance-norm backward descriptor (whose own call ran with dy = NULL: reductions only), then the two
        transforms.  The emulator materialises dY in the buffer the descriptors describe (the device never touches it)."""
        c, y, n = obj(cref), obj(yref), obj(nref)
        self.calls.append("wino6_dy_norm")
        if n.dy or not n.norm or n.B != c.B or n.C != c.C or n.H != y.H or n.W != y.W or self._r6(c.r) != 6 or c.C % 32:
            return self._fail("wino6_input_dy_norm: bad instance-norm descriptor")
        full = type(n)()
        C.memmove(C.byref(full), C.byref(n), C.sizeof(n))
        full.dy, full.d_hp, full.d_wp, full.d_pad = y.dy, y.dy_hp, y.dy_wp, y.dy_pad
        mark = len(self.calls)
        rc = self.nirgan_instnorm_bwd(full)
        del self.calls[mark:]                 # the device runs no second instance-norm call: keep the call log as the device's
        return rc if rc else self.nirgan_wino6_input_dy(cref, yref)

    def nirgan_wino6_wgrad_finish(self, slabs, nsplit, K, Cc, grad, accumulate, stream=None):
        return self.nirgan_wino6_wgrad_finish_r(slabs, nsplit, K, Cc, 3, grad, accumulate)

    def nirgan_wino6_wgrad_finish_r(self, slabs, nsplit, K, Cc, r, grad, accumulate, stream=None):
        self.calls.append("wino6_fin")
        v = self._r6(r)
        r, _, n = self._geo6(v)
        u = arr(slabs, n * n * nsplit * K * Cc).reshape(n, n, nsplit, K, Cc).astype(np.float64).sum(2)
        Gm = self._W6[v][0]
        g = np.einsum("ai,abkc,bj->kcij", Gm, u, Gm)
        o = arr(grad, K * Cc * r * r).reshape(K, Cc, r, r)
        if accumulate:
            o += g.astype(np.float32)
        else:
            o[:] = g.astype(np.float32)
        return 0

    def nirgan_wino6_wgrad_finish_batch(self, slabs, grads, n, nsplit, K, Cc, r, accumulate, stream=None):
        if not (1 <= n <= 16):
            return self._fail("wino6_wgrad_finish_batch: 1..16 layers")
        sp = C.cast(slabs, C.POINTER(C.c_void_p))
        gp = C.cast(grads, C.POINTER(C.c_void_p))
        self.calls.append("wino6_fin_batch")
        for i in range(n):
            rc = self.nirgan_wino6_wgrad_finish_r(sp[i], nsplit, K, Cc, r, gp[i], accumulate)
            if rc:
                return rc
        return 0

    # ------------------------------------------------------------------ tiled inference
    def nirgan_tile_count(self, B, H, W, tile, margin):
        if B <= 0 or H <= 0 or W <= 0 or tile <= 0 or margin < 0 or 2 * margin >= tile:
            return 0
        core = tile - 2 * margin
        return B * (-(-H // core)) * (-(-W // core))

    def _tiling(self, B, H, W, tile, margin, first, n, who):
        core = tile - 2 * margin
        nth, ntw = -(-H // core), -(-W // core)
        if not (margin < H and margin < W and nth * core - H + margin < H and ntw * core - W + margin < W):
            return None, self._fail(f"{who}: the reflected border is wider than the scene")
        if first < 0 or n <= 0 or first + n > B * nth * ntw:
            return None, self._fail(f"{who}: tile range")
        return (core, nth, ntw), 0

    def nirgan_tile_gather(self, scene, B, Cc, H, W, tile, margin, first, n, tiles, stream=None):
        self.calls.append("tile_gather")
        geo, rc = self._tiling(B, H, W, tile, margin, first, n, "tile_gather")
        if rc:
            return rc
        core, nth, ntw = geo
        src = arr(scene, B * Cc * H * W).reshape(B, Cc, H, W)
        dst = arr(tiles, n * Cc * tile * tile).reshape(n, Cc, tile, tile)

        def refl(i, m):
            i = np.abs(i)
            return np.where(i >= m, 2 * m - 2 - i, i)
        for k in range(n):
            idx = first + k
            b, t = divmod(idx, nth * ntw)
            ti, tj = divmod(t, ntw)
            hh = refl(ti * core + np.arange(tile) - margin, H)
            ww = refl(tj * core + np.arange(tile) - margin, W)
            dst[k] = src[b][:, hh][:, :, ww]
        return 0

    def nirgan_tile_scatter(self, tiles, B, Cc, H, W, tile, margin, first, n, scene, stream=None):
        self.calls.append("tile_scatter")
        geo, rc = self._tiling(B, H, W, tile, margin, first, n, "tile_scatter")
        if rc:
            return rc
        core, nth, ntw = geo
        src = arr(tiles, n * Cc * tile * tile).reshape(n, Cc, tile, tile)
        dst = arr(scene, B * Cc * H * W).reshape(B, Cc, H, W)
        for k in range(n):
            b, t = divmod(first + k, nth * ntw)
            ti, tj = divmod(t, ntw)
            h0, w0 = ti * core, tj * core
            h1, w1 = min(h0 + core, H), min(w0 + core, W)
            dst[b, :, h0:h1, w0:w1] = src[k, :, margin:margin + h1 - h0, margin:margin + w1 - w0]
        return 0

    # ------------------------------------------------------------------ histogram matching
    def nirgan_hist_match_ws_bytes(self, B, N):
        P = 2048
        while P < N:
            P *= 2
        return B * P * 12

    def nirgan_hist_match(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("hist_match")
        if d.ws_bytes < self.nirgan_hist_match_ws_bytes(d.B, d.N):
            return self._fail("hist_match: workspace too small")
        img = arr(d.image, d.B * d.N).reshape(d.B, d.N)
        tmp = arr(d.reference, d.B * d.N).reshape(d.B, d.N)
        out = arr(d.out, d.B * d.N).reshape(d.B, d.N)
        for b in range(d.B):
            sv, lookup, sc = np.unique(img[b], return_inverse=True, return_counts=True)
            tv, tc = np.unique(tmp[b], return_counts=True)
            vals = np.interp(np.cumsum(sc) / d.N, np.cumsum(tc) / d.N, tv)
            out[b] = vals[lookup].astype(np.float32)
        return 0

    # ------------------------------------------------------------------ image metrics
    def nirgan_image_metrics_ws_elems(self, planes, H, W):
        return planes * ((H + 31) // 32) * ((W + 31) // 32) * 3

    def nirgan_image_metrics(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("metrics")
        r = d.window // 2
        if d.window % 2 == 0 or d.window > 11 or d.H <= r or d.W <= r:
            return self._fail("image_metrics: bad window")
        if d.ws_elems < self.nirgan_image_metrics_ws_elems(d.planes, d.H, d.W):
            return self._fail("image_metrics: workspace too small")
        n = d.planes * d.H * d.W
        a = torch.from_numpy(arr(d.pred, n).reshape(d.planes, 1, d.H, d.W).astype(np.float64))
        b = torch.from_numpy(arr(d.target, n).reshape(d.planes, 1, d.H, d.W).astype(np.float64))
        x = torch.arange(d.window, dtype=torch.float64) - r
        k = torch.exp(-x * x / (2.0 * d.sigma ** 2))
        k = (k / k.sum())
        k2 = (k[:, None] * k[None, :])[None, None]

        def filt(t):
            return F.conv2d(F.pad(t, (r, r, r, r), mode="reflect"), k2)
        c1, c2 = (0.01 * d.max_val) ** 2, (0.03 * d.max_val) ** 2
        mu1, mu2 = filt(a), filt(b)
        s1, s2, s12 = filt(a * a) - mu1 * mu1, filt(b * b) - mu2 * mu2, filt(a * b) - mu1 * mu2
        ssim = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2) + d.eps)
        out = arr(d.means, 3)
        out[0], out[1], out[2] = (a - b).abs().mean().item(), ((a - b) ** 2).mean().item(), ssim.mean().item()
        return 0

    # ------------------------------------------------------------------ SSIM loss (value + gradient through torch autograd, fp64)
    def nirgan_ssim_loss_ws_elems(self, planes, H, W, window):
        if planes <= 0 or H <= 0 or W <= 0 or window < 1 or window > 11 or window % 2 == 0:
            return 0
        r = window // 2
        return 3 * planes * H * W + 3 * planes * (H + 2 * r) * (W + 2 * r) + planes * ((H + 31) // 32) * ((W + 31) // 32)

    def nirgan_ssim_loss(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("ssim_loss")
        r = d.window // 2
        if d.window % 2 == 0 or d.window > 11 or d.H <= r or d.W <= r:
            return self._fail("ssim_loss: bad window")
        if d.ws_elems < self.nirgan_ssim_loss_ws_elems(d.planes, d.H, d.W, d.window):
            return self._fail("ssim_loss: workspace too small")
        n = d.planes * d.H * d.W
        with torch.enable_grad():                 # (callers may sit inside an autograd.Function.forward)
            a = torch.from_numpy(arr(d.pred, n).reshape(d.planes, 1, d.H, d.W).astype(np.float64)).requires_grad_(True)
            b = torch.from_numpy(arr(d.target, n).reshape(d.planes, 1, d.H, d.W).astype(np.float64))
            x = torch.arange(d.window, dtype=torch.float64) - r
            k = torch.exp(-x * x / (2.0 * d.sigma ** 2))
            k = (k / k.sum())
            k2 = (k[:, None] * k[None, :])[None, None]

            def filt(t):
                return F.conv2d(F.pad(t, (r, r, r, r), mode="reflect"), k2)
            c1, c2 = (0.01 * d.max_val) ** 2, (0.03 * d.max_val) ** 2
            mu1, mu2 = filt(a), filt(b)
            s1, s2, s12 = filt(a * a) - mu1 * mu1, filt(b * b) - mu2 * mu2, filt(a * b) - mu1 * mu2
            ssim = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2) + d.eps)
            v = 1.0 - ssim.mean()
            g = torch.autograd.grad(v, a)[0] if d.grad_pred else None
        if d.value:
            arr(d.value, 1)[0] = v.item()
        if d.loss:
            arr(d.loss, 1)[0] += d.weight * v.item()
        if d.grad_pred:
            arr(d.grad_pred, n)[:] += (d.weight * g).reshape(-1).numpy().astype(np.float32)
        return 0

    # ------------------------------------------------------------------ emd_loss (softmax -> cumsum -> mean |difference|)
    def nirgan_emd_loss_ws_bytes(self, B, N, with_grad):
        return 0 if B <= 0 or N <= 0 else B * 8 + (B * N * 4 if with_grad else 0)

    def nirgan_emd_loss(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("emd_loss")
        if d.B <= 0 or d.N <= 0 or d.N >= 1 << 24:
            return self._fail("emd_loss: out of range")
        if d.ws_bytes < self.nirgan_emd_loss_ws_bytes(d.B, d.N, bool(d.grad_pred)):
            return self._fail("emd_loss: workspace too small")
        n = d.B * d.N
        with torch.enable_grad():
            a = torch.from_numpy(arr(d.pred, n).reshape(d.B, d.N).astype(np.float64)).requires_grad_(True)
            b = torch.from_numpy(arr(d.target, n).reshape(d.B, d.N).astype(np.float64))
            v = (torch.cumsum(torch.softmax(a, 1), 1) - torch.cumsum(torch.softmax(b, 1), 1)).abs().mean()
            g = torch.autograd.grad(v, a)[0] if d.grad_pred else None
        if d.value:
            arr(d.value, 1)[0] = v.item()
        if d.loss:
            arr(d.loss, 1)[0] += d.weight * v.item()
        if d.grad_pred:
            arr(d.grad_pred, n)[:] += (d.weight * g).reshape(-1).numpy().astype(np.float32)
        return 0

    # ------------------------------------------------------------------ instance norm
    def nirgan_instnorm_ws_elems(self, B, H, W, Cc):
        return B * _in_nchunk(B, H * W, Cc) * 2 * Cc + B * 2 * Cc

    @staticmethod
    def _act(z, act, slope):
        if act == 1:
            return np.maximum(z, 0)
        if act == 2:
            return np.where(z > 0, z, z * slope)
        return z

    def nirgan_instnorm_fwd(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("in_fwd")
        B, H, W, Cc = d.B, d.H, d.W, d.C
        if d.out and (d.o_hp != H + 2 * d.o_pad or d.o_wp != W + 2 * d.o_pad):
            return self._fail("in_fwd: geometry")
        y = load_y(d.y, B * H * W * Cc, d.y_bf16).reshape(B, H * W, Cc).astype(np.float64)
        if d.norm:
            if d.stats_chunks > 0:
                # partial sums left by the producer: [B][chunks][4][C] = {k, sum (v - k), sum (v - k)^2, count}, re-based onto chunk 0's shift
                n = d.stats_chunks
                if d.ws_elems < B * n * 4 * Cc:
                    return self._fail("in_fwd: ws too small")
                self.calls.append("in_fwd_pre")
                part = arr(d.ws, B * n * 4 * Cc).reshape(B, n, 4, Cc).astype(np.float64)
                K0 = part[:, :1, 0]
                dk = part[:, :, 0] - K0
                S1 = (part[:, :, 1] + part[:, :, 3] * dk).sum(1)
                S2 = (part[:, :, 2] + 2 * dk * part[:, :, 1] + part[:, :, 3] * dk * dk).sum(1)
                k = arr(d.stats_shift, Cc).astype(np.float64) if d.stats_shift else np.zeros(Cc)
                m = S1 / (H * W)
                mean = k + K0[:, 0] + m
                var = np.maximum(S2 / (H * W) - m * m, 0.0)
            else:
                if d.ws_elems < self.nirgan_instnorm_ws_elems(B, H, W, Cc):
                    return self._fail("in_fwd: ws too small")
                mean = y.mean(1)
                var = y.var(1)
            rstd = 1.0 / np.sqrt(var + d.eps)
            arr(d.mean, B * Cc).reshape(B, Cc)[:] = mean
            arr(d.rstd, B * Cc).reshape(B, Cc)[:] = rstd
            z = (y - mean[:, None]) * rstd[:, None]
        else:
            z = y
        if not d.out and not d.out_bf16:   # statistics only
            return 0 if d.norm else self._fail("in_fwd: null pointer")
        a = self._act(z, d.act, d.slope).reshape(B, H, W, Cc)
        if d.residual:
            r = arr(d.residual, B * d.r_hp * d.r_wp * Cc).reshape(B, d.r_hp, d.r_wp, Cc)
            a = a + r[:, d.r_pad:d.r_pad + H, d.r_pad:d.r_pad + W]
        # out = NULL with a twin: only the bf16 twin is stored (its halo, untouched by KEEP borders, is zero like the fp32 buffer's would be)
        out = arr(d.out, B * d.o_hp * d.o_wp * Cc).reshape(B, d.o_hp, d.o_wp, Cc) if d.out else np.zeros((B, d.o_hp, d.o_wp, Cc), np.float32)
        P = d.o_pad
        out[:, P:P + H, P:P + W] = a
        if d.border == 1 and P > 0:
            hh = reflect(np.arange(d.o_hp) - P, H)
            ww = reflect(np.arange(d.o_wp) - P, W)
            out[:] = a[:, hh][:, :, ww]
        mirror_twin(d.out_bf16, out)      # the device mirrors exactly the stores it makes; untouched (zero) halo stays zero in both
        return 0

    def nirgan_instnorm_bwd(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("in_bwd")
        B, H, W, Cc = d.B, d.H, d.W, d.C
        ga = np.zeros((B, H, W, Cc), dtype=np.float64)
        pre = d.norm and d.sums_chunks > 0
        if pre:
            # the producer left the first pass's partial sums in ws: nirgan_wino6_output in its fused mode (g_a then sits in gsum_out), or
            # a convolution launch with fuse_* (the gradient is g itself: no fold, no second gradient)
            if d.ws_elems < B * d.sums_chunks * 2 * Cc + B * 2 * Cc:
                return self._fail("in_bwd: sums_chunks needs a large enough ws")
            if not d.gsum_out and (not d.g or d.g_fold or d.g2):
                return self._fail("in_bwd: sums_chunks needs gsum_out or a plain g")
            self.calls.append("in_bwd_pre")
            psum = arr(d.ws, B * d.sums_chunks * 2 * Cc).reshape(B, d.sums_chunks, 2, Cc).astype(np.float64).sum(1) / (H * W)
            if d.gsum_out:
                ga = arr(d.gsum_out, B * H * W * Cc).reshape(B, H, W, Cc).astype(np.float64)
            else:
                g = load_y(d.g, B * d.g_hp * d.g_wp * Cc, d.g_bf16).reshape(B, d.g_hp, d.g_wp, Cc).astype(np.float64)
                ga = g[:, d.g_pad:d.g_pad + H, d.g_pad:d.g_pad + W].copy()
        elif d.g:
            if d.g_hp != H + 2 * d.g_pad or d.g_wp != W + 2 * d.g_pad:
                return self._fail("in_bwd: g geometry")
            g = load_y(d.g, B * d.g_hp * d.g_wp * Cc, d.g_bf16).reshape(B, d.g_hp, d.g_wp, Cc).astype(np.float64)
            P = d.g_pad
            if d.g_fold:
                hh = reflect(np.arange(d.g_hp) - P, H)
                ww = reflect(np.arange(d.g_wp) - P, W)
                tmp = np.zeros((B, H, d.g_wp, Cc))
                np.add.at(tmp, (slice(None), hh), g)
                tmp2 = np.zeros((B, H, W, Cc))
                np.add.at(tmp2, (slice(None), slice(None), ww), tmp)
                ga += tmp2
            else:
                ga += g[:, P:P + H, P:P + W]
        if d.g2 and not pre:
            ga += arr(d.g2, B * H * W * Cc).reshape(B, H, W, Cc)
        if d.gsum_out and not pre:
            arr(d.gsum_out, B * H * W * Cc).reshape(B, H, W, Cc)[:] = ga
        z = None
        if d.norm or d.act in (1, 2):
            y = load_y(d.y, B * H * W * Cc, d.y_bf16).reshape(B, H * W, Cc).astype(np.float32)
            if d.norm:
                mean = arr(d.mean, B * Cc).reshape(B, 1, Cc)
                rstd = arr(d.rstd, B * Cc).reshape(B, 1, Cc)
                z = ((y - mean) * rstd)            # fp32, as the kernel (and the forward) computes it
            else:
                z = y
        gz = ga
        if d.act in (1, 2):
            gz = np.where(z.reshape(B, H, W, Cc) > 0, ga, ga * (0.0 if d.act == 1 else d.slope))
        if d.norm:
            zz = z.astype(np.float64)
            gzf = gz.reshape(B, H * W, Cc)
            m1, m2 = gzf.mean(1, keepdims=True), (gzf * zz).mean(1, keepdims=True)
            if pre:                          # the means as the producer's partial sums give them
                m1, m2 = psum[:, 0][:, None, :], psum[:, 1][:, None, :]
            dy = rstd * (gzf - m1 - zz * m2)
            dy = dy.reshape(B, H, W, Cc)
        else:
            dy = gz
        if not d.dy and not d.dy_bf16:     # reductions only (with norm): a consumer evaluates dy on the fly; gsum_out was written above
            return 0 if d.norm else self._fail("in_bwd: dy missing")
        if not d.dy and not d.norm:
            return self._fail("in_bwd: dy missing")
        o = arr(d.dy, B * d.d_hp * d.d_wp * Cc).reshape(B, d.d_hp, d.d_wp, Cc) if d.dy else np.zeros((B, d.d_hp, d.d_wp, Cc), np.float32)
        o[:, d.d_pad:d.d_pad + H, d.d_pad:d.d_pad + W] = dy
        mirror_twin(d.dy_bf16, o)
        if d.dbias:
            arr(d.dbias, Cc)[:] += dy.sum((0, 1, 2))
        return 0

    # ------------------------------------------------------------------ layout
    def nirgan_nchw_to_halo(self, src, B, Cs, H, W, dst, cs, c0, pad1, pad2, mode, stream=None):
        self.calls.append("nchw_to_halo")
        P = pad1 + pad2
        Hp, Wp = H + 2 * P, W + 2 * P
        s = arr(src, B * Cs * H * W).reshape(B, Cs, H, W)
        o = arr(dst, B * Hp * Wp * cs).reshape(B, Hp, Wp, cs)
        if mode == 1:
            hh = reflect(reflect(np.arange(Hp) - pad2, H + 2 * pad1) - pad1, H)
            ww = reflect(reflect(np.arange(Wp) - pad2, W + 2 * pad1) - pad1, W)
            o[..., c0:c0 + Cs] = s[:, :, hh][:, :, :, ww].transpose(0, 2, 3, 1)
        else:
            o[:, P:P + H, P:P + W, c0:c0 + Cs] = s.transpose(0, 2, 3, 1)
        return 0

    def nirgan_conv_channel_dgrad(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("chan_dgrad")
        OH, OW = (d.H + 2 * d.pad - d.k) // d.stride + 1, (d.W + 2 * d.pad - d.k) // d.stride + 1
        if d.dy_hp != OH + 2 * d.dy_pad or d.dy_wp != OW + 2 * d.dy_pad:
            return self._fail("chan_dgrad: geometry")
        dy = arr(d.dy, d.B * d.dy_hp * d.dy_wp * d.C).reshape(d.B, d.dy_hp, d.dy_wp, d.C)
        dy = dy[:, d.dy_pad:d.dy_pad + OH, d.dy_pad:d.dy_pad + OW].transpose(0, 3, 1, 2)
        w = arr(d.w, d.C * d.cin * d.k * d.k).reshape(d.C, d.cin, d.k, d.k)[:, d.channel:d.channel + 1]
        x = torch.zeros(d.B, 1, d.H, d.W, requires_grad=True)
        with torch.enable_grad():
            y = F.conv2d(x, torch.from_numpy(w.copy()), stride=d.stride, padding=d.pad)
            y.backward(torch.from_numpy(np.ascontiguousarray(dy)))
        arr(d.out, d.B * d.H * d.W)[:] = x.grad.reshape(-1).numpy()
        return 0

    def nirgan_tap_gather(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("tap_gather")
        q = arr(d.q, d.B * d.q_hp * d.q_wp * d.q_cs).reshape(d.B, d.q_hp, d.q_wp, d.q_cs)
        acc = np.zeros((d.B, d.OH, d.OW), dtype=np.float64)
        for t in range(d.ntaps):
            acc += q[:, d.tap_dh[t]:d.tap_dh[t] + d.OH, d.tap_dw[t]:d.tap_dw[t] + d.OW, t]
        if d.bias:
            acc += arr(d.bias, 1)[0]
        if d.act == 3:
            acc = np.tanh(acc)
        c = d.crop
        arr(d.dst, d.B * (d.OH - 2 * c) * (d.OW - 2 * c)).reshape(d.B, d.OH - 2 * c, d.OW - 2 * c)[:] = acc[:, c:d.OH - c, c:d.OW - c]
        return 0

    def nirgan_tap_scatter(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("tap_scatter")
        c = d.crop
        H2, W2 = d.OH - 2 * c, d.OW - 2 * c
        dz = arr(d.dout, d.B * H2 * W2).reshape(d.B, H2, W2).astype(np.float64)
        if d.act == 3:
            y = arr(d.out, d.B * H2 * W2).reshape(d.B, H2, W2)
            dz = dz * (1 - y.astype(np.float64) ** 2)
        full = np.zeros((d.B, d.OH, d.OW))
        full[:, c:c + H2, c:c + W2] = dz
        dq = arr(d.dq, d.B * d.q_hp * d.q_wp * d.q_cs).reshape(d.B, d.q_hp, d.q_wp, d.q_cs)
        dq[:] = 0
        for t in range(d.ntaps):
            dq[:, d.tap_dh[t]:d.tap_dh[t] + d.OH, d.tap_dw[t]:d.tap_dw[t] + d.OW, t] = full
        if d.dbias:
            arr(d.dbias, 1)[0] += dz.sum()
        return 0

    # ------------------------------------------------------------------ direct Conv2d(64, 1, 7) (+tanh, crop)
    def nirgan_endconv_dz_elems(self, B, OH, OW):
        return B * (OH + 12) * ((OW + 12 + 3) // 4 * 4 + 8) if min(B, OH, OW) > 0 else 0

    def nirgan_endconv_ws_elems(self, B, OH, OW):
        return min((B * (OH + 6) + 3) // 4, 512) * 49 * 64 if min(B, OH) > 0 else 0

    def _endconv(self, ref, what):
        d = obj(ref)
        if d.C != 64 or d.k != 7:
            return None, self._fail(f"{what}: the direct kernels cover Conv2d(64, 1, 7) only")
        if d.x_hp != d.OH + 6 or d.x_wp != d.OW + 6 or d.B <= 0 or d.OH <= 2 * d.crop or d.OW <= 2 * d.crop:
            return None, self._fail(f"{what}: bad shape")
        self.calls.append(what)
        return d, 0

    def _endconv_dz_image(self, d):
        S = (d.x_wp + 6 + 3) // 4 * 4 + 8
        return arr(d.dz, d.B * (d.x_hp + 6) * S).reshape(d.B, d.x_hp + 6, S)

    def nirgan_endconv_fwd(self, ref, stream=None):
        d, rc = self._endconv(ref, "endconv_fwd")
        if d is None:
            return rc
        x = arr(d.x, d.B * d.x_hp * d.x_wp * 64).reshape(d.B, d.x_hp, d.x_wp, 64).astype(np.float64)
        w = arr(d.w, 49 * 64).reshape(7, 7, 64).astype(np.float64)
        z = np.zeros((d.B, d.OH, d.OW))
        for a in range(7):
            for b in range(7):
                z += x[:, a:a + d.OH, b:b + d.OW, :] @ w[a, b]
        if d.bias:
            z += arr(d.bias, 1)[0]
        if d.act == 3:
            z = np.tanh(z)
        c = d.crop
        arr(d.out, d.B * (d.OH - 2 * c) * (d.OW - 2 * c)).reshape(d.B, d.OH - 2 * c, d.OW - 2 * c)[:] = z[:, c:d.OH - c, c:d.OW - c]
        return 0

    def nirgan_endconv_dz(self, ref, stream=None):
        d, rc = self._endconv(ref, "endconv_dz")
        if d is None:
            return rc
        c = d.crop
        H2, W2 = d.OH - 2 * c, d.OW - 2 * c
        dz = arr(d.dout, d.B * H2 * W2).reshape(d.B, H2, W2).astype(np.float64)
        if d.act == 3:
            dz = dz * (1 - arr(d.out, d.B * H2 * W2).reshape(d.B, H2, W2).astype(np.float64) ** 2)
        img = self._endconv_dz_image(d)
        img[:] = 0
        img[:, 6 + c:6 + c + H2, 6 + c:6 + c + W2] = dz
        if d.gbias:
            arr(d.gbias, 1)[0] += dz.sum()
        return 0

    def nirgan_endconv_dgrad(self, ref, stream=None):
        d, rc = self._endconv(ref, "endconv_dgrad")
        if d is None:
            return rc
        img = self._endconv_dz_image(d).astype(np.float64)
        w = arr(d.w, 49 * 64).reshape(7, 7, 64).astype(np.float64)
        gx = np.zeros((d.B, d.x_hp, d.x_wp, 64))
        for a in range(7):
            for b in range(7):
                gx += img[:, 6 - a:6 - a + d.x_hp, 6 - b:6 - b + d.x_wp, None] * w[a, b]
        arr(d.gx, gx.size).reshape(gx.shape)[:] = gx
        return 0

    def nirgan_endconv_wgrad(self, ref, stream=None):
        d, rc = self._endconv(ref, "endconv_wgrad")
        if d is None:
            return rc
        img = self._endconv_dz_image(d).astype(np.float64)
        x = arr(d.x, d.B * d.x_hp * d.x_wp * 64).reshape(d.B, d.x_hp, d.x_wp, 64).astype(np.float64)
        gw = np.zeros((7, 7, 64))
        for a in range(7):
            for b in range(7):
                gw[a, b] = np.einsum("bhw,bhwc->c", img[:, 6 - a:6 - a + d.x_hp, 6 - b:6 - b + d.x_wp], x)
        arr(d.gw, 49 * 64).reshape(64, 49)[:] = gw.reshape(49, 64).T
        return 0

    # ------------------------------------------------------------------ losses
    def nirgan_lsgan(self, pred, n, target, weight, loss_out, grad, stream=None):
        self.calls.append("lsgan")
        p = arr(pred, n).astype(np.float64)
        arr(loss_out, 1)[0] += weight * np.mean((p - target) ** 2)
        if grad:
            arr(grad, n)[:] = weight * 2 * (p - target) / n
        return 0

    def nirgan_pix_loss(self, ref, stream=None):
        with torch.enable_grad():
            return self._pix_loss(obj(ref))

    def _pix_loss(self, d):
        self.calls.append("pix_loss")
        B, H, W = d.B, d.H, d.W
        n = B * H * W
        # fp32 like the kernel: the indices are singular where pred + band ~ 0 (pred comes out of tanh)
        if not d.rgb and (d.log_all or any(w != 0 for w in (d.w_ndvi, d.w_ndwi, d.w_gndvi, d.w_savi, d.w_msavi, d.w_evi))):
            return self._fail("pix_loss: the spectral indices need rgb")
        rgb = torch.from_numpy(arr(d.rgb, 3 * n).reshape(B, 3, H, W).copy()) if d.rgb else torch.zeros(B, 3, H, W)
        x = torch.from_numpy(arr(d.nir, n).reshape(B, 1, H, W).copy())
        y = torch.from_numpy(arr(d.pred, n).reshape(B, 1, H, W).copy()).requires_grad_(True)
        R, Gc, Bl = rgb[:, 0:1], rgb[:, 1:2], rgb[:, 2:3]
        w = [d.w_l1, d.w_ndvi, d.w_ndwi, d.w_gndvi, d.w_savi, d.w_msavi, d.w_evi]
        crit = (lambda a, b: (a - b).abs().sum()) if d.criterion == 0 else (lambda a, b: ((a - b) ** 2).sum())

        def idx(k, v):
            if k == 1:
                return (v - R) / (v + R + 1e-6)
            if k == 2:
                return (v - Gc) / (v + Gc + 1e-6)
            if k == 3:
                return (v - Gc) / ((v - R) / (v + R) + Gc)
            if k == 4:
                return 1.5 * (v - R) / (v + R + 0.5)
            if k == 5:
                return (2 * v + 1 - torch.sqrt((2 * v + 1) ** 2 - 8 * (v - R))) / 2
            return 2.5 * ((v - R) / ((v + 6) * (R - 7.5) * (Bl + 1) + 1e-6))

        sums = arr(d.sums, 7)
        total = 0.0
        s0 = (y - x).abs().sum()
        sums[0] += float(s0.detach())
        total = total + w[0] * s0
        for k in range(1, 7):
            if d.log_all or w[k] != 0:
                sk = crit(idx(k, x), idx(k, y))
                sums[k] += float(sk.detach())
                if w[k] != 0:
                    total = total + w[k] * sk
        if d.grad_pred:
            (total / n).backward()
            g = y.grad.numpy().reshape(-1).copy()
            if d.extra:
                g += d.extra_scale * arr(d.extra, n * d.extra_cs)[d.extra_c::d.extra_cs]
            arr(d.grad_pred, n)[:] = g
        return 0

    # ------------------------------------------------------------------ elementwise
    def nirgan_adam(self, p, g, m, v, n, lr, b1, b2, eps, step, stream=None):
        self.calls.append("adam")
        P, G_, M, V = arr(p, n), arr(g, n), arr(m, n), arr(v, n)
        M[:] = M * np.float32(b1) + np.float32(1 - b1) * G_
        V[:] = V * np.float32(b2) + np.float32(1 - b2) * G_ * G_
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        P[:] -= np.float32(lr / bc1) * (M / (np.sqrt(V) / np.float32(np.sqrt(bc2)) + np.float32(eps)))
        return 0

    def nirgan_bilinear_fwd(self, src, B, SH, SW, dst, OH, OW, stream=None):
        self.calls.append("bilinear_fwd")
        s = torch.from_numpy(arr(src, B * SH * SW).reshape(B, 1, SH, SW).copy())
        o = F.interpolate(s, size=(OH, OW), mode="bilinear", align_corners=False)
        arr(dst, B * OH * OW)[:] = o.reshape(-1).numpy()
        return 0

    def nirgan_bilinear_bwd(self, ddst, B, OH, OW, dsrc, SH, SW, stream=None):
        self.calls.append("bilinear_bwd")
        with torch.enable_grad():
            s = torch.zeros(B, 1, SH, SW, requires_grad=True)
            o = F.interpolate(s, size=(OH, OW), mode="bilinear", align_corners=False)
            o.backward(torch.from_numpy(arr(ddst, B * OH * OW).reshape(B, 1, OH, OW).copy()))
        arr(dsrc, B * SH * SW)[:] = s.grad.reshape(-1).numpy()
        return 0

    def nirgan_inject_fwd(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("inject_fwd")
        B, H, W, Cc = d.B, d.H, d.W, d.C
        z = arr(d.z, B * H * W * Cc).reshape(B, H, W, Cc)
        e = arr(d.e, B * H * W).reshape(B, H, W, 1)
        s = arr(d.scale, 1)[0] if d.scale else 1.0
        v = z * (1 + s * e) if d.style == 0 else z + s * e
        o = arr(d.out, B * d.o_hp * d.o_wp * Cc).reshape(B, d.o_hp, d.o_wp, Cc)
        o[:, d.o_pad:d.o_pad + H, d.o_pad:d.o_pad + W] = np.maximum(v, 0)
        return 0

    def nirgan_inject_bwd(self, ref, stream=None):
        d = obj(ref)
        self.calls.append("inject_bwd")
        B, H, W, Cc = d.B, d.H, d.W, d.C
        g = arr(d.g, B * H * W * Cc).reshape(B, H, W, Cc).astype(np.float64)
        a = arr(d.a, B * d.a_hp * d.a_wp * Cc).reshape(B, d.a_hp, d.a_wp, Cc)[:, d.a_pad:d.a_pad + H, d.a_pad:d.a_pad + W]
        z = arr(d.z, B * H * W * Cc).reshape(B, H, W, Cc).astype(np.float64)
        e = arr(d.e, B * H * W).reshape(B, H, W, 1).astype(np.float64)
        s = float(arr(d.scale, 1)[0]) if d.scale else 1.0
        gm = np.where(a > 0, g, 0.0)
        if d.style == 0:
            dz, de, ds = gm * (1 + s * e), (gm * z * s).sum(-1), (gm * z * e).sum()
        else:
            dz, de, ds = gm, (gm * s).sum(-1), (gm * e).sum()
        arr(d.dz, B * H * W * Cc).reshape(B, H, W, Cc)[:] = dz
        arr(d.de, B * H * W).reshape(B, H, W)[:] = de
        if d.dscale:
            arr(d.dscale, 1)[0] += ds
        return 0

    def nirgan_colsum(self, x, rows, cols, out, accumulate, stream=None):
        s = arr(x, rows * cols).reshape(rows, cols).sum(0)
        o = arr(out, cols)
        o[:] = o + s if accumulate else s
        return 0

    def nirgan_param_scale_fwd(self, x, param, out, n, stream=None):
        arr(out, n)[:] = arr(x, n) * arr(param, 1)[0]
        return 0

    def nirgan_param_scale_bwd(self, gout, x, param, gx, dparam, ws, ws_elems, n, stream=None):
        g = arr(gout, n)
        arr(gx, n)[:] = g * arr(param, 1)[0]
        arr(dparam, 1)[0] += np.float32(np.dot(g.astype(np.float64), arr(x, n).astype(np.float64)))
        return 0

    def nirgan_fill(self, dst, n, value, stream=None):
        arr(dst, n)[:] = value
        return 0

    def nirgan_axpy(self, y, x, n, alpha, stream=None):
        arr(y, n)[:] += alpha * arr(x, n)
        return 0

    def nirgan_run_plan(self, entries, n, stream=None):
        return self._fail("run_plan is not emulated")
