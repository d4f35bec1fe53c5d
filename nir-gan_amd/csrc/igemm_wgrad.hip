// Implicit-GEMM weight gradient on the gfx950 matrix pipe, deterministic split over pixels.
//
//   slab[s][n][J] = sum_{m in split s} P[pix(m)][n] * Q[pix(m)*q_stride + tap(J)][c(J)],  J = t*run + c
//   GEMM view: rows n (TN = 128 or 64 per block), columns J (128 per block), reduction over
//   pixels m in steps of 32.  4 waves as 2x2; fragments come straight from the [m][n] / [m][J]
//   LDS images: for v_mfma_f32_32x32x2_f32 the two k lanes-halves read pixel rows 2kk and
//   2kk+1, and one ds_read_b64 hands a lane two adjacent n (or J): MFMA tile e covers rows
//   n = base + 2*i + e (a stride-2 interleave, undone at the store).  Both images are filled
//   by LDS-DMA, rows are >= 256 B so no swizzle is needed (a half-wave reads 256 contiguous
//   bytes).  Two LDS stages, one barrier per 32-pixel step.
//   The split partial sums go to slabs (plain stores) and are summed by reduce_rows in a fixed
//   order, so gradients are bitwise reproducible run to run.
#include "igemm_tiles.h"
#include "igemm_tile256.h"
#include "igemm_x3.h"
#include <cstdlib>

namespace {

template <int TN, int PREC = 0>
__global__ __launch_bounds__(256, 2) void wgrad_igemm_kernel(const ng::WgradParams p) {
    __shared__ __attribute__((aligned(16))) char st0[32 * (TN + 128) * 4];
    __shared__ __attribute__((aligned(16))) char st1[32 * (TN + 128) * 4];
    ng::wgrad_tile<TN, PREC>(p, blockIdx.x, st0, st1);
}

// weight gradient over the producers' bf16 twins (bf16 operand mode, N > 64)
__global__ __launch_bounds__(256, 2) void wgrad_persist_kernel(const ng::WgradParams p) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    ng::wgrad_persist(p, blockIdx.x, gridDim.x, lds);
}

__global__ __launch_bounds__(256, 2) void wgrad_igemm16_kernel(const ng::WgradParams p) {
    __shared__ __attribute__((aligned(16))) char st0[32768];
    __shared__ __attribute__((aligned(16))) char st1[32768];
    ng::wgrad_tile16(p, blockIdx.x, st0, st1);
}


// horizontally fused launch: the data-gradient tiles of a stride-1 convolution followed by the tiles of its
// weight gradient (both consume the same dY).  One grid: the weight-gradient blocks fill the partly empty
// last round of the data-gradient, and the other way round.
template <int PREC = 0, bool WB16 = false, bool AB16 = false, bool TW16 = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pair_kernel(const ng::ConvParams cp, const ng::WgradParams wp, const int conv_blocks) {
    __shared__ __attribute__((aligned(16))) char st0[32768];
    __shared__ __attribute__((aligned(16))) char st1[32768];
    if (int(blockIdx.x) < conv_blocks)
        ng::conv_tile<128, PREC, WB16, AB16>(cp, blockIdx.x, st0, st1);
    else if constexpr (TW16)
        ng::wgrad_tile16(wp, int(blockIdx.x) - conv_blocks, st0, st1);
    else
        ng::wgrad_tile<128, PREC>(wp, int(blockIdx.x) - conv_blocks, st0, st1);
}

// The bf16 operand mode's 256-wide tiles (igemm_tile256.h) as PERSISTENT workgroups, one per CU.  A weight gradient on its own: every
// workgroup walks units (256 x 256 x one split).  The fused launch: the first conv_wgs workgroups walk the data-gradient tiles, the
// others the weight-gradient units -- the host divides the CUs so that both kinds finish together (pair256_split): a data gradient over
// the padded extent of the benchmark layer has 273 tiles, 1.07 rounds of the chip on its own.
// Between two items of a workgroup one barrier: every wave has read its epilogue staging back before the next item's LDS-DMA lands.
// S = slots of the weight-gradient tile's LDS ring: 8 = 128 KB (default); 10 = the CU's whole 160 KB, seven half-tiles in flight instead of
// five (NIRGAN_WGRAD_RING10, A/B: measured 3 039 against 3 014 cycles per K-tile -- the deeper ring buys nothing, profiles/r04_tile256_ring10.txt)
template <int S>
__global__ __launch_bounds__(512, 2) void wgrad_igemm256_kernel(const ng::WgradParams p, const int units) {
    __shared__ __attribute__((aligned(16))) char lds[S * ng::T256_HALF];
    bool again = false;
    for (int u = ng_xcd_remap(blockIdx.x, gridDim.x); u < units; u += gridDim.x) {
        if (again) ng::t256_bar();
        ng::wgrad_tile256<S>(p, u, lds);
        again = true;
    }
}

template <int S>
__global__ __launch_bounds__(512, 2) void conv_wgrad_pair256_kernel(const ng::ConvParams cp, const ng::WgradParams wp, const int conv_wgs,
                                                                     const int conv_tiles, const int wgrad_units) {
    __shared__ __attribute__((aligned(16))) char lds[S * ng::T256_HALF];
    bool again = false;
    if (int(blockIdx.x) < conv_wgs) {
        for (int t = ng_xcd_remap(blockIdx.x, conv_wgs); t < conv_tiles; t += conv_wgs) {
            if (again) ng::t256_bar();
            ng::conv_tile256<false>(cp, t, lds);
            again = true;
        }
    } else {
        const int y = int(gridDim.x) - conv_wgs;
        for (int u = ng_xcd_remap(int(blockIdx.x) - conv_wgs, y); u < wgrad_units; u += y) {
            if (again) ng::t256_bar();
            ng::wgrad_tile256<S>(wp, u, lds);
            again = true;
        }
    }
}

// precision 3 (igemm_x3.h): the weight gradient with both fp32 operands split into three bf16 terms in the kernel; persistent, one workgroup per CU
template <int TN>
__global__ __launch_bounds__(512, 2) void wgrad_x3_kernel(const ng::WgradParams p, const int units) {
    __shared__ __attribute__((aligned(16))) char sP0[3 * 32 * TN * 2];
    __shared__ __attribute__((aligned(16))) char sP1[3 * 32 * TN * 2];
    __shared__ __attribute__((aligned(16))) char sQ0[3 * 32 * 256];
    __shared__ __attribute__((aligned(16))) char sQ1[3 * 32 * 256];
    for (int u = ng_xcd_remap(blockIdx.x, gridDim.x); u < units; u += gridDim.x) {
        ng::wgrad_tile_x3<TN>(p, u, sP0, sP1, sQ0, sQ1);
        __syncthreads();                // every wave has read its staging back before the next unit's images land
    }
}

// grid (K/256, N): 64 lanes x float4 cover 256 consecutive k of one row; the 4 waves of the block take the
// splits sp = wave, wave+4, ... (independent loads in flight) and are combined through LDS in a fixed order
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ slabs, int nsplit, int N, int K,
                                                          const int32_t* __restrict__ map, float* __restrict__ dst,
                                                          int64_t dst_elems, int dst_row_stride, int accumulate, int row0) {
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int k4 = (blockIdx.x * 64 + lane) * 4;
    const int n = blockIdx.y;                   // destination row; slab row row0 + n (nirgan_reduce_rows_part: a band of the slab's rows)
    const size_t total = size_t(N) * K;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
        const float* src = slabs + size_t(row0 + n) * K + k4;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
        int sp = grp;
        for (; sp + 12 < nsplit; sp += 16) {                 // four slabs in flight per wave (fixed association)
            s += *reinterpret_cast<const f32x4*>(src + sp * total);
            s1 += *reinterpret_cast<const f32x4*>(src + (sp + 4) * total);
            s2 += *reinterpret_cast<const f32x4*>(src + (sp + 8) * total);
            s3 += *reinterpret_cast<const f32x4*>(src + (sp + 12) * total);
        }
        for (; sp < nsplit; sp += 4) s += *reinterpret_cast<const f32x4*>(src + sp * total);
        s = (s + s1) + (s2 + s3);
    }
    part[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || k4 >= K) return;
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int mk = map[k4 + j];
        if (mk < 0) continue;
        const int64_t o = int64_t(n) * dst_row_stride + mk;
        if (o < dst_elems) dst[o] = accumulate ? dst[o] + s[j] : s[j];
    }
}

// The slab sums of several layers in ONE launch (a network's weight gradients: a dozen launches of 12-14 us -- launch latency for up to
// 31 MB each -- become one at the memory rate).  jobs in device memory, 10 x int64 each:
// {slabs, dst, map, nsplit, N, K, dst_elems, dst_row_stride | accumulate << 32, first_block, taps}.
//   taps = 0: the general form, job j owns N_j * ceil(K_j / 256) blocks -- a block sums 256 consecutive k of one row n and scatters them
//             through map (same arithmetic and association as reduce_rows_kernel);
//   taps = T: the map is the Conv2d weight's own (t, c) -> c * T + t (K = T * Cin, Cin % 64 == 0, geometry.conv_fwd_pack): job j owns
//             N_j * Cin / 64 blocks -- a block sums the T x 64 values of 64 input channels of one row n (T contiguous 256-byte pieces per
//             slab), transposes them through LDS and stores 64 * T CONTIGUOUS floats of the gradient.  The scatter of the general form
//             writes 4 bytes every 4 * T: the PMC counted 402 MB of write traffic for 42 MB of gradients (18 layers).
// The splits are added in the same order either way: wave w takes splits w, w + 4, ... (four running sums), then wave 0 joins them.
__global__ __launch_bounds__(256) void reduce_rows_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    __shared__ f32x4 part[4][256];
    __shared__ float tile[64 * 16];
    int j = 0;
    for (int i = 1; i < njobs; ++i)
        if (int(blockIdx.x) >= int(jobs[i * 10 + 8])) j = i;
    const long long* J = jobs + j * 10;
    const float* slabs = reinterpret_cast<const float*>(J[0]);
    float* dst = reinterpret_cast<float*>(J[1]);
    const int32_t* map = reinterpret_cast<const int32_t*>(J[2]);
    const int nsplit = int(J[3]), N = int(J[4]), K = int(J[5]);
    const long long dst_elems = J[6];
    const int dst_row_stride = int(J[7] & 0xffffffffll), accumulate = int(J[7] >> 32);
    const int taps = int(J[9]);
    const int local = int(blockIdx.x) - int(J[8]);
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t total = size_t(N) * K;
    auto sum_splits = [&](const float* src) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
        int sp = grp;
        for (; sp + 12 < nsplit; sp += 16) {
            s += *reinterpret_cast<const f32x4*>(src + sp * total);
            s1 += *reinterpret_cast<const f32x4*>(src + (sp + 4) * total);
            s2 += *reinterpret_cast<const f32x4*>(src + (sp + 8) * total);
            s3 += *reinterpret_cast<const f32x4*>(src + (sp + 12) * total);
        }
        for (; sp < nsplit; sp += 4) s += *reinterpret_cast<const f32x4*>(src + sp * total);
        return (s + s1) + (s2 + s3);
    };
    if (taps > 0) {
        const int cin = K / taps, cb = cin >> 6;
        const int n = local / cb, c0 = (local - n * cb) << 6;
        const int nf = taps * 16;                                       // float4s of the block: tap t, channels c0 + 4 q .. + 3
        for (int f = lane; f < nf; f += 64) {
            const int t = f >> 4, q = f & 15;
            part[grp][f] = sum_splits(slabs + size_t(n) * K + t * cin + c0 + q * 4);
        }
        __syncthreads();
        for (int f = threadIdx.x; f < nf; f += 256) {
            const int t = f >> 4, q = f & 15;
            const f32x4 v = (part[0][f] + part[1][f]) + (part[2][f] + part[3][f]);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(q * 4 + e) * taps + t] = v[e];
        }
        __syncthreads();
        float* drow = dst + (long long)n * dst_row_stride + (long long)c0 * taps;
        for (int i = threadIdx.x; i < 64 * taps; i += 256) drow[i] = accumulate ? drow[i] + tile[i] : tile[i];
        return;
    }
    const int kblocks = (K + 255) / 256;
    const int n = local / kblocks, kb = local - n * kblocks;
    const int k4 = (kb * 64 + lane) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) s = sum_splits(slabs + size_t(n) * K + k4);
    part[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || k4 >= K) return;
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mk = map[k4 + q];
        if (mk < 0) continue;
        const long long o = (long long)n * dst_row_stride + mk;
        if (o < dst_elems) dst[o] = accumulate ? dst[o] + s[q] : s[q];
    }
}

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// four packed values to their destination: fp32, or bf16 (round to nearest even: what the bf16 operand mode would do at every
// fragment read -- done once here when the weights are stored as bf16)
__device__ __forceinline__ void pack_store(float* dst, size_t elem, f32x4 v, int bf16) {
    if (bf16) *reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(dst) + elem) = __builtin_convertvector(v, bf16x4_t);
    else *reinterpret_cast<f32x4*>(dst + elem) = v;
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, int64_t src_elems, int src_row_stride,
                                                        const int32_t* __restrict__ map, float* __restrict__ dst, int N, int K, int bf16) {
    const int k4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int n = blockIdx.y;
    if (k4 >= K) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int mk = map[k4 + j];
        if (mk >= 0) {
            const int64_t o = int64_t(n) * src_row_stride + mk;
            if (o < src_elems) v[j] = src[o];
        }
    }
    pack_store(dst, size_t(n) * K + k4, v, bf16);
}

// all weight packs of a plan in one launch: jobs live in device memory (10 x int64 each):
// src, dst, map, src_elems, N, K, src_row_stride | (bf16 destination << 32), first_block, w3, w3_plane
// (w3 != 0: the fp32 pack also leaves its three bf16 terms h, m, l -- nirgan_split3 -- in the planes w3, w3 + w3_plane, w3 + 2 w3_plane)
__global__ __launch_bounds__(256) void pack_rows_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    for (int i = 1; i < njobs; ++i)
        if (int(blockIdx.x) >= int(jobs[i * 10 + 7])) j = i;
    const long long* J = jobs + j * 10;
    const float* src = reinterpret_cast<const float*>(J[0]);
    float* dst = reinterpret_cast<float*>(J[1]);
    const int32_t* map = reinterpret_cast<const int32_t*>(J[2]);
    const long long src_elems = J[3];
    const int K = int(J[5]), stride = int(J[6] & 0xffffffffll), bf16 = int(J[6] >> 32);
    const int kblocks = (K + 1023) / 1024;
    const int local = int(blockIdx.x) - int(J[7]);
    const int n = local / kblocks, kb = local - n * kblocks;
    const int k4 = (kb * 256 + threadIdx.x) * 4;
    if (k4 >= K) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mk = map[k4 + q];
        if (mk >= 0) {
            const long long o = (long long)n * stride + mk;
            if (o < src_elems) v[q] = src[o];
        }
    }
    pack_store(dst, size_t(n) * K + k4, v, bf16);
    if (J[8] != 0) {
        unsigned short* w3 = reinterpret_cast<unsigned short*>(J[8]) + size_t(n) * K + k4;
        const long long plane = J[9];
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 h, m, l;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float x0 = v[2 * i], x1 = v[2 * i + 1];
            h[i] = ng::x3_pk(x0, x1);
            const float r0 = x0 - __builtin_bit_cast(float, h[i] << 16), r1 = x1 - __builtin_bit_cast(float, h[i] & 0xffff0000u);
            m[i] = ng::x3_pk(r0, r1);
            l[i] = ng::x3_pk(r0 - __builtin_bit_cast(float, m[i] << 16), r1 - __builtin_bit_cast(float, m[i] & 0xffff0000u));
        }
        *reinterpret_cast<u32x2*>(w3) = h;
        *reinterpret_cast<u32x2*>(w3 + plane) = m;
        *reinterpret_cast<u32x2*>(w3 + 2 * plane) = l;
    }
}

}  // namespace


// CUs of the current device (the persistent launches start one workgroup per CU)
static int ng_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

// Workgroups of the fused 256-wide launch that walk the data-gradient tiles (the rest walk weight-gradient units): the division that
// minimises the longer of the two walks.  Mirrored by nirgan_hip/geometry.py::pair256_plan, which chooses the number of splits with the
// same cost figures.
static int pair256_split(const int G, const int conv_tiles, const int conv_nk, const int units, const int unit_nk) {
    // in tenths of a convolution K-tile (2 500 cycles): a weight-gradient K-tile costs 1.2 (3 010 cycles: every operand byte comes from beyond
    // L2), a convolution tile 6.5 more for prologue + epilogue, a weight-gradient unit 4 (profiles/r04_tile256_stamps.txt)
    int best = 1;
    long long best_cost = -1;
    for (int x = 1; x < G; ++x) {
        const long long c = (long long)((conv_tiles + x - 1) / x) * (conv_nk * 10 + 65), w = (long long)((units + (G - x) - 1) / (G - x)) * (unit_nk * 12 + 40);
        const long long cost = c > w ? c : w;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = x; }
    }
    return best;
}

extern "C" int nirgan_wgrad_igemm(const nirgan_wgrad_desc* d, void* stream) {
    ng::WgradParams p;
    const int rc = ng::build_wgrad_params(d, p);
    if (rc != NIRGAN_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p.prec == 3) {
        if (ng::wgrad_x3_ok(p)) {
            const int tn = ng::wgrad_x3_tn(p);
            const int units = (p.N / tn) * ((p.K + 127) >> 7) * p.nsplit * p.nplanes, G = ng_cu_count();
            if (tn == 256) hipLaunchKernelGGL(wgrad_x3_kernel<256>, dim3(units < G ? units : G), dim3(512), 0, st, p, units);
            else hipLaunchKernelGGL(wgrad_x3_kernel<128>, dim3(units < G ? units : G), dim3(512), 0, st, p, units);
            return nirgan_check_launch("wgrad_igemm (three-term split tile)");
        }
        p.prec = 0;             // what the split tile does not cover runs as exact fp32
    }
    if (d->algo != NIRGAN_WGRAD_TILE128 && d->algo != NIRGAN_WGRAD_ONE_UNIT && ng::wgrad_tile256_ok(p)) {
        const int units = (p.N >> 8) * (p.K >> 8) * p.nsplit, G = ng_cu_count();
        if (d->algo == NIRGAN_WGRAD_RING10) hipLaunchKernelGGL(wgrad_igemm256_kernel<10>, dim3(units < G ? units : G), dim3(512), 0, st, p, units);
        else hipLaunchKernelGGL(wgrad_igemm256_kernel<8>, dim3(units < G ? units : G), dim3(512), 0, st, p, units);
        return nirgan_check_launch("wgrad_igemm (256 x 256 tile)");
    }
    const dim3 grid(p.ntiles_n * p.ntiles_k * p.nsplit * p.nplanes);
    // matrix-form problems (the transform-domain weight gradient of a Winograd layer launched on its own) with more units than resident
    // workgroups: persistent workgroups with the epilogue folded into the next unit's K loop (igemm_tiles.h::wgrad_persist)
    if (ng::wgrad_persist_ok(p) && ng::wgrad_matrix_form(p) && grid.x > 512 && d->algo != NIRGAN_WGRAD_ONE_UNIT) {
        hipLaunchKernelGGL(wgrad_persist_kernel, dim3(512), dim3(256), 0, st, p);
        return nirgan_check_launch("wgrad_igemm");
    }
    if (p.pq_bf16) {
        hipLaunchKernelGGL(wgrad_igemm16_kernel, grid, dim3(256), 0, st, p);
        return nirgan_check_launch("wgrad_igemm");
    }
#define NG_LAUNCH_WGRAD(TN, PREC) hipLaunchKernelGGL((wgrad_igemm_kernel<TN, PREC>), grid, dim3(256), 0, st, p)
    if (d->N > 64) {
        if (p.prec == 0) NG_LAUNCH_WGRAD(128, 0); else if (p.prec == 1) NG_LAUNCH_WGRAD(128, 1); else NG_LAUNCH_WGRAD(128, 2);
    } else {
        if (p.prec == 0) NG_LAUNCH_WGRAD(64, 0); else if (p.prec == 1) NG_LAUNCH_WGRAD(64, 1); else NG_LAUNCH_WGRAD(64, 2);
    }
#undef NG_LAUNCH_WGRAD
    return nirgan_check_launch("wgrad_igemm");
}

extern "C" int nirgan_conv_wgrad_pair(const nirgan_conv_desc* c, const nirgan_wgrad_desc* w, void* stream) {
    ng::ConvParams cp;
    ng::WgradParams wp;
    int rc = ng::build_conv_params(c, cp);
    if (rc != NIRGAN_OK) return rc;
    rc = ng::build_wgrad_params(w, wp);
    if (rc != NIRGAN_OK) return rc;
    if (c->N <= 64 || w->N <= 64 || c->ksplit > 1 || (wp.pq_bf16 && !cp.in_bf16) || cp.prec == 3 || wp.prec == 3) {      // narrow, split-K, mixed-storage or three-term variants: two ordinary launches
        rc = nirgan_conv_igemm(c, stream);
        return rc != NIRGAN_OK ? rc : nirgan_wgrad_igemm(w, stream);
    }
    NG_REQUIRE(cp.prec == wp.prec, "conv_wgrad_pair: both halves must use the same precision");
    // (the fused 256-wide launch divides the CUs between the two kinds of item: it needs at least two; a one-CU device takes the 128-row pair)
    if (cp.prec == 1 && cp.algo != NIRGAN_CONV_TILE128 && w->algo != NIRGAN_WGRAD_TILE128 && ng_cu_count() >= 2 && ng::conv_tile256_ok(cp, true, ng_cu_count()) && ng::wgrad_tile256_ok(wp)) {
        const int conv_tiles = ((cp.M + 255) >> 8) * (cp.N >> 8), units = (wp.N >> 8) * (wp.K >> 8) * wp.nsplit, G = ng_cu_count();
        const int conv_wgs = pair256_split(G, conv_tiles, cp.ntaps * (cp.run >> 6), units, wp.rows_per_split >> 6);
        if (w->algo == NIRGAN_WGRAD_RING10) hipLaunchKernelGGL(conv_wgrad_pair256_kernel<10>, dim3(G), dim3(512), 0, static_cast<hipStream_t>(stream), cp, wp, conv_wgs, conv_tiles, units);
        else hipLaunchKernelGGL(conv_wgrad_pair256_kernel<8>, dim3(G), dim3(512), 0, static_cast<hipStream_t>(stream), cp, wp, conv_wgs, conv_tiles, units);
        return nirgan_check_launch("conv_wgrad_pair (256 x 256 tiles)");
    }
    const int conv_blocks = cp.mtiles * cp.ntiles;
    const int wgrad_blocks = wp.ntiles_n * wp.ntiles_k * wp.nsplit;
    const dim3 grid(conv_blocks + wgrad_blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cp.prec == 0) hipLaunchKernelGGL(conv_wgrad_pair_kernel<0>, grid, dim3(256), 0, st, cp, wp, conv_blocks);
    else if (cp.prec == 1 && cp.in_bf16 && wp.pq_bf16) hipLaunchKernelGGL((conv_wgrad_pair_kernel<1, true, true, true>), grid, dim3(256), 0, st, cp, wp, conv_blocks);
    else if (cp.prec == 1 && cp.in_bf16) hipLaunchKernelGGL((conv_wgrad_pair_kernel<1, true, true>), grid, dim3(256), 0, st, cp, wp, conv_blocks);
    else if (cp.prec == 1 && cp.w_bf16) hipLaunchKernelGGL((conv_wgrad_pair_kernel<1, true>), grid, dim3(256), 0, st, cp, wp, conv_blocks);
    else if (cp.prec == 1) hipLaunchKernelGGL(conv_wgrad_pair_kernel<1>, grid, dim3(256), 0, st, cp, wp, conv_blocks);
    else hipLaunchKernelGGL(conv_wgrad_pair_kernel<2>, grid, dim3(256), 0, st, cp, wp, conv_blocks);
    return nirgan_check_launch("conv_wgrad_pair");
}

extern "C" const char* nirgan_wgrad_kernel_name(const nirgan_wgrad_desc* d) {
    ng::WgradParams p;
    if (ng::build_wgrad_params(d, p) != NIRGAN_OK) return nullptr;
    if (p.prec == 3) {
        if (ng::wgrad_x3_ok(p)) return ng::wgrad_x3_tn(p) == 256 ? "wgrad_x3_kernel<256>" : "wgrad_x3_kernel<128>";
        p.prec = 0;
    }
    if (d->algo != NIRGAN_WGRAD_TILE128 && d->algo != NIRGAN_WGRAD_ONE_UNIT && ng::wgrad_tile256_ok(p)) return "wgrad_igemm256_kernel";      // (the launcher's own predicate)
    if (ng::wgrad_persist_ok(p) && ng::wgrad_matrix_form(p) && p.ntiles_n * p.ntiles_k * p.nsplit * p.nplanes > 512 && d->algo != NIRGAN_WGRAD_ONE_UNIT) return "wgrad_persist_kernel";
    if (p.pq_bf16) return "wgrad_igemm16_kernel";
    return d->N > 64 ? "wgrad_igemm_kernel<128>" : "wgrad_igemm_kernel<64>";
}

extern "C" const char* nirgan_conv_wgrad_pair_kernel_name(const nirgan_conv_desc* c, const nirgan_wgrad_desc* w) {
    ng::ConvParams cp;
    ng::WgradParams wp;
    if (ng::build_conv_params(c, cp) != NIRGAN_OK || ng::build_wgrad_params(w, wp) != NIRGAN_OK) return nullptr;
    if (c->N <= 64 || w->N <= 64 || c->ksplit > 1 || (wp.pq_bf16 && !cp.in_bf16) || cp.prec == 3 || wp.prec == 3) return "(two launches)";
    if (cp.prec == 1 && cp.algo != NIRGAN_CONV_TILE128 && w->algo != NIRGAN_WGRAD_TILE128 && ng_cu_count() >= 2 && ng::conv_tile256_ok(cp, true, ng_cu_count()) && ng::wgrad_tile256_ok(wp)) return "conv_wgrad_pair256_kernel";
    return "conv_wgrad_pair_kernel";
}

extern "C" int nirgan_reduce_rows(const float* slabs, int nsplit, int N, int K, const int32_t* map,
                                  float* dst, int64_t dst_elems, int dst_row_stride, int accumulate, void* stream) {
    NG_REQUIRE(slabs && map && dst && nsplit >= 1 && N > 0 && K > 0 && K % 4 == 0 && N <= 65535, "reduce_rows: bad arguments");
    NG_REQUIRE(ng_aligned16(slabs), "reduce_rows: slabs must be 16-byte aligned");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((K + 255) / 256, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slabs, nsplit, N, K, map, dst, dst_elems, dst_row_stride, accumulate, 0);
    return nirgan_check_launch("reduce_rows");
}

extern "C" int nirgan_reduce_rows_part(const float* slabs, int nsplit, int N, int row0, int rows, int K, const int32_t* map,
                                       float* dst, int64_t dst_elems, int dst_row_stride, int accumulate, void* stream) {
    NG_REQUIRE(slabs && map && dst && nsplit >= 1 && N > 0 && K > 0 && K % 4 == 0 && N <= 65535, "reduce_rows_part: bad arguments");
    NG_REQUIRE(row0 >= 0 && rows > 0 && row0 + rows <= N, "reduce_rows_part: rows [%d, %d) outside the slab's %d rows", row0, row0 + rows, N);
    NG_REQUIRE(ng_aligned16(slabs), "reduce_rows_part: slabs must be 16-byte aligned");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((K + 255) / 256, rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slabs, nsplit, N, K, map, dst, dst_elems, dst_row_stride, accumulate, row0);
    return nirgan_check_launch("reduce_rows_part");
}

extern "C" int nirgan_reduce_rows_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream) {
    NG_REQUIRE(jobs_device && njobs >= 1 && njobs <= 64 && total_blocks >= 1, "reduce_rows_batch: bad arguments");
    hipLaunchKernelGGL(reduce_rows_batch_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(jobs_device), njobs);
    return nirgan_check_launch("reduce_rows_batch");
}

static int pack_rows_impl(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                          float* dst, int N, int K, int bf16, void* stream) {
    NG_REQUIRE(src && map && dst && N > 0 && K > 0 && K % 4 == 0 && N <= 65535, "pack_rows: bad arguments");
    NG_REQUIRE(ng_aligned16(dst), "pack_rows: dst must be 16-byte aligned");
    NG_REQUIRE(!bf16 || K % 8 == 0, "pack_rows: bf16 rows need K %% 8 == 0 (16-byte aligned rows)");
    hipLaunchKernelGGL(pack_rows_kernel, dim3((K + 1023) / 1024, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, src_elems, src_row_stride, map, dst, N, K, bf16);
    return nirgan_check_launch("pack_rows");
}

extern "C" int nirgan_pack_rows(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                                float* dst, int N, int K, void* stream) {
    return pack_rows_impl(src, src_elems, src_row_stride, map, dst, N, K, 0, stream);
}

extern "C" int nirgan_pack_rows_bf16(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                                     void* dst_bf16, int N, int K, void* stream) {
    return pack_rows_impl(src, src_elems, src_row_stride, map, static_cast<float*>(dst_bf16), N, K, 1, stream);
}

extern "C" int nirgan_pack_rows_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream) {
    NG_REQUIRE(jobs_device && njobs >= 1 && njobs <= 256 && total_blocks >= 1, "pack_rows_batch: bad arguments");
    hipLaunchKernelGGL(pack_rows_batch_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(jobs_device), njobs);
    return nirgan_check_launch("pack_rows_batch");
}
