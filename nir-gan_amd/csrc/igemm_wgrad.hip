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
#include "common.h"

namespace {

struct WgradParams {
    const float* p;
    const float* q;
    float* slabs;
    const float* zero;
    int p_row, p_img, p_cs, p_org;
    int q_row, q_img, q_cs, q_stride, q_org;
    int run, ntaps;
    int tap_off[NIRGAN_MAX_TAPS];
    int K, OW, OHW, M, N;
    int rows_per_split, nsplit, ntiles_n, ntiles_k;
};

template <int TN>
__global__ __launch_bounds__(256, 2) void wgrad_igemm_kernel(const WgradParams p) {
    constexpr int P_BYTES = 32 * TN * 4;
    constexpr int Q_BYTES = 32 * 128 * 4;
    constexpr int STAGE = P_BYTES + Q_BYTES;
    constexpr int LPR = TN / 4;          // lanes per P row
    constexpr int RPI = 64 / LPR;        // P rows per wave-instruction
    constexpr int PI = TN / 32;          // P instructions per wave
    constexpr int EA = TN / 64;          // n-interleave: floats per lane per P fragment read
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int id = ng_xcd_remap(blockIdx.x, tiles * p.nsplit);
    const int split = id / tiles, tile = id - split * tiles;
    const int n0 = (tile % p.ntiles_n) * TN, j0 = (tile / p.ntiles_n) * 128;
    const int mstart = split * p.rows_per_split;
    int mend = mstart + p.rows_per_split;
    mend = mend < p.M ? mend : p.M;
    const int nk = mend > mstart ? (mend - mstart + 31) >> 5 : 0;

    // ---------------- loader state
    const int p_lrow = lane / LPR, p_chunk = lane % LPR;
    const int p_n = n0 + p_chunk * 4;
    const bool p_ok = p_n < p.N;
    const int q_lrow = lane >> 5, q_chunk = lane & 31;
    const int q_j = j0 + q_chunk * 4;
    const bool q_ok = q_j < p.K;
    int q_add = 0;
    if (q_ok) {
        const int t = q_j / p.run;
        q_add = p.tap_off[t] + (q_j - t * p.run);
    }

    auto issue = [&](int stage, int mb) {
        char* sP = smem + stage * STAGE;
        char* sQ = sP + P_BYTES;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int ins = wave * PI + i;
            const int m = mb + ins * RPI + p_lrow;
            const float* src = p.zero;
            if (p_ok && m < mend) {
                const int b = m / p.OHW, r = m - b * p.OHW;
                const int oh = r / p.OW, ow = r - oh * p.OW;
                src = p.p + (b * p.p_img + oh * p.p_row + ow * p.p_cs + p.p_org + p_n);
            }
            ng_glds16(src, sP + ins * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ins = wave * 4 + i;
            int m = mb + ins * 2 + q_lrow;
            m = m < p.M ? m : p.M - 1;
            const float* src = p.zero;
            if (q_ok) {
                const int b = m / p.OHW, r = m - b * p.OHW;
                const int oh = r / p.OW, ow = r - oh * p.OW;
                src = p.q + (b * p.q_img + oh * p.q_stride * p.q_row + ow * p.q_stride * p.q_cs + p.q_org + q_add);
            }
            ng_glds16(src, sQ + ins * 1024);
        }
    };

    // ---------------- compute state
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    const int a_off = half * (TN * 4) + (wr * (TN / 2) + EA * (lane & 31)) * 4;
    const int b_off = half * 512 + (wc * 64 + 2 * (lane & 31)) * 4;
    f32x16 acc[EA][2];
#pragma unroll
    for (int e = 0; e < EA; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;

    auto compute = [&](int stage) {
        const char* sP = smem + stage * STAGE;
        const char* sQ = sP + P_BYTES;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[EA];
            if constexpr (EA == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(sP + a_off + kk * (2 * TN * 4));
                a[0] = v[0]; a[1] = v[1];
            } else {
                a[0] = *reinterpret_cast<const float*>(sP + a_off + kk * (2 * TN * 4));
            }
            const f32x2 bv = *reinterpret_cast<const f32x2*>(sQ + b_off + kk * 1024);
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bv[0], acc[e][0], 0, 0, 0);
                acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bv[1], acc[e][1], 0, 0, 0);
            }
        }
    };

    if (nk > 0) issue(0, mstart);
    for (int s = 0; s < nk; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) issue((s + 1) & 1, mstart + (s + 1) * 32);
        compute(s & 1);
    }

    // ---------------- store the partial tile: row i = (r&3)+8*(r>>2)+4*half, col = lane&31
    const int jj = j0 + wc * 64 + 2 * (lane & 31);
    if (jj < p.K) {
        float* slab = p.slabs + size_t(split) * p.N * p.K;
#pragma unroll
        for (int e = 0; e < EA; ++e) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int n = n0 + wr * (TN / 2) + EA * i + e;
                if (n < p.N) {
                    f32x2 v;
                    v[0] = acc[e][0][r];
                    v[1] = acc[e][1][r];
                    *reinterpret_cast<f32x2*>(slab + size_t(n) * p.K + jj) = v;
                }
            }
        }
    }
}

__global__ void reduce_rows_kernel(const float* __restrict__ slabs, int nsplit, int N, int K,
                                   const int32_t* __restrict__ map, float* __restrict__ dst,
                                   int64_t dst_elems, int dst_row_stride, int accumulate) {
    const int64_t total = int64_t(N) * K;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int n = int(i / K), k = int(i - int64_t(n) * K);
        const int mk = map[k];
        if (mk < 0) continue;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += slabs[int64_t(sp) * total + i];
        const int64_t o = int64_t(n) * dst_row_stride + mk;
        if (o < dst_elems) dst[o] = accumulate ? dst[o] + s : s;
    }
}

__global__ void pack_rows_kernel(const float* __restrict__ src, int64_t src_elems, int src_row_stride,
                                 const int32_t* __restrict__ map, float* __restrict__ dst, int N, int K) {
    const int64_t total = int64_t(N) * K;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int n = int(i / K), k = int(i - int64_t(n) * K);
        const int mk = map[k];
        float v = 0.f;
        if (mk >= 0) {
            const int64_t o = int64_t(n) * src_row_stride + mk;
            if (o < src_elems) v = src[o];
        }
        dst[i] = v;
    }
}

}  // namespace

extern "C" int nirgan_wgrad_igemm(const nirgan_wgrad_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr, "wgrad_igemm: null descriptor");
    NG_REQUIRE(d->p && d->q && d->slabs && d->zero_page, "wgrad_igemm: null pointer");
    NG_REQUIRE(ng_aligned16(d->p) && ng_aligned16(d->q) && ng_aligned16(d->slabs) && ng_aligned16(d->zero_page), "wgrad_igemm: pointers must be 16-byte aligned");
    NG_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->N > 0, "wgrad_igemm: empty problem");
    NG_REQUIRE(d->p_cs % 4 == 0 && d->q_cs % 4 == 0 && d->run % 4 == 0 && d->run > 0, "wgrad_igemm: p_cs, q_cs, run must be multiples of 4");
    NG_REQUIRE(d->ntaps >= 1 && d->ntaps <= NIRGAN_MAX_TAPS, "wgrad_igemm: ntaps=%d out of range", d->ntaps);
    NG_REQUIRE(d->q_stride >= 1, "wgrad_igemm: q_stride must be >= 1");
    NG_REQUIRE(d->p_elems < (int64_t(1) << 31) && d->q_elems < (int64_t(1) << 31), "wgrad_igemm: buffers must be < 2^31 floats");
    NG_REQUIRE(d->p_elems >= int64_t(d->B) * d->p_hp * d->p_wp * d->p_cs, "wgrad_igemm: p_elems too small");
    NG_REQUIRE(d->q_elems >= int64_t(d->B) * d->q_hp * d->q_wp * d->q_cs, "wgrad_igemm: q_elems too small");
    NG_REQUIRE(((d->N + 3) & ~3) <= d->p_cs, "wgrad_igemm: N (rounded up to 4) exceeds p_cs");
    NG_REQUIRE(d->p_oh >= 0 && d->OH - 1 + d->p_oh < d->p_hp && d->p_ow >= 0 && d->OW - 1 + d->p_ow < d->p_wp, "wgrad_igemm: p window out of range");
    int dh0 = d->tap_dh[0], dh1 = d->tap_dh[0], dw0 = d->tap_dw[0], dw1 = d->tap_dw[0];
    for (int t = 1; t < d->ntaps; ++t) {
        dh0 = d->tap_dh[t] < dh0 ? d->tap_dh[t] : dh0; dh1 = d->tap_dh[t] > dh1 ? d->tap_dh[t] : dh1;
        dw0 = d->tap_dw[t] < dw0 ? d->tap_dw[t] : dw0; dw1 = d->tap_dw[t] > dw1 ? d->tap_dw[t] : dw1;
    }
    NG_REQUIRE(d->q_oh + dh0 >= 0 && (d->OH - 1) * d->q_stride + d->q_oh + dh1 < d->q_hp, "wgrad_igemm: q rows out of range");
    NG_REQUIRE(d->q_ow + dw0 >= 0 && int64_t((d->OW - 1) * d->q_stride + d->q_ow + dw1) * d->q_cs + d->run <= int64_t(d->q_wp) * d->q_cs, "wgrad_igemm: q columns out of range");
    const int64_t M = int64_t(d->B) * d->OH * d->OW;
    NG_REQUIRE(M < (int64_t(1) << 31), "wgrad_igemm: too many pixels");
    NG_REQUIRE(d->nsplit >= 1 && d->rows_per_split > 0 && d->rows_per_split % 32 == 0 && int64_t(d->nsplit) * d->rows_per_split >= M, "wgrad_igemm: bad split (nsplit=%d rows=%d M=%lld)", d->nsplit, d->rows_per_split, (long long)M);
    const int K = d->ntaps * d->run;
    NG_REQUIRE(d->slab_elems >= int64_t(d->nsplit) * d->N * K, "wgrad_igemm: slab_elems too small");

    WgradParams p;
    p.p = d->p; p.q = d->q; p.slabs = d->slabs; p.zero = d->zero_page;
    p.p_cs = d->p_cs; p.p_row = d->p_wp * d->p_cs; p.p_img = d->p_hp * p.p_row; p.p_org = d->p_oh * p.p_row + d->p_ow * d->p_cs;
    p.q_cs = d->q_cs; p.q_row = d->q_wp * d->q_cs; p.q_img = d->q_hp * p.q_row; p.q_stride = d->q_stride;
    p.q_org = d->q_oh * p.q_row + d->q_ow * d->q_cs;
    p.run = d->run; p.ntaps = d->ntaps;
    for (int t = 0; t < NIRGAN_MAX_TAPS; ++t) p.tap_off[t] = t < d->ntaps ? d->tap_dh[t] * p.q_row + d->tap_dw[t] * d->q_cs : 0;
    p.K = K; p.OW = d->OW; p.OHW = d->OH * d->OW; p.M = int(M); p.N = d->N;
    p.rows_per_split = d->rows_per_split; p.nsplit = d->nsplit;
    p.ntiles_k = (K + 127) / 128;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->N > 64) {
        p.ntiles_n = (d->N + 127) / 128;
        static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_igemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess; }();
        (void)once;
        hipLaunchKernelGGL(wgrad_igemm_kernel<128>, dim3(p.ntiles_n * p.ntiles_k * p.nsplit), dim3(256), 65536, st, p);
    } else {
        p.ntiles_n = 1;
        hipLaunchKernelGGL(wgrad_igemm_kernel<64>, dim3(p.ntiles_k * p.nsplit), dim3(256), 49152, st, p);
    }
    return nirgan_check_launch("wgrad_igemm");
}

extern "C" int nirgan_reduce_rows(const float* slabs, int nsplit, int N, int K, const int32_t* map,
                                  float* dst, int64_t dst_elems, int dst_row_stride, int accumulate, void* stream) {
    NG_REQUIRE(slabs && map && dst && nsplit >= 1 && N > 0 && K > 0, "reduce_rows: bad arguments");
    const int64_t total = int64_t(N) * K;
    const int grid = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slabs, nsplit, N, K, map, dst, dst_elems, dst_row_stride, accumulate);
    return nirgan_check_launch("reduce_rows");
}

extern "C" int nirgan_pack_rows(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                                float* dst, int N, int K, void* stream) {
    NG_REQUIRE(src && map && dst && N > 0 && K > 0, "pack_rows: bad arguments");
    const int64_t total = int64_t(N) * K;
    const int grid = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, src_elems, src_row_stride, map, dst, N, K);
    return nirgan_check_launch("pack_rows");
}
