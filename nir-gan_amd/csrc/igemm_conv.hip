// Implicit-GEMM valid convolution over halo'd NHWC fp32 buffers on the gfx950 matrix pipe.
//
//   M = B*OH*OW output pixels, N = output channels, K = ntaps*run.
//   Block tile 128(M) x BN(N) x 32(K), 4 waves as 2x2, each wave 64 x BN/2 built from
//   v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/issue/SIMD).  A (pixels x K-slice) and
//   B (weights, [N][K] K-contiguous) tiles are staged by LDS-DMA (global_load_lds_dwordx4):
//   one wave-instruction lands 8 rows x 128 B.  Rows are 128 B, so the 16-byte chunk index
//   is XOR-swizzled with (row>>1)&7 on the *source* side (the LDS image stays lane-linear)
//   and on the ds_read_b128 side, which makes the fragment reads bank-conflict free.
//   One ds_read_b128 gives a lane 4 consecutive k of its row: lanes 0-31 take chunk 2g,
//   lanes 32-63 chunk 2g+1, so MFMA j of the group contracts k = 8g+j and 8g+4+j; A and B
//   use the same order, the sum is a permutation of the k order only.
//   Two LDS stages, one barrier per K-step: the DMA of step s+1 flies under the 64 MFMAs
//   (4096 cycles) of step s.  Out-of-range rows/channels read a zero page.
#include "igemm_tiles.h"
#include "igemm_tile256.h"
#include "igemm_x3.h"
#include <cstdlib>
#include "igemm_x3r.h"

namespace {

template <int BN, int PREC = 0, bool WB16 = false, bool AB16 = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ng::ConvParams p) {
    __shared__ __attribute__((aligned(16))) char st0[(128 + BN) * 128];
    __shared__ __attribute__((aligned(16))) char st1[(128 + BN) * 128];
    ng::conv_tile<BN, PREC, WB16, AB16>(p, blockIdx.x, st0, st1);
}

// the bf16 operand mode's 256 x 256 x 64 eight-phase tile (igemm_tile256.h): one workgroup of eight waves per CU
template <bool F32>
__global__ __launch_bounds__(512, 2) void conv_igemm256_kernel(const ng::ConvParams p) {
    __shared__ __attribute__((aligned(16))) char lds[ng::T256_LDS];
    ng::conv_tile256<F32>(p, ng_xcd_remap(blockIdx.x, gridDim.x), lds);
}

// precision 3 (igemm_x3.h): fp32 operands as three bf16 terms, six bf16 products -- 256 x BN x 32 tiles walked by persistent workgroups,
// one of eight waves per CU; up to four problems (sub-pixel phases) or a batch of planes per launch
template <int BN>
__global__ __launch_bounds__(512, 2) void conv_x3_kernel(const ng::X3Work w) {
    __shared__ __attribute__((aligned(16))) char sA0[ng::X3_A_BYTES];
    __shared__ __attribute__((aligned(16))) char sA1[ng::X3_A_BYTES];
    __shared__ __attribute__((aligned(16))) char sB0[3 * BN * 64];
    __shared__ __attribute__((aligned(16))) char sB1[3 * BN * 64];
    __shared__ __attribute__((aligned(16))) char sRed[8 * (BN / 8) * 2 * 16];
    // (the work list is read through the kernarg segment: a runtime problem index then costs scalar loads, not a scratch copy)
    ng::conv_x3_persist<BN>((const NG_CONST ng::X3Work*)__builtin_amdgcn_kernarg_segment_ptr(), sA0, sA1, sB0, sB1, sRed);
}

// the same arithmetic as ONE wave per SIMD (igemm_x3r.h): four waves, wave tile 64 x BN, the activation operand split in registers and
// never staged, the weight planes through a four-stage LDS-DMA ring, the epilogue's slices under the item's last K-tile
template <int BN, int KIND>
__global__ __launch_bounds__(256, 1) void conv_x3r_kernel(const ng::X3Work w) {
    __shared__ __attribute__((aligned(1024))) char ring[ng::X3R<BN>::RING];
    __shared__ __attribute__((aligned(16))) char stg[4 * ng::X3R<BN>::STG];
    __shared__ __attribute__((aligned(16))) char sRed[4 * (BN / 4) * 2 * 16];
    ng::conv_x3r_persist<BN, KIND>((const NG_CONST ng::X3Work*)__builtin_amdgcn_kernarg_segment_ptr(), ring, stg, sRed);
}

// fp32 -> three bf16 planes h, m, l with x = h + m + l exactly (each term the RNE bf16 of what the previous ones left)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, const long long n, const long long plane) {
    const long long i = (blockIdx.x * 256ll + threadIdx.x) * 8;
    if (i >= n) return;
    ng::bf16x8 H, M, L;
    ng::x3_split8(*reinterpret_cast<const f32x4*>(src + i), *reinterpret_cast<const f32x4*>(src + i + 4), H, M, L);
    *reinterpret_cast<ng::bf16x8*>(dst + i) = H;
    *reinterpret_cast<ng::bf16x8*>(dst + plane + i) = M;
    *reinterpret_cast<ng::bf16x8*>(dst + 2 * plane + i) = L;
}

// split-K second stage: out(m, n) = bias[n] + sum_s ws[s][m][n], written with the descriptor's output geometry
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ng::ConvParams p) {
    const int n4 = p.N / 4;
    const long long total = (long long)p.M * n4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = int(i / n4), q = int(i - (long long)m * n4);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.ksplit; ++k) s += *reinterpret_cast<const f32x4*>(p.split_ws + ((size_t(k) * p.M + m) * p.N + q * 4));
        if (p.bias != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += p.bias[q * 4 + j];
        }
        const int b = m / p.OHW, r = m - b * p.OHW;
        const int oh = r / p.OW, ow = r - oh * p.OW;
        *reinterpret_cast<f32x4*>(p.out + (b * p.out_img + oh * p.out_stride * p.out_row + ow * p.out_stride * p.out_cs + p.out_org + q * 4)) = s;
    }
}

// up to 4 independent problems (the sub-pixel phases of a stride-2 data gradient / transposed convolution) in ONE
// grid: the phases are small (a quarter of the layer each), one launch fills the chip instead of four partial ones
struct ConvGroup {
    ng::ConvParams p[4];
    int first[5];        // first block id of each problem; first[n] = total
    int n;
};

template <int BN, int PREC = 0, bool WB16 = false, bool AB16 = false>
__global__ __launch_bounds__(256, 2) void conv_group_kernel(const ConvGroup g) {
    __shared__ __attribute__((aligned(16))) char st0[(128 + BN) * 128];
    __shared__ __attribute__((aligned(16))) char st1[(128 + BN) * 128];
    const int bid = blockIdx.x;
    int k = 0;
    if (bid >= g.first[1]) k = 1;
    if (bid >= g.first[2]) k = 2;
    if (bid >= g.first[3]) k = 3;
    if (k == 0) ng::conv_tile<BN, PREC, WB16, AB16>(g.p[0], bid, st0, st1);
    else if (k == 1) ng::conv_tile<BN, PREC, WB16, AB16>(g.p[1], bid - g.first[1], st0, st1);
    else if (k == 2) ng::conv_tile<BN, PREC, WB16, AB16>(g.p[2], bid - g.first[2], st0, st1);
    else ng::conv_tile<BN, PREC, WB16, AB16>(g.p[3], bid - g.first[3], st0, st1);
}

}  // namespace

// CUs of the current device (the persistent launches start one workgroup per CU)
int ng::ng_cu_count_conv() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
#ifdef NG_X3R_STAMP
        if (const char* e = getenv("NG_X3R_CUS")) cus = atoi(e);        // (diagnostic build: a grid smaller than the chip)
#endif
    }
    return cus;
}
using ng::ng_cu_count_conv;

// one persistent launch of the three-term split tile over 1..4 problems of one tile width, or over nplanes problems of ps[0]'s geometry
int ng::ng_launch_conv_x3(const ng::ConvParams* ps, const int n, const int bn, const int nplanes, const long long in_plane, const long long w3_pstride,
                          const long long out_plane, hipStream_t st, const char* what) {
    ng::X3Work w;
    int total = 0;
    for (int i = 0; i < n; ++i) { w.p[i] = ps[i]; w.first[i] = total; total += ng::conv_x3_tiles(ps[i], bn) * (nplanes > 1 ? nplanes : 1); }
    for (int i = n; i < 4; ++i) { w.p[i] = ps[0]; w.first[i] = 0x7fffffff; }
    w.first[n] = total;
    for (int i = n + 1; i < 5; ++i) w.first[i] = 0x7fffffff;
    w.n = n;
    w.nplanes = nplanes > 1 ? nplanes : 1;
    w.in_plane = in_plane; w.w3_pstride = w3_pstride; w.out_plane = out_plane;
    w.spread = 0;
    for (int i = 0; i < 4; ++i) w.start[i] = 0;
    const int G = ng_cu_count_conv();
    // problems of unequal K (sub-pixel phases of a 3 x 3 stride-2 layer: 1 / 2 / 2 / 4 taps) on a full grid: the spread walk of
    // igemm_x3.h.  Every position gets T / G tiles of a problem anyway; the T % G left over go to consecutive positions from start[k],
    // chosen longest problem first where the window's load so far is lowest.
    bool unequal = false;
    for (int i = 1; i < n; ++i) unequal = unequal || ps[i].K != ps[0].K;
    if (n >= 2 && w.nplanes == 1 && unequal && total >= G && G <= 1024) {
        long long load[2 * 1024 + 1];
        for (int i = 0; i < G; ++i) load[i] = 0;
        int order[4] = {0, 1, 2, 3};
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j)
                if (ps[order[j]].K > ps[order[i]].K) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
        for (int oi = 0; oi < n; ++oi) {
            const int k = order[oi];
            const int rem = (w.first[k + 1] - w.first[k]) % G;
            if (rem == 0) continue;
            const long long cost = ps[k].K / 32 + 3;                 // K-tiles of a tile + its epilogue, in K-tile units
            long long prefix[2 * 1024 + 1];
            prefix[0] = 0;
            for (int i = 0; i < 2 * G; ++i) prefix[i + 1] = prefix[i] + load[i % G];
            int best = 0;
            for (int s0 = 1; s0 < G; ++s0)
                if (prefix[s0 + rem] - prefix[s0] < prefix[best + rem] - prefix[best]) best = s0;
            for (int t = 0; t < rem; ++t) load[(best + t) % G] += cost;
            w.start[k] = best;
        }
        w.spread = 1;
    }
    const dim3 grid(total < G ? total : G);
    bool reg_fed = true, gen = false, stats = false;
    for (int i = 0; i < n; ++i) {
        reg_fed = reg_fed && ng::conv_x3r_ok(ps[i], bn);
        gen = gen || ng::conv_x3r_generic(ps[i]);
        stats = stats || ng::conv_x3r_stats(ps[i]);
    }
    if (reg_fed && gen) hipLaunchKernelGGL((conv_x3r_kernel<128, 2>), grid, dim3(256), 0, st, w);
    else if (reg_fed && stats) hipLaunchKernelGGL((conv_x3r_kernel<128, 1>), grid, dim3(256), 0, st, w);
    else if (reg_fed) hipLaunchKernelGGL((conv_x3r_kernel<128, 0>), grid, dim3(256), 0, st, w);
    else if (bn == 128) hipLaunchKernelGGL(conv_x3_kernel<128>, grid, dim3(512), 0, st, w);
    else hipLaunchKernelGGL(conv_x3_kernel<64>, grid, dim3(512), 0, st, w);
    return nirgan_check_launch(what);
}
static int launch_conv_x3(const ng::ConvParams* ps, const int n, const int bn, hipStream_t st, const char* what) {
    return ng::ng_launch_conv_x3(ps, n, bn, 1, 0, 0, 0, st, what);
}

#ifdef NG_X3R_STAMP
extern "C" int nirgan_x3r_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ng::ng_x3r_stamps), size_t(n) * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int nirgan_split3(const float* src, void* dst_bf16, int64_t n, int64_t plane, void* stream) {
    NG_REQUIRE(src && dst_bf16 && n > 0 && n % 8 == 0 && plane >= n && plane % 8 == 0, "split3: n and plane must be positive multiples of 8, plane >= n");
    NG_REQUIRE(ng_aligned16(src) && ng_aligned16(dst_bf16), "split3: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, static_cast<unsigned short*>(dst_bf16), (long long)n, (long long)plane);
    return nirgan_check_launch("split3");
}

extern "C" int nirgan_conv_igemm(const nirgan_conv_desc* d, void* stream) {
    ng::ConvParams p;
    const int rc = ng::build_conv_params(d, p);
    if (rc != NIRGAN_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p.prec == 3) {
        if (ng::conv_x3_ok(p)) return launch_conv_x3(&p, 1, ng::conv_x3_bn(p, ng_cu_count_conv()), st, "conv_igemm (three-term split tile)");
        NG_REQUIRE(p.ch == p.N, "conv: out_span = 2 needs a problem the three-term split tile covers (run %% 32 == 0, N %% 64 == 0)");
        p.prec = 0;             // what the split tile does not cover runs as exact fp32
    }
    if (p.algo != NIRGAN_CONV_TILE128 && ng::conv_tile256_ok(p)) {
        if (p.prec == 0) hipLaunchKernelGGL(conv_igemm256_kernel<true>, dim3(((p.M + 255) >> 8) * (p.N >> 8)), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(conv_igemm256_kernel<false>, dim3(((p.M + 255) >> 8) * (p.N >> 8)), dim3(512), 0, st, p);
        return nirgan_check_launch("conv_igemm (256 x 256 tile)");
    }
    const dim3 grid(p.mtiles * p.ntiles * p.ksplit);
#define NG_LAUNCH_CONV(BN, PREC) hipLaunchKernelGGL((conv_igemm_kernel<BN, PREC>), grid, dim3(256), 0, st, p)
    if (d->N > 64) {
        if (p.prec == 0) NG_LAUNCH_CONV(128, 0);
        else if (p.prec == 1 && p.in_bf16) hipLaunchKernelGGL((conv_igemm_kernel<128, 1, true, true>), grid, dim3(256), 0, st, p);
        else if (p.prec == 1 && p.w_bf16) hipLaunchKernelGGL((conv_igemm_kernel<128, 1, true>), grid, dim3(256), 0, st, p);
        else if (p.prec == 1) NG_LAUNCH_CONV(128, 1);
        else NG_LAUNCH_CONV(128, 2);
    } else {
        if (p.prec == 0) NG_LAUNCH_CONV(64, 0);
        else if (p.prec == 1 && p.in_bf16) hipLaunchKernelGGL((conv_igemm_kernel<64, 1, true, true>), grid, dim3(256), 0, st, p);
        else if (p.prec == 1 && p.w_bf16) hipLaunchKernelGGL((conv_igemm_kernel<64, 1, true>), grid, dim3(256), 0, st, p);
        else if (p.prec == 1) NG_LAUNCH_CONV(64, 1);
        else NG_LAUNCH_CONV(64, 2);
    }
#undef NG_LAUNCH_CONV
    if (p.ksplit > 1) {
        const long long total = (long long)p.M * (p.N / 4);
        const int grid = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, p);
    }
    return nirgan_check_launch("conv_igemm");
}

extern "C" const char* nirgan_conv_kernel_name(const nirgan_conv_desc* d) {
    ng::ConvParams p;
    if (ng::build_conv_params(d, p) != NIRGAN_OK) return nullptr;
    if (p.prec == 3) {
        if (ng::conv_x3_ok(p)) {
            const int bn = ng::conv_x3_bn(p, ng_cu_count_conv());
            return ng::conv_x3r_ok(p, bn) ? "conv_x3r_kernel<128>" : (bn == 128 ? "conv_x3_kernel<128>" : "conv_x3_kernel<64>");
        }
        p.prec = 0;
    }
    if (p.algo != NIRGAN_CONV_TILE128 && ng::conv_tile256_ok(p)) return p.prec == 0 ? "conv_igemm256_kernel<fp32>" : "conv_igemm256_kernel";
    return d->N > 64 ? "conv_igemm_kernel<128>" : "conv_igemm_kernel<64>";
}

extern "C" int nirgan_conv_igemm_group(const nirgan_conv_desc* const* descs, int n, void* stream) {
    NG_REQUIRE(descs != nullptr && n >= 1 && n <= 4, "conv_igemm_group: 1..4 descriptors");
    ConvGroup g;
    int total = 0;
    bool wide = false;
    for (int i = 0; i < n; ++i) {
        NG_REQUIRE(descs[i] != nullptr, "conv_igemm_group: null descriptor %d", i);
        const int rc = ng::build_conv_params(descs[i], g.p[i]);
        if (rc != NIRGAN_OK) return rc;
        NG_REQUIRE(descs[i]->ksplit <= 1, "conv_igemm_group: split-K descriptors are not groupable");
        const bool w = descs[i]->N > 64;
        NG_REQUIRE(i == 0 || w == wide, "conv_igemm_group: all problems must use the same tile width (N <= 64 or N > 64)");
        NG_REQUIRE(g.p[i].prec == g.p[0].prec && g.p[i].w_bf16 == g.p[0].w_bf16 && g.p[i].in_bf16 == g.p[0].in_bf16, "conv_igemm_group: all problems must use the same precision and weight storage");
        wide = w;
        g.first[i] = total;
        total += g.p[i].mtiles * g.p[i].ntiles;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g.p[0].prec == 3) {
        // the three-term split tile for every problem of the group, or exact fp32 for all of them
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && ng::conv_x3_ok(g.p[i]) && (g.p[i].N % 128 == 0) == (g.p[0].N % 128 == 0);
        if (ok) {
            // (the tile width from the whole group's tile count: the phases of one launch share the chip)
            long long t128 = 0;
            for (int i = 0; i < n; ++i) t128 += g.p[i].N % 128 == 0 ? ng::conv_x3_tiles(g.p[i], 128) : 0;
            const int bn = (g.p[0].N % 128 != 0 || g.p[0].algo == NIRGAN_CONV_X3_BN64 || t128 * 4 < 3ll * ng_cu_count_conv()) ? 64 : 128;
            return launch_conv_x3(g.p, n, bn, st, "conv_igemm_group (three-term split tile)");
        }
        for (int i = 0; i < n; ++i) {
            NG_REQUIRE(g.p[i].ch == g.p[i].N, "conv_igemm_group: out_span = 2 needs problems the three-term split tile covers (run %% 32 == 0, N %% 64 == 0, one tile width)");
            g.p[i].prec = 0;
        }
    }
    for (int i = n; i < 4; ++i) { g.p[i] = g.p[0]; g.first[i] = 0x7fffffff; }
    g.first[4] = total;
    g.n = n;
    const int prec = g.p[0].prec;
#define NG_LAUNCH_GROUP(BN, PREC) hipLaunchKernelGGL((conv_group_kernel<BN, PREC>), dim3(total), dim3(256), 0, st, g)
    const bool wb = g.p[0].w_bf16 != 0, ab = g.p[0].in_bf16 != 0;
    if (wide) {
        if (prec == 0) NG_LAUNCH_GROUP(128, 0);
        else if (prec == 1 && ab) hipLaunchKernelGGL((conv_group_kernel<128, 1, true, true>), dim3(total), dim3(256), 0, st, g);
        else if (prec == 1 && wb) hipLaunchKernelGGL((conv_group_kernel<128, 1, true>), dim3(total), dim3(256), 0, st, g);
        else if (prec == 1) NG_LAUNCH_GROUP(128, 1);
        else NG_LAUNCH_GROUP(128, 2);
    } else {
        if (prec == 0) NG_LAUNCH_GROUP(64, 0);
        else if (prec == 1 && ab) hipLaunchKernelGGL((conv_group_kernel<64, 1, true, true>), dim3(total), dim3(256), 0, st, g);
        else if (prec == 1 && wb) hipLaunchKernelGGL((conv_group_kernel<64, 1, true>), dim3(total), dim3(256), 0, st, g);
        else if (prec == 1) NG_LAUNCH_GROUP(64, 1);
        else NG_LAUNCH_GROUP(64, 2);
    }
#undef NG_LAUNCH_GROUP
    return nirgan_check_launch("conv_igemm_group");
}
