// Tile bodies of the two implicit-GEMM kernels (device functions shared by the stand-alone
// kernels and the horizontally fused data-gradient + weight-gradient launch), plus the
// descriptor -> kernel-parameter translation with validation.
#pragma once
#include <type_traits>
#include "common.h"

// In-kernel stamps for the diagnostic build only (scripts/diag/conv_stamp.hip defines NG_DIAG); no stamp
// executes in the product build.
#ifdef NG_DIAG
__device__ __forceinline__ unsigned long long ng_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define NG_DIAG_ENTRY const unsigned long long ng_entry = ng_stamp();
#define NG_DIAG_DECL unsigned long long ng_t0 = ng_stamp(), ng_a = ng_t0, ng_wait = 0, ng_body = 0, ng_b = 0;
#define NG_WAIT_BEGIN ng_a = ng_stamp();
#define NG_WAIT_END ng_b = ng_stamp(); ng_wait += ng_b - ng_a; ng_a = ng_b;
#define NG_BODY_END ng_b = ng_stamp(); ng_body += ng_b - ng_a; ng_a = ng_b;
#define NG_DIAG_STORE(dbg, blk)                                                                          \
    if ((threadIdx.x & 63) == 0 && (dbg) != nullptr) {                                                     \
        unsigned long long* o = (dbg) + (size_t(blk) * 4 + (threadIdx.x >> 6)) * 6;                         \
        const unsigned long long te = ng_stamp();                                                           \
        o[0] = ng_t0; o[1] = ng_loop_end; o[2] = te; o[3] = ng_wait; o[4] = ng_body;                        \
        o[5] = ng_t0 - ng_entry;                              /* set-up: kernel entry to the first LDS-DMA issue */ \
    }
#else
#define NG_DIAG_ENTRY
#define NG_DIAG_DECL
#define NG_WAIT_BEGIN
#define NG_WAIT_END
#define NG_BODY_END
#define NG_DIAG_STORE(dbg, blk)
#endif

namespace ng {

// Operand precision of the contraction (descriptor field `precision`):
//   0  fp32 operands, v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain)
//   1  operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32) as they leave LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   2  fp32 operands split x = hi + mid (two bf16 terms) and contracted as hi*hi + hi*mid + mid*hi on the bf16 pipe
//      (relative error of a product <= ~2^-16), fp32 accumulate
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 ng_bf16_round(const f32x8 v) { return __builtin_convertvector(v, bf16x8); }
__device__ __forceinline__ f32x8 ng_bf16_widen(const bf16x8 v) { return __builtin_convertvector(v, f32x8); }

template <int PREC>
__device__ __forceinline__ void ng_mfma_bf16(const f32x8 a, const f32x8 b, f32x16& acc) {
    const bf16x8 ah = ng_bf16_round(a), bh = ng_bf16_round(b);
    if constexpr (PREC == 2) {
        const bf16x8 am = ng_bf16_round(a - ng_bf16_widen(ah)), bm = ng_bf16_round(b - ng_bf16_widen(bh));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

struct ConvParams {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    const float* zero;
    int in_row, in_img, in_cs, run, in_stride, in_org;
    int ntaps;
    int tap_off[NIRGAN_MAX_TAPS];
    int K;
    int out_cs, out_row, out_img, out_stride, out_org;
    int OW, OHW, M, N;
    int mtiles, ntiles;
    int ksplit;                // > 1: the K-steps are divided over ksplit blocks per tile, partial tiles go to split_ws
    float* split_ws;           // [ksplit][M][N] dense
    unsigned long long* dbg;   // diagnostic build only
    int w_bf16;                // packed weights stored as bf16 (bf16 operand mode only)
    int in_bf16;               // activations read from a bf16 twin (with w_bf16)
    int prec;                  // 0 fp32, 1 bf16 operands, 2 bf16x3 split, 3 fp32-equivalent three-term split (host-side dispatch only)
    float* stats;              // partial sums for the instance norm that follows: [B][stats_cps][2][N], see nirgan_conv_desc
    int stats_chunk0, stats_cps;
    // first pass of the instance-norm backward of the layer whose output gradient this launch writes (nirgan_conv_desc.fuse_*)
    const float* f_y; const float* f_mean; const float* f_rstd; float* f_part;
    int f_img, f_row, f_org, f_act, f_chunk0, f_cps;
    float f_slope;
    int out16, f_y16;          // the output / the fused pass's y are stored as bf16 (nirgan_conv_desc.out_bf16 / fuse_y_bf16)
    int algo;                  // nirgan_conv_desc.algo
    int off32;                 // both operand buffers span < 4 GB: per-lane 32-bit byte offsets from a scalar base (the loader's fast path)
    const unsigned short* w3;  // precision 3: the packed weights as three bf16 planes h, m, l (nirgan_split3), w3_plane elements apart
    long long w3_plane;
    int ch;                    // channels per output pixel: N, or N / 2 with nirgan_conv_desc.out_span = 2 (conv_x3_persist only)
};


// WB16 (bf16 mode only): the packed weights are STORED as bf16 (rounded once by the pack kernel instead of at every
// fragment read -- same values): B rows are 64 B per K-step, one ds_read_b128 is one MFMA operand, 25 % fewer bytes and
// LDS-DMA pieces per K-step.  64-byte rows: a wave-instruction lands 16 rows, the chunk swizzle is chunk ^ ((row>>2)&3).
// AB16 (with WB16): the activations come from a producer's bf16 twin as well -- both operands are read as stored, no
// conversion in the K loop; the 128-byte staging rows then hold 64 k, so a K-step contracts 64 k (half the barriers per product).
template <int BN, int PREC, bool WB16 = false, bool AB16 = false>
__device__ __forceinline__ void conv_tile(const ConvParams& p, const int block_id, char* st0, char* st1,
                                          const float* in_base = nullptr, const float* w_base = nullptr, float* out_base = nullptr) {
    NG_DIAG_ENTRY
    // optional base overrides: a plane-batched launch (csrc/wino6.hip) runs many problems of one geometry from one parameter block
    const float* const p_in = in_base ? in_base : p.in;
    const float* const p_w = w_base ? w_base : p.w;
    float* const p_out = out_base ? out_base : p.out;
    static_assert(!WB16 || PREC == 1, "bf16-stored weights are consumed by the bf16 operand mode only");
    static_assert(!AB16 || WB16, "bf16 activations come with bf16-stored weights");
    constexpr int BM = 128;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NT = BN / 64;   // 32-column MFMA tiles per wave
    constexpr int BI = BN / 32;   // B loader instructions per wave (8 rows each)
    // AB16: both operands are bf16 in memory -> the 128-byte staging rows of the fp32 geometry hold 64 k each: a K-step of 64
    // (half the barriers and DMA waits per product of the 32-k steps the fp32-fed paths take)
    constexpr bool WIDE = AB16;
    constexpr int KS = WIDE ? 64 : 32;            // k per K-step

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.mtiles * p.ntiles;
    const int rid = ng_xcd_remap(block_id, tiles * p.ksplit);
    const int ksp = rid / tiles, id = rid - ksp * tiles;
    const int n0 = (id % p.ntiles) * BN, m0 = (id / p.ntiles) * BM;

    // ---------------- loader state: each lane owns one 16-byte chunk of 4 A rows and BI B rows (fp32 rows of 128 B, 8 per
    // wave-instruction); bf16 rows are 64 B, 16 per wave-instruction
    const int lrow = lane >> 3, lchunk = lane & 7;
    constexpr int AIW = 4;                       // A loader instructions per wave
    int a_base[AIW], a_col[AIW];
#pragma unroll
    for (int i = 0; i < AIW; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        const int lc = lchunk ^ ((row >> 1) & 7);
        const int cw = AB16 ? 8 : 4;             // elements per 16-byte chunk
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const int b = m / p.OHW, r = m - b * p.OHW;
        const int oh = r / p.OW, ow = r - oh * p.OW;
        a_col[i] = lc * cw;
        a_base[i] = b * p.in_img + oh * p.in_stride * p.in_row + ow * p.in_stride * p.in_cs + p.in_org + lc * cw;
    }
    constexpr int BIW = (WB16 && !WIDE) ? BN / 64 : BI;     // B loader instructions per wave (16 rows of 64 B each for bf16 weights under fp32 activations)
    int b_base[BIW], b_col[BIW];
    bool b_ok[BIW];
#pragma unroll
    for (int i = 0; i < BIW; ++i) {
        if constexpr (WIDE) {
            const int row = (wave * BI + i) * 8 + lrow;
            const int lc = lchunk ^ ((row >> 1) & 7);
            const int n = n0 + row;
            b_ok[i] = n < p.N;
            b_col[i] = lc * 8;                                   // in bf16 elements
            b_base[i] = (b_ok[i] ? n : 0) * p.K + lc * 8;
        } else if constexpr (WB16) {
            const int row = (wave * BIW + i) * 16 + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            const int n = n0 + row;
            b_ok[i] = n < p.N;
            b_col[i] = lc * 8;                                   // in bf16 elements
            b_base[i] = (b_ok[i] ? n : 0) * p.K + lc * 8;
        } else {
            const int row = (wave * BI + i) * 8 + lrow;
            const int lc = lchunk ^ ((row >> 1) & 7);
            const int n = n0 + row;
            b_ok[i] = n < p.N;
            b_col[i] = lc * 4;
            b_base[i] = (b_ok[i] ? n : 0) * p.K + lc * 4;
        }
    }

    // fast path of a K-step that lies inside the run, for a tile whose weight rows all exist: the address of a piece is a scalar base
    // (tap + slice of the run) + a per-lane byte offset that is constant for the whole tile -- no vector arithmetic per piece
    constexpr int ESA = AB16 ? 2 : 4, ESB = WB16 ? 2 : 4;
    unsigned a_boff[AIW], b_boff[BIW];
#pragma unroll
    for (int i = 0; i < AIW; ++i) a_boff[i] = unsigned(a_base[i]) * unsigned(ESA);
#pragma unroll
    for (int i = 0; i < BIW; ++i) b_boff[i] = unsigned(b_base[i]) * unsigned(ESB);
    // (weight rows past N read row 0 -- b_base already points there: they feed columns that are never stored and never summed)
    const bool inside = p.off32 != 0;

    auto issue = [&](char* sA, int t, int c0) {
        char* sB = sA + A_BYTES;
        const int toff = p.tap_off[t] + c0;
        if (inside && c0 + KS <= p.run) {
            const char* ab = ng_uniform_ptr(reinterpret_cast<const char*>(p_in) + (long long)toff * ESA);
            const char* bb = ng_uniform_ptr(reinterpret_cast<const char*>(p_w) + (long long)(t * p.run + c0) * ESB);
#pragma unroll
            for (int i = 0; i < AIW; ++i) ng_glds16_so(ab, a_boff[i], sA + (wave * 4 + i) * 1024);
#ifdef NG_DIAG_SKIP_B            // diagnostic build only (wrong results): the weight tile is staged for the first K-step only -- what do its pieces cost?
            if (t != 0 || c0 != 0) return;
#endif
#pragma unroll
            for (int i = 0; i < BIW; ++i) ng_glds16_so(bb, b_boff[i], sB + (wave * (WB16 ? BIW : BI) + i) * 1024);
            return;
        }
#pragma unroll
        for (int i = 0; i < AIW; ++i) {
            const bool ok = c0 + a_col[i] < p.run;
            if constexpr (AB16) {
                const unsigned short* in16 = reinterpret_cast<const unsigned short*>(p_in);
                const float* src = ok ? reinterpret_cast<const float*>(in16 + (a_base[i] + toff)) : p.zero;
                ng_glds16(src, sA + (wave * 4 + i) * 1024);
            } else {
                const float* src = ok ? p_in + (a_base[i] + toff) : p.zero;
                ng_glds16(src, sA + (wave * 4 + i) * 1024);
            }
        }
        const int woff = t * p.run + c0;
#pragma unroll
        for (int i = 0; i < BIW; ++i) {
            const bool ok = b_ok[i] && c0 + b_col[i] < p.run;
            if constexpr (WB16) {                                   // (wide or narrow rows: BIW pieces of 1 KB per wave either way)
                const unsigned short* w16 = reinterpret_cast<const unsigned short*>(p_w);
                const float* src = ok ? reinterpret_cast<const float*>(w16 + (b_base[i] + woff)) : p.zero;
                ng_glds16(src, sB + (wave * BIW + i) * 1024);
            } else {
                const float* src = ok ? p_w + (b_base[i] + woff) : p.zero;
                ng_glds16(src, sB + (wave * BI + i) * 1024);
            }
        }
    };

    // ---------------- compute state
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    int a_off[2], a_key[2], b_off[NT], b_key[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wr * 64 + mt * 32 + (lane & 31);
        a_off[mt] = row * 128;
        a_key[mt] = (row >> 1) & 7;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wc * (BN / 2) + nt * 32 + (lane & 31);
        b_off[nt] = row * ((WB16 && !WIDE) ? 64 : 128);
        b_key[nt] = (WB16 && !WIDE) ? (row >> 2) & 3 : (row >> 1) & 7;
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // fragments of group g+1 are read from LDS before the 16 MFMAs of group g are issued (register double
    // buffer), so the LDS latency sits under 1024 MFMA cycles instead of in front of them
    auto compute = [&](const char* sA) {
        const char* sB = sA + A_BYTES;
        // the MFMA block outranks the other workgroup's loader phase on the shared SIMD (measured: fused Winograd backward launch 0.675 ->
        // 0.664 ms, bf16 step -1 %; the split mode's conversion-heavy block loses 2 % with it)
        if constexpr (PREC != 2) __builtin_amdgcn_s_setprio(2);
        if constexpr (PREC == 0) {
            f32x4 a[2][2], b[2][NT];
            auto load = [&](int g, int slot) {
                const int chunk = 2 * g + half;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    a[slot][mt] = *reinterpret_cast<const f32x4*>(sA + a_off[mt] + ((chunk ^ a_key[mt]) << 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[slot][nt] = *reinterpret_cast<const f32x4*>(sB + b_off[nt] + ((chunk ^ b_key[nt]) << 4));
            };
            load(0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) load(g + 1, (g + 1) & 1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][mt][j], b[g & 1][nt][j], acc[mt][nt], 0, 0, 0);
            }
        } else if constexpr (WIDE) {
            // both operands stored as bf16, 128-byte rows = 64 k: group h contracts k 16h .. 16h+15, lanes 0-31 take the 16-byte
            // chunk 2h of their row, lanes 32-63 chunk 2h+1; fragments of group h+1 are read under the MFMAs of group h
            bf16x8 aw[2][2], bw[2][NT];
            auto load = [&](int h, int slot) {
                const int chunk = 2 * h + half;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    aw[slot][mt] = *reinterpret_cast<const bf16x8*>(sA + a_off[mt] + ((chunk ^ a_key[mt]) << 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bw[slot][nt] = *reinterpret_cast<const bf16x8*>(sB + b_off[nt] + ((chunk ^ b_key[nt]) << 4));
            };
            load(0, 0);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (h + 1 < 4) load(h + 1, (h + 1) & 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[h & 1][mt], bw[h & 1][nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            // bf16 pipe: one MFMA contracts 16 k = the chunks 4h..4h+3 of the 32-float slice; lanes 0-31 hold the 8 k of
            // chunks 4h, 4h+1 of their row, lanes 32-63 those of chunks 4h+2, 4h+3 (same assignment for A and B)
            f32x8 a[2][2], b[2][NT];
            bf16x8 bw[2][NT];
            auto load = [&](int h, int slot) {
                const int c0 = 4 * h + 2 * half;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(sA + a_off[mt] + ((c0 ^ a_key[mt]) << 4));
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(sA + a_off[mt] + (((c0 + 1) ^ a_key[mt]) << 4));
                    a[slot][mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (WB16) {
                        // 64-byte rows of bf16: 16-byte chunk 2h + half holds exactly this lane's 8 k
                        bw[slot][nt] = *reinterpret_cast<const bf16x8*>(sB + b_off[nt] + (((2 * h + half) ^ b_key[nt]) << 4));
                    } else {
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(sB + b_off[nt] + ((c0 ^ b_key[nt]) << 4));
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(sB + b_off[nt] + (((c0 + 1) ^ b_key[nt]) << 4));
                        b[slot][nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
            };
            // PREC 1 prefetches the second half's fragments under the first half's MFMAs; the split variant needs the
            // registers for the hi/mid terms and reads each half just before use
            constexpr bool PF = PREC == 1;
            load(0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int slot = PF ? (h & 1) : 0;
                if (PF && h + 1 < 2) load(h + 1, (h + 1) & 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (WB16) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ng_bf16_round(a[slot][mt]), bw[slot][nt], acc[mt][nt], 0, 0, 0);
                        else ng_mfma_bf16<PREC>(a[slot][mt], b[slot][nt], acc[mt][nt]);
                    }
                if (!PF && h + 1 < 2) load(h + 1, 0);
            }
        }
        if constexpr (PREC != 2) __builtin_amdgcn_s_setprio(0);
    };

    // ---------------- main loop: K-steps enumerate (tap, 32-float slice of the run)
    const int csteps = (p.run + KS - 1) / KS;
    const int nk_all = p.ntaps * csteps;
    const int per = (nk_all + p.ksplit - 1) / p.ksplit;
    const int ks0 = ksp * per;
    const int nk = (ks0 + per < nk_all ? per : nk_all - ks0);      // > 0: the host never over-splits
    int t = ks0 / csteps, c0 = (ks0 - t * csteps) * KS;
    // steady state: the LDS-DMA of step s+1 and its address arithmetic are issued BETWEEN the MFMAs of step s.
    // The two stages are distinct LDS objects and the loop is unrolled by two, so the compiler knows the DMA
    // writes do not alias the fragment reads and can interleave them; the sched_group_barrier sequence asks for
    // fragment reads first, then quads of MFMAs with one load's arithmetic + issue in the shadow of each quad.
    auto hints = [&]() {
        if constexpr (PREC != 0) return;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
#pragma unroll
            for (int q = 0; q < 2 * NT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (g < 2) {
                    __builtin_amdgcn_sched_group_barrier(0x006, 14, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        }
    };
    auto next = [&](int& tt, int& cc) {
        cc += KS;
        if (cc >= p.run) { cc = 0; ++tt; }
    };
    NG_DIAG_DECL
    issue(st0, t, c0);
    int s = 0;
    for (; s + 2 < nk; s += 2) {
        next(t, c0);
        NG_WAIT_BEGIN
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        NG_WAIT_END
        issue(st1, t, c0);
        compute(st0);
        hints();
        NG_BODY_END
        next(t, c0);
        NG_WAIT_BEGIN
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        NG_WAIT_END
        issue(st0, t, c0);
        compute(st1);
        hints();
        NG_BODY_END
    }
    // tail: one or two steps left, stage parity is even at s
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nk) {
        next(t, c0);
        issue(st1, t, c0);
        compute(st0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute(st1);
    } else {
        compute(st0);
    }

#ifdef NG_DIAG
    const unsigned long long ng_loop_end = ng_stamp();
#endif
#ifdef NG_DIAG_SKIP_EPILOGUE   // diagnostic build only (scripts/diag/conv_noepi.hip): upper bound of what hiding the epilogue can buy
    if (p.dbg == nullptr) {
        float keep = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[mt][nt][r];
        if (keep == 123456.789f) p_out[0] = keep;        // keeps the accumulation alive, never true on real data
        return;
    }
#endif
    // ---------------- partial sums for the instance norm that follows: the wave's 64 rows x BN/2 columns leave, per column, FOUR values:
    // a shift k (the column's value in the chunk's first row, without the bias), sum (v - k), sum (v - k)^2 and the count 64.  Taken
    // about a value of the data itself the sums carry no cancellation (about the bias alone, a channel whose mean is far from its bias
    // -- the first layer on all-positive reflectances -- lost digits of its variance: 4e-3 on a weight gradient of the golden net);
    // nirgan_instnorm_fwd re-bases the chunks onto one shift in a fixed order.  The host guarantees OH*OW % 128 == 0: every row of the
    // tile is a real pixel of ONE sample.  The two half-waves hold the same column; one exchange joins them.  No atomics.
    if (p.stats != nullptr) {
        const int b = m0 / p.OHW;
        const int chunk = p.stats_chunk0 + ((m0 - b * p.OHW) >> 7) * 2 + wr;
        float* sp = p.stats + (size_t(b) * p.stats_cps + chunk) * 4 * p.N;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float own = acc[0][nt][0], other = __shfl_xor(own, 32, 64);
            const float k = half == 0 ? own : other;                 // row 0 of the chunk lives in the lower half-wave
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mt][nt][r] - k;
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int col = n0 + wc * (BN / 2) + nt * 32 + (lane & 31);
            if (half == 0 && col < p.N) {
                sp[col] = k;
                sp[p.N + col] = s1;
                sp[2 * p.N + col] = s2;
                sp[3 * p.N + col] = 64.f;
            }
        }
    }
    // ---------------- epilogue: the accumulators go through LDS (the two stage buffers are free now: rows 0-63
    // of the tile in st0, rows 64-127 in st1) so that every output pixel row is written with 16 bytes per lane in
    // whole 128-B lines.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    __syncthreads();                                   // all fragment reads of the last step are done
    {
        float* half_base = reinterpret_cast<float*>(wr == 0 ? st0 : st1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = wc * (BN / 2) + nt * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row_l = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    half_base[row_l * BN + col] = acc[mt][nt][r];
                }
            }
    }
    __syncthreads();
    // W output channels per lane: 4 (16 bytes of fp32), or 8 for a bf16 output (16 bytes again: the store path is bound by the NUMBER of
    // store instructions -- the 8-byte stores of a bf16 output cost as many as the fp32 ones for half the bytes)
    auto write_out = [&](auto wtag) {
        constexpr int W = decltype(wtag)::value;
        constexpr int LPR = BN / W;            // lanes per output row
        constexpr int RPP = 256 / LPR;         // rows per pass
        typedef float fW __attribute__((ext_vector_type(W)));
        typedef __bf16 bW __attribute__((ext_vector_type(W)));
        const int chunk = tid % LPR, row0 = tid / LPR;
        const int n = n0 + chunk * W;
        fW bv = {};
        const bool to_ws = p.ksplit > 1;
        if (p.bias != nullptr && !to_ws) {
#pragma unroll
            for (int j = 0; j < W; ++j) bv[j] = n + j < p.N ? p.bias[n + j] : 0.f;
        }
        int m = m0 + row0;
        const int mc = m < p.M ? m : p.M - 1;
        int b = mc / p.OHW;
        const int r0 = mc - b * p.OHW;
        int oh = r0 / p.OW, ow = r0 - oh * p.OW;
        const int OH = p.OHW / p.OW;
        // fused first pass of the consumer layer's instance-norm backward: this thread's W channels over its rows of the tile (host:
        // OHW % 128 == 0 -- one sample per tile -- and N % 4 == 0)
        const bool fused = p.f_y != nullptr;
        const int fb = m0 / p.OHW;
        fW fm = {}, fr = {}, s1 = {}, s2 = {};
        if (fused && n < p.N) {
            fm = *reinterpret_cast<const fW*>(p.f_mean + size_t(fb) * p.N + n);
            fr = *reinterpret_cast<const fW*>(p.f_rstd + size_t(fb) * p.N + n);
        }
        const float fneg = p.f_act == NIRGAN_ACT_RELU ? 0.f : (p.f_act == NIRGAN_ACT_LRELU ? p.f_slope : 1.f);
#pragma unroll 4
        for (int row = row0; row < BM; row += RPP) {
            if (m < p.M && n < p.N) {
                const float* src = reinterpret_cast<const float*>(row < 64 ? st0 : st1) + (row & 63) * BN + chunk * W;
                fW v = *reinterpret_cast<const fW*>(src);
                v += bv;
                const int oidx = b * p.out_img + oh * p.out_stride * p.out_row + ow * p.out_stride * p.out_cs + p.out_org + n;
                float* dst = to_ws ? p.split_ws + ((size_t(ksp) * p.M + m) * p.N + n) : p_out + oidx;
                if constexpr (W == 8) {               // (host: out16, N % 8 == 0, no split-K) eight bf16, rounded to nearest even
                    *reinterpret_cast<bW*>(reinterpret_cast<unsigned short*>(p_out) + oidx) = __builtin_convertvector(v, bW);
                } else if (p.out16) {                 // (host: N % 4 == 0, no split-K) four bf16
                    *reinterpret_cast<bW*>(reinterpret_cast<unsigned short*>(p_out) + oidx) = __builtin_convertvector(v, bW);
                } else if (n + 4 <= p.N) {
                    *reinterpret_cast<fW*>(dst) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < W; ++j)
                        if (n + j < p.N) dst[j] = v[j];
                }
                if (fused) {
                    const size_t yidx = size_t(b) * p.f_img + size_t(oh * p.out_stride) * p.f_row + size_t(ow * p.out_stride) * p.N + p.f_org + n;
                    fW y4;
                    if (p.f_y16) y4 = __builtin_convertvector(*reinterpret_cast<const bW*>(reinterpret_cast<const unsigned short*>(p.f_y) + yidx), fW);
                    else y4 = *reinterpret_cast<const fW*>(p.f_y + yidx);
                    const fW z = (y4 - fm) * fr;
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const float gz = z[j] > 0.f ? v[j] : v[j] * fneg;
                        s1[j] += gz;
                        s2[j] += gz * z[j];
                    }
                }
            }
            m += RPP;
            ow += RPP;
            while (ow >= p.OW) { ow -= p.OW; ++oh; }
            while (oh >= OH) { oh -= OH; ++b; }
        }
        if (fused) {
            // the RPP threads of a channel group join through LDS (the staged tile has been read), fixed order
            __syncthreads();
            fW* red = reinterpret_cast<fW*>(st0);
            red[tid * 2] = s1;
            red[tid * 2 + 1] = s2;
            __syncthreads();
            if (tid < LPR && n < p.N) {
                fW t1 = {}, t2 = {};
#pragma unroll 4
                for (int r = 0; r < RPP; ++r) {
                    t1 += red[(r * LPR + tid) * 2];
                    t2 += red[(r * LPR + tid) * 2 + 1];
                }
                float* pp = p.f_part + (size_t(fb) * p.f_cps + p.f_chunk0 + ((m0 - fb * p.OHW) >> 7)) * 2 * p.N + n;
                *reinterpret_cast<fW*>(pp) = t1;
                *reinterpret_cast<fW*>(pp + p.N) = t2;
            }
        }
    };
    // eight channels per lane when the output is bf16, the tile's columns are whole (N % BN == 0: no partial channel group), every pixel's
    // channel group is 16-byte aligned, and the fused sums' staging fits the first stage buffer (256 threads x 2 x 32 B = 16 KB)
    if (p.out16 && p.N % BN == 0 && p.ksplit <= 1 && ((p.out_cs | p.out_org | p.out_row | p.out_img) & 7) == 0) write_out(std::integral_constant<int, 8>{});
    else write_out(std::integral_constant<int, 4>{});
    NG_DIAG_STORE(p.dbg, block_id)
}

struct WgradParams {
    const float* p;
    const float* q;
    float* slabs;
    const float* zero;
    int p_row, p_img, p_cs, p_org;
    int q_row, q_img, q_cs, q_stride, q_org;
    int run, ntaps;
    int tap_off[NIRGAN_MAX_TAPS];
    int K, OW, OH, OHW, M, N;
    int rows_per_split, nsplit, ntiles_n, ntiles_k;
    int prec;
    int pq_bf16;               // p and q point to bf16 twins (bf16 operand mode, N > 64)
    int nplanes;               // independent problems of identical geometry in one grid (Winograd-domain weight gradient: 16)
    long long p_plane, q_plane;   // floats between consecutive planes of p / q; slabs are [plane][split][N][K]
    unsigned long long* dbg;   // diagnostic build only
    int fast32_bytes;          // both twins span < 4 GB as bf16 (32-bit byte offsets)
    int fast32;                // buffers < 4 GB, taps at non-negative offsets, OW / M / split length multiples of the K-step (32 pixels; 64 for bf16 twins): the scalar-walk loader applies
};


template <int TN, int PREC>
__device__ __forceinline__ void wgrad_tile(const WgradParams& p, const int block_id, char* st0, char* st1) {
    constexpr int P_BYTES = 32 * TN * 4;
    constexpr int Q_BYTES = 32 * 128 * 4;
    constexpr int STAGE = P_BYTES + Q_BYTES;
    constexpr int LPR = TN / 4;          // lanes per P row
    constexpr int RPI = 64 / LPR;        // P rows per wave-instruction
    constexpr int PI = TN / 32;          // P instructions per wave
    constexpr int EA = TN / 64;          // n-interleave: floats per lane per P fragment read

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int per_plane = tiles * p.nsplit;
    const int rid = ng_xcd_remap(block_id, per_plane * p.nplanes);
    const int plane = rid / per_plane, id = rid - plane * per_plane;
    const int split = id / tiles, tile = id - split * tiles;
    const int n0 = (tile % p.ntiles_n) * TN, j0 = (tile / p.ntiles_n) * 128;
    const int mstart = split * p.rows_per_split;
    int mend = mstart + p.rows_per_split;
    mend = mend < p.M ? mend : p.M;
    const int nk = mend > mstart ? (mend - mstart + 31) >> 5 : 0;
    const float* const Pp = p.p + size_t(plane) * p.p_plane;
    const float* const Qp = p.q + size_t(plane) * p.q_plane;

    // ---------------- loader state
    const int p_lrow = lane / LPR, p_chunk = lane % LPR;
    const bool p_ok = n0 + p_chunk * 4 < p.N;
    const int p_n = p_ok ? n0 + p_chunk * 4 : 0;
    const int q_lrow = lane >> 5, q_chunk = lane & 31;
    const int q_j = j0 + q_chunk * 4;
    const bool q_ok = q_j < p.K;
    int q_add = 0;
    if (q_ok) {
        const int t = q_j / p.run;
        q_add = p.tap_off[t] + (q_j - t * p.run);
    }

    // pixel coordinates of this lane's rows, advanced by 32 pixels per step without integer division
    struct Pix { int b, oh, ow; };
    auto decompose = [&](int m) {
        Pix x;
        x.b = m / p.OHW;
        const int r = m - x.b * p.OHW;
        x.oh = r / p.OW;
        x.ow = r - x.oh * p.OW;
        return x;
    };
    auto advance = [&](Pix& x) {
        x.ow += 32;
        while (x.ow >= p.OW) { x.ow -= p.OW; ++x.oh; }
        while (x.oh >= p.OH) { x.oh -= p.OH; ++x.b; }
    };
    Pix pp[PI], qp[4];
#pragma unroll
    for (int i = 0; i < PI; ++i) pp[i] = decompose(mstart + (wave * PI + i) * RPI + p_lrow);
#pragma unroll
    for (int i = 0; i < 4; ++i) qp[i] = decompose(mstart + (wave * 4 + i) * 2 + q_lrow);

    // scalar-walk fast path (host: fast32; here: the tile's rows and columns all exist): a K-step is 32 consecutive pixels of ONE image
    // row, so its first pixel (sb, soh, sow) walks in scalar registers, the address of a piece is a scalar base + a per-lane byte
    // offset that is constant for the whole tile, and the per-lane pixel walk below (multiplies, compares, wrap loops) is not needed
    // (lanes whose channel n or column j does not exist read a valid address instead of the zero page: what they feed are rows / columns
    // of the tile that are never stored, and no product crosses rows or columns)
    const bool fast = p.fast32 != 0;
    unsigned pl_off[PI], ql_off[4];
#pragma unroll
    for (int i = 0; i < PI; ++i) pl_off[i] = unsigned(((wave * PI + i) * RPI + p_lrow) * p.p_cs + (p_ok ? p_chunk * 4 : 0)) * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i) ql_off[i] = unsigned(((wave * 4 + i) * 2 + q_lrow) * p.q_stride * p.q_cs + q_add) * 4u;
    int sb = __builtin_amdgcn_readfirstlane(mstart / p.OHW);
    int soh = __builtin_amdgcn_readfirstlane((mstart - sb * p.OHW) / p.OW);
    int sow = __builtin_amdgcn_readfirstlane(mstart - sb * p.OHW - soh * p.OW);

    auto issue = [&](char* sP, int mb) {
        char* sQ = sP + P_BYTES;
        if (fast) {
            const char* pb = ng_uniform_ptr(reinterpret_cast<const char*>(Pp + (size_t(sb) * p.p_img + size_t(soh) * p.p_row + sow * p.p_cs + p.p_org + (n0 < p.N ? n0 : 0))));
            const char* qb = ng_uniform_ptr(reinterpret_cast<const char*>(Qp + (size_t(sb) * p.q_img + size_t(soh) * p.q_stride * p.q_row + sow * p.q_stride * p.q_cs + p.q_org)));
#pragma unroll
            for (int i = 0; i < PI; ++i) ng_glds16_so(pb, pl_off[i], sP + (wave * PI + i) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) ng_glds16_so(qb, ql_off[i], sQ + (wave * 4 + i) * 1024);
            sow += 32;
            if (sow >= p.OW) { sow = 0; ++soh; }
            if (soh >= p.OH) { soh = 0; ++sb; }
            return;
        }
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int ins = wave * PI + i;
            const int m = mb + ins * RPI + p_lrow;
            const float* src = Pp + (pp[i].b * p.p_img + pp[i].oh * p.p_row + pp[i].ow * p.p_cs + p.p_org + p_n);
            src = (p_ok && m < mend) ? src : p.zero;
            ng_glds16(src, sP + ins * 1024);
            advance(pp[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ins = wave * 4 + i;
            const int m = mb + ins * 2 + q_lrow;
            const float* src = Qp + (qp[i].b * p.q_img + qp[i].oh * p.q_stride * p.q_row + qp[i].ow * p.q_stride * p.q_cs + p.q_org + q_add);
            src = (q_ok && m < p.M) ? src : (q_ok ? Qp + (p.q_org + q_add) : p.zero);
            ng_glds16(src, sQ + ins * 1024);
            advance(qp[i]);
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

    // 4 pixel-pair steps per group: the 8 fragment reads of group g+1 are issued before the 16 MFMAs of group g
    auto compute = [&](const char* sP) {
        const char* sQ = sP + P_BYTES;
        if constexpr (PREC != 2) __builtin_amdgcn_s_setprio(2);
        if constexpr (PREC == 0) {
            float a[2][4][EA];
            f32x2 b[2][4];
            auto load = [&](int g, int slot) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = 4 * g + i;
                    if constexpr (EA == 2) {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(sP + a_off + kk * (2 * TN * 4));
                        a[slot][i][0] = v[0]; a[slot][i][1] = v[1];
                    } else {
                        a[slot][i][0] = *reinterpret_cast<const float*>(sP + a_off + kk * (2 * TN * 4));
                    }
                    b[slot][i] = *reinterpret_cast<const f32x2*>(sQ + b_off + kk * 1024);
                }
            };
            load(0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) load(g + 1, (g + 1) & 1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < EA; ++e) {
                        acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][e], b[g & 1][i][0], acc[e][0], 0, 0, 0);
                        acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][e], b[g & 1][i][1], acc[e][1], 0, 0, 0);
                    }
            }
        } else {
            // bf16 pipe: one MFMA contracts the 16 pixels 16q..16q+15; lanes 0-31 hold the even ones, lanes 32-63 the odd
            // ones of their n (A) / J (B) column -- the same per-lane addresses as the fp32 path, 8 pixel pairs at a time
            f32x8 a[2][EA], b[2][2];
            auto load = [&](int q, int slot) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = 8 * q + i;
                    if constexpr (EA == 2) {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(sP + a_off + kk * (2 * TN * 4));
                        a[slot][0][i] = v[0]; a[slot][1][i] = v[1];
                    } else {
                        a[slot][0][i] = *reinterpret_cast<const float*>(sP + a_off + kk * (2 * TN * 4));
                    }
                    const f32x2 w = *reinterpret_cast<const f32x2*>(sQ + b_off + kk * 1024);
                    b[slot][0][i] = w[0]; b[slot][1][i] = w[1];
                }
            };
            constexpr bool PF = PREC == 1;
            load(0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int slot = PF ? (q & 1) : 0;
                if (PF && q + 1 < 2) load(q + 1, (q + 1) & 1);
#pragma unroll
                for (int e = 0; e < EA; ++e) {
                    ng_mfma_bf16<PREC>(a[slot][e], b[slot][0], acc[e][0]);
                    ng_mfma_bf16<PREC>(a[slot][e], b[slot][1], acc[e][1]);
                }
                if (!PF && q + 1 < 2) load(q + 1, 0);
            }
        }
        if constexpr (PREC != 2) __builtin_amdgcn_s_setprio(0);
    };

    // same structure as conv_tile: two distinct LDS stage objects, loop unrolled by two, the LDS-DMA of the next
    // 32 pixels issued between the MFMAs of the current ones
    auto hints = [&]() {
        if constexpr (PREC != 0) return;
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int q = 0; q < 2 * EA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (g < 2) {
                    __builtin_amdgcn_sched_group_barrier(0x006, 16, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        }
    };
    if (nk > 0) {
        issue(st0, mstart);
        int s = 0;
        for (; s + 2 < nk; s += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue(st1, mstart + (s + 1) * 32);
            compute(st0);
            hints();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue(st0, mstart + (s + 2) * 32);
            compute(st1);
            hints();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) {
            issue(st1, mstart + (s + 1) * 32);
            compute(st0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(st1);
        } else {
            compute(st0);
        }
    }

    // ---------------- store the partial tile through LDS (rows = n, 128 columns J): whole 512-B slab rows,
    // 16 bytes per lane.  Accumulator layout: row i = (r&3)+8*(r>>2)+4*half -> n = base + EA*i + e, col = lane&31.
    __syncthreads();
    {
        float* half_base = reinterpret_cast<float*>(wr == 0 ? st0 : st1);
#pragma unroll
        for (int e = 0; e < EA; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                f32x2 v;
                v[0] = acc[e][0][r];
                v[1] = acc[e][1][r];
                *reinterpret_cast<f32x2*>(half_base + (EA * i + e) * 128 + wc * 64 + 2 * (lane & 31)) = v;
            }
    }
    __syncthreads();
    {
        float* slab = p.slabs + (size_t(plane) * p.nsplit + split) * p.N * p.K;
        const int chunk = tid & 31, row0 = tid >> 5;
        const int jj = j0 + chunk * 4;
        if (jj < p.K) {
#pragma unroll 4
            for (int row = row0; row < TN; row += 8) {
                const int n = n0 + row;
                if (n < p.N) {
                    const float* src = reinterpret_cast<const float*>(row < TN / 2 ? st0 : st1) + (row % (TN / 2)) * 128 + chunk * 4;
                    *reinterpret_cast<f32x4*>(slab + size_t(n) * p.K + jj) = *reinterpret_cast<const f32x4*>(src);
                }
            }
        }
    }
}

// The fp32 128 x 128 weight-gradient tile as PERSISTENT workgroups (round 2).  Equal tiles keep the resident workgroups of a CU in
// lockstep, so their MFMA-free phases (first DMA wait, epilogue through LDS) coincide instead of filling each other; here a workgroup
// walks its (plane, split, tile) units -- unit, unit + stride, ... -- as ONE stream of 32-pixel K-steps with the LDS images and
// fragment reads of wgrad_tile<128, 0>, issues the next unit's first DMA during this unit's last step, and drains the finished
// unit's accumulators (a second set, 64 AGPRs) two 8-byte stores per K-step straight from registers: a lane's (acc[e][0][r],
// acc[e][1][r]) are two adjacent columns of a slab row, 32 lanes = 256 contiguous bytes, no LDS transpose.
// For problems in MATRIX FORM only: both operands plain row-major matrices per plane (one tap, unit stride, one image row -- the
// transform-domain weight gradient dU[f] = Yt[f]^T V[f] of the Winograd layers), where the pixel walk reduces to `t * row stride`.
// The same skeleton with wgrad_tile's general walk (taps, strides, (b, oh, ow) advanced per step) was built and measured on the
// stride-2 / transposed layers' weight gradients: correct, but 256 VGPRs + 320 B of scratch and the walk inside 16 unrolled steps made
// it 40-50 % SLOWER than one tile per workgroup (582 vs 383 us); removed.
// Host: every split holds rows (nk >= 1), fp32 operands, N > 64 (wgrad_persist_ok && wgrad_matrix_form).
struct WgradUnit { const float* Pp; const float* Qp; float* slab; int n0, j0, mstart, mend, nk; bool p_ok, q_ok; int p_n, q_add; bool inside; };

__device__ __forceinline__ void wgrad_persist(const WgradParams& p, const int first, const int stride, char* lds) {
    constexpr int TN = 128, P_BYTES = 32 * TN * 4, Q_BYTES = 32 * 128 * 4, STAGE = P_BYTES + Q_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.ntiles_n * p.ntiles_k, per_plane = tiles * p.nsplit, total = per_plane * p.nplanes;
    const int lrow = lane >> 5, chunk = lane & 31;

    // (32-bit byte offsets inside a plane)
    const bool off32 = (long long)p.M * p.p_cs < (1ll << 29) && (long long)p.M * p.q_cs < (1ll << 29);
    auto setup = [&](int logical, WgradUnit& u) {
        const int rid = ng_xcd_remap(logical, total);
        const int plane = rid / per_plane, id = rid - plane * per_plane;
        const int split = id / tiles, tile = id - split * tiles;
        u.n0 = (tile % p.ntiles_n) * TN;
        u.j0 = (tile / p.ntiles_n) * 128;
        u.mstart = split * p.rows_per_split;
        const int mend = u.mstart + p.rows_per_split;
        u.mend = mend < p.M ? mend : p.M;
        u.nk = (u.mend - u.mstart + 31) >> 5;
        u.Pp = p.p + size_t(plane) * p.p_plane + p.p_org;
        u.Qp = p.q + size_t(plane) * p.q_plane + p.q_org;
        u.slab = p.slabs + (size_t(plane) * p.nsplit + split) * p.N * p.K;
        u.p_ok = u.n0 + chunk * 4 < p.N;
        u.p_n = u.p_ok ? u.n0 + chunk * 4 : 0;
        const int q_j = u.j0 + chunk * 4;
        u.q_ok = q_j < p.K;
        u.q_add = 0;
        if (u.q_ok) u.q_add = q_j;
        u.inside = u.n0 + TN <= p.N && u.j0 + 128 <= p.K && off32;       // every lane's column exists: no zero-page selects
    };
    // per-lane byte offsets of a piece from the step's first pixel (constants of the launch): piece i of this wave holds pixel rows
    // (4 wave + i) * 2 + lrow
    unsigned pl_off[4], ql_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pl_off[i] = unsigned(((wave * 4 + i) * 2 + lrow) * p.p_cs + chunk * 4) * 4u;
        ql_off[i] = unsigned(((wave * 4 + i) * 2 + lrow) * p.q_cs + chunk * 4) * 4u;
    }
    // wave w owns P pieces 4w .. 4w+3 and Q pieces 4w .. 4w+3 (a piece = 2 pixel rows x 512 B); mb = first pixel of the step
    auto issue = [&](const WgradUnit& u, char* sP, int mb) {
        char* sQ = sP + P_BYTES;
        if (u.inside && mb + 32 <= u.mend) {
            // a whole step inside the unit: SGPR base + constant per-lane offsets, no vector arithmetic per piece
            const char* pb = ng_uniform_ptr(reinterpret_cast<const char*>(u.Pp + (size_t(mb) * p.p_cs + u.n0)));
            const char* qb = ng_uniform_ptr(reinterpret_cast<const char*>(u.Qp + (size_t(mb) * p.q_cs + u.j0)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ng_glds16_so(pb, pl_off[i], sP + (wave * 4 + i) * 1024);
                ng_glds16_so(qb, ql_off[i], sQ + (wave * 4 + i) * 1024);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ins = wave * 4 + i;
            const int t = mb + ins * 2 + lrow;
            ng_glds16((u.p_ok && t < u.mend) ? u.Pp + (t * p.p_cs + u.p_n) : p.zero, sP + ins * 1024);
            // rows past M meet P = 0: any valid address does
            ng_glds16(u.q_ok ? u.Qp + ((t < p.M ? t : 0) * p.q_cs + u.q_add) : p.zero, sQ + ins * 1024);
        }
    };

    const int wr = wave >> 1, wc = wave & 1, half = lane >> 5;
    const int a_off = half * (TN * 4) + (wr * (TN / 2) + 2 * (lane & 31)) * 4;
    const int b_off = half * 512 + (wc * 64 + 2 * (lane & 31)) * 4;
    // 4 pixel-pair steps per group: the 8 fragment reads of group g+1 are issued before the 16 MFMAs of group g
    // `between(g)` runs after the MFMAs of group g have been issued: the drain of the previous unit executes in their shadow
    auto compute = [&](const char* sP, f32x16 (&acc)[2][2], auto&& between) {
        const char* sQ = sP + P_BYTES;
        __builtin_amdgcn_s_setprio(2);
        f32x2 a[2][4], b[2][4];
        auto load = [&](int g, int slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = 4 * g + i;
                a[slot][i] = *reinterpret_cast<const f32x2*>(sP + a_off + kk * (2 * TN * 4));
                b[slot][i] = *reinterpret_cast<const f32x2*>(sQ + b_off + kk * 1024);
            }
        };
        load(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) load(g + 1, (g + 1) & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][e], b[g & 1][i][0], acc[e][0], 0, 0, 0);
                    acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][e], b[g & 1][i][1], acc[e][1], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            between(g);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // piece q = (e, r): accumulator row i = (r & 3) + 8 (r >> 2) + 4 half  ->  slab row n0 + wr * 64 + 2 i + e, columns j0 + wc * 64 + 2 (lane & 31) + {0, 1}
    struct Done { float* slab; int n0, j0; };                     // what the drain of a finished unit needs
    auto drain = [&](const f32x16 (&acc)[2][2], int q, const Done& u) {
        const int e = q >> 4, r = q & 15;
        const int n = u.n0 + wr * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + e;
        const int c = u.j0 + wc * 64 + 2 * (lane & 31);
        if (n < p.N && c < p.K) {                                  // K % 4 == 0 and c even: both columns exist
            f32x2 v;
            v[0] = acc[e][0][r];
            v[1] = acc[e][1][r];
            *reinterpret_cast<f32x2*>(u.slab + size_t(n) * p.K + c) = v;
        }
    };
    auto zero = [&](f32x16 (&acc)[2][2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    };

    f32x16 accA[2][2], accB[2][2];
    int step = 0;
    WgradUnit ucur, unext;
    Done uprev = {nullptr, 0, 0};
    auto one_step = [&](f32x16 (&cur)[2][2], int s, int next_logical, auto&& between) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < ucur.nk) issue(ucur, lds + ((step + 1) & 1) * STAGE, ucur.mstart + (s + 1) * 32);
        else if (next_logical < total) {
            setup(next_logical, unext);
            issue(unext, lds + ((step + 1) & 1) * STAGE, unext.mstart);
        }
        compute(lds + (step & 1) * STAGE, cur, between);
        ++step;
    };
    auto run_unit = [&](f32x16 (&cur)[2][2], const f32x16 (&prev)[2][2], bool have_prev, int next_logical) {
        zero(cur);
#pragma unroll
        for (int s = 0; s < 16; ++s) {                       // compile-time piece indices: register-indexed drain
            if (s < ucur.nk) {
                one_step(cur, s, next_logical, [&](int g) {
                    if (have_prev && g == 0) drain(prev, 2 * s, uprev);
                    if (have_prev && g == 1) drain(prev, 2 * s + 1, uprev);
                });
            } else if (have_prev) {                          // a unit shorter than 16 steps: what is left of the drain goes out here
                drain(prev, 2 * s, uprev);
                drain(prev, 2 * s + 1, uprev);
            }
        }
        for (int s = 16; s < ucur.nk; ++s) one_step(cur, s, next_logical, [](int) {});
        uprev = Done{ucur.slab, ucur.n0, ucur.j0};
        ucur = unext;
    };
    if (first >= total) return;
    setup(first, ucur);
    issue(ucur, lds, ucur.mstart);
    int logical = first;
    bool have_prev = false;
    while (true) {
        run_unit(accA, accB, have_prev, logical + stride);
        logical += stride;
        have_prev = true;
        if (logical >= total) {
#pragma unroll
            for (int q = 0; q < 32; ++q) drain(accA, q, uprev);
            break;
        }
        run_unit(accB, accA, true, logical + stride);
        logical += stride;
        if (logical >= total) {
#pragma unroll
            for (int q = 0; q < 32; ++q) drain(accB, q, uprev);
            break;
        }
    }
}

// whether wgrad_persist covers a problem: the wide fp32 tile, rows in every split
inline bool wgrad_persist_ok(const WgradParams& p) {
    return p.prec == 0 && !p.pq_bf16 && p.N > 64 && p.K % 4 == 0 && (long long)(p.nsplit - 1) * p.rows_per_split < p.M;
}
inline bool wgrad_matrix_form(const WgradParams& p) {
    return p.ntaps == 1 && p.tap_off[0] == 0 && p.q_stride == 1 && p.OH == 1 && p.OW == p.M;
}

// Weight-gradient tile over bf16 TWINS (bf16 operand mode, 128 rows n x 128 columns J): both operands are read as stored.
// LDS images P16[m][n] and Q16[m][J]: 32 pixel rows of 256 B, filled by LDS-DMA (a piece = 4 rows), the 16-byte chunk of a row
// XOR-swizzled with f(row) = ((row&3)<<2) | ((row>>2)&3) on the source side.  The MFMA wants k (= pixel m) contiguous per
// lane, the images have it along rows: ds_read_b64_tr_b16 reads, per 16-lane group, a 4-row x 16-column block and hands lane
// i column i with the 4 rows in its 4 elements -- two of them are one 32x32x16 operand (8 k).  16 reads of 8 B per wave and
// K-step instead of 32 ds_read_b64 of fp32, 16 DMA pieces instead of 32, no conversion.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void wgrad_tile16(const WgradParams& p, const int block_id, char* st0, char* st1) {
    constexpr int TN = 128;
    constexpr int MS = 64;                     // pixel rows per K-step (a K-step of 64: half the barriers of the 32-row steps the fp32 tile takes)
    constexpr int P_BYTES = MS * 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int id = ng_xcd_remap(block_id, tiles * p.nsplit);
    const int split = id / tiles, tile = id - split * tiles;
    const int n0 = (tile % p.ntiles_n) * TN, j0 = (tile / p.ntiles_n) * 128;
    const int mstart = split * p.rows_per_split;
    int mend = mstart + p.rows_per_split;
    mend = mend < p.M ? mend : p.M;
    const int nk = mend > mstart ? (mend - mstart + MS - 1) / MS : 0;
    const unsigned short* P16 = reinterpret_cast<const unsigned short*>(p.p);
    const unsigned short* Q16 = reinterpret_cast<const unsigned short*>(p.q);

    // ---------------- loader state: wave w owns pieces 4w .. 4w+3 of each image (rows 16w .. 16w+15)
    const int lrow = lane >> 4, pch = lane & 15;
    struct Pix { int b, oh, ow; };
    auto decompose = [&](int m) {
        Pix x;
        x.b = m / p.OHW;
        const int r = m - x.b * p.OHW;
        x.oh = r / p.OW;
        x.ow = r - x.oh * p.OW;
        return x;
    };
    auto advance = [&](Pix& x) {
        x.ow += MS;
        while (x.ow >= p.OW) { x.ow -= p.OW; ++x.oh; }
        while (x.oh >= p.OH) { x.oh -= p.OH; ++x.b; }
    };
    constexpr int PI = MS / 16;                // loader instructions per wave and image
    Pix px[PI];
    int p_n[PI], q_add[PI];
    bool p_ok[PI], q_ok[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int row = (wave * PI + i) * 4 + lrow;
        const int lc = pch ^ (((row & 3) << 2) | ((row >> 2) & 3));      // logical chunk (8 elements) this lane fetches
        px[i] = decompose(mstart + row);
        p_ok[i] = n0 + lc * 8 < p.N;
        p_n[i] = p_ok[i] ? n0 + lc * 8 : 0;
        const int j = j0 + lc * 8;
        q_ok[i] = j < p.K;
        q_add[i] = 0;
        if (q_ok[i]) {
            const int t = j / p.run;
            q_add[i] = p.tap_off[t] + (j - t * p.run);
        }
    }
    // scalar-walk fast path, as in wgrad_tile: a K-step is 64 consecutive pixels of one image row
    const bool fast = p.fast32 != 0;
    unsigned pl_off[PI], ql_off[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int row = (wave * PI + i) * 4 + lrow;
        pl_off[i] = unsigned(row * p.p_cs + (p_ok[i] ? p_n[i] - n0 : 0)) * 2u;
        ql_off[i] = unsigned(row * p.q_stride * p.q_cs + q_add[i]) * 2u;
    }
    int sb = __builtin_amdgcn_readfirstlane(mstart / p.OHW);
    int soh = __builtin_amdgcn_readfirstlane((mstart - sb * p.OHW) / p.OW);
    int sow = __builtin_amdgcn_readfirstlane(mstart - sb * p.OHW - soh * p.OW);
    auto issue = [&](char* sP, int mb) {
        char* sQ = sP + P_BYTES;
        if (fast) {
            const char* pb = ng_uniform_ptr(reinterpret_cast<const char*>(P16 + (size_t(sb) * p.p_img + size_t(soh) * p.p_row + sow * p.p_cs + p.p_org + (n0 < p.N ? n0 : 0))));
            const char* qb = ng_uniform_ptr(reinterpret_cast<const char*>(Q16 + (size_t(sb) * p.q_img + size_t(soh) * p.q_stride * p.q_row + sow * p.q_stride * p.q_cs + p.q_org)));
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                ng_glds16_so(pb, pl_off[i], sP + (wave * PI + i) * 1024);
                ng_glds16_so(qb, ql_off[i], sQ + (wave * PI + i) * 1024);
            }
            sow += MS;
            if (sow >= p.OW) { sow = 0; ++soh; }
            if (soh >= p.OH) { soh = 0; ++sb; }
            return;
        }
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int ins = wave * PI + i;
            const int m = mb + ins * 4 + lrow;
            const unsigned short* sp = P16 + (px[i].b * p.p_img + px[i].oh * p.p_row + px[i].ow * p.p_cs + p.p_org + p_n[i]);
            const float* src = (p_ok[i] && m < mend) ? reinterpret_cast<const float*>(sp) : p.zero;
            ng_glds16(src, sP + ins * 1024);
            const unsigned short* sq = Q16 + (px[i].b * p.q_img + px[i].oh * p.q_stride * p.q_row + px[i].ow * p.q_stride * p.q_cs + p.q_org + q_add[i]);
            const unsigned short* sq0 = Q16 + (p.q_org + q_add[i]);
            src = (q_ok[i] && m < p.M) ? reinterpret_cast<const float*>(sq) : (q_ok[i] ? reinterpret_cast<const float*>(sq0) : p.zero);
            ng_glds16(src, sQ + ins * 1024);
            advance(px[i]);
        }
    };

    // ---------------- compute state
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    // byte offset of this lane's 8 bytes inside a 4-row block whose first row is 16h + 8*(g>>1) + 4j and first chunk c0:
    //   256*(row) + 16*((c0 + (pp>>1)) ^ f(row)) + 8*(pp&1),  f(row) = (q<<2) | ((2*(g>>1) + j) & 3)   (h drops out of f)
    int rowoff[2], key[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rowoff[j] = 256 * (8 * (g >> 1) + 4 * j + q) + 8 * (pp & 1);
        key[j] = (q << 2) | ((2 * (g >> 1) + j) & 3);
    }
    int a_ch[2], b_ch[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        a_ch[e] = wr * 8 + e * 4 + 2 * (g & 1) + (pp >> 1);
        b_ch[e] = wc * 8 + e * 4 + 2 * (g & 1) + (pp >> 1);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;

    auto frag = [&](const char* img, int h, int ch) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NG_LDS s16x4*)(img + 4096 * h + rowoff[0] + 16 * (ch ^ key[0])));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NG_LDS s16x4*)(img + 4096 * h + rowoff[1] + 16 * (ch ^ key[1])));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](const char* sP) {
        const char* sQ = sP + P_BYTES;
#pragma unroll
        for (int h = 0; h < MS / 16; ++h) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                a[e] = frag(sP, h, a_ch[e]);
                b[e] = frag(sQ, h, b_ch[e]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[e], b[f], acc[e][f], 0, 0, 0);
        }
    };

    if (nk > 0) {
        issue(st0, mstart);
        int s = 0;
        for (; s + 2 < nk; s += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue(st1, mstart + (s + 1) * MS);
            compute(st0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue(st0, mstart + (s + 2) * MS);
            compute(st1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) {
            issue(st1, mstart + (s + 1) * MS);
            compute(st0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(st1);
        } else {
            compute(st0);
        }
    }

    // ---------------- store the partial tile through LDS: accumulator row i = (r&3)+8*(r>>2)+4*half is n = wr*64 + e*32 + i
    __syncthreads();
    {
        float* half_base = reinterpret_cast<float*>(wr == 0 ? st0 : st1);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                    half_base[(e * 32 + i) * 128 + wc * 64 + f * 32 + (lane & 31)] = acc[e][f][r];
                }
    }
    __syncthreads();
    {
        float* slab = p.slabs + size_t(split) * p.N * p.K;
        const int chunk = tid & 31, row0 = tid >> 5;
        const int jj = j0 + chunk * 4;
        if (jj < p.K) {
#pragma unroll 4
            for (int row = row0; row < TN; row += 8) {
                const int n = n0 + row;
                if (n < p.N) {
                    const float* src = reinterpret_cast<const float*>(row < TN / 2 ? st0 : st1) + (row % (TN / 2)) * 128 + chunk * 4;
                    *reinterpret_cast<f32x4*>(slab + size_t(n) * p.K + jj) = *reinterpret_cast<const f32x4*>(src);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- host: descriptor -> parameters
inline int build_conv_params(const nirgan_conv_desc* d, ConvParams& p) {
    NG_REQUIRE(d != nullptr, "conv: null descriptor");
    NG_REQUIRE(d->in && d->w && d->out && d->zero_page, "conv: null pointer");
    NG_REQUIRE(ng_aligned16(d->in) && ng_aligned16(d->w) && ng_aligned16(d->zero_page) && ng_aligned16(d->out), "conv: in/w/out/zero_page must be 16-byte aligned");
    NG_REQUIRE(d->out_cs % 4 == 0 || d->N < 4, "conv: out_cs must be a multiple of 4");
    NG_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->N > 0, "conv: empty problem B=%d OH=%d OW=%d N=%d", d->B, d->OH, d->OW, d->N);
    NG_REQUIRE(d->ntaps >= 1 && d->ntaps <= NIRGAN_MAX_TAPS, "conv: ntaps=%d out of range", d->ntaps);
    NG_REQUIRE(d->run > 0 && d->run % 4 == 0 && d->in_cs > 0 && d->in_cs % 4 == 0, "conv: run=%d and in_cs=%d must be positive multiples of 4", d->run, d->in_cs);
    NG_REQUIRE(d->in_stride >= 1 && d->out_stride >= 1, "conv: strides must be >= 1");
    NG_REQUIRE(d->in_elems < (int64_t(1) << 31) && d->out_elems < (int64_t(1) << 31) && d->w_elems < (int64_t(1) << 31), "conv: buffers must be < 2^31 floats");
    NG_REQUIRE(d->in_elems >= int64_t(d->B) * d->in_hp * d->in_wp * d->in_cs, "conv: in_elems too small");
    NG_REQUIRE(d->out_elems >= int64_t(d->B) * d->out_hp * d->out_wp * d->out_cs, "conv: out_elems too small");
    NG_REQUIRE(d->w_elems >= int64_t(d->N) * d->ntaps * d->run, "conv: w_elems too small");
    const int span = d->out_span > 1 ? d->out_span : 1;
    // (out_cs == N with out_span = 2: the caller describes the output in pixel PAIRS already -- one 'pixel' of N channels per GEMM row, any
    // stride; only the per-channel records of the statistics follow out_span then)
    const bool span_view = span == 2 && d->out_cs == d->N;
    NG_REQUIRE(span == 1 || (span == 2 && d->precision == 3 && d->N % 8 == 0 && (span_view ? d->fuse_y == nullptr : (d->out_cs == d->N / 2 && d->out_stride >= 2))
                             && !d->out_bf16 && !d->fuse_y_bf16 && d->ksplit <= 1 && d->w_x3 != nullptr),
               "conv: out_span=%d needs precision 3 with its weight planes, out_cs == N / 2 and out_stride >= 2 (or out_cs == N), fp32 tensors and no split-K", d->out_span);
    const int ch = d->N / span;
    NG_REQUIRE(ch <= d->out_cs, "conv: %d channels per pixel exceed out_cs=%d", ch, d->out_cs);
    int dh0 = d->tap_dh[0], dh1 = d->tap_dh[0], dw0 = d->tap_dw[0], dw1 = d->tap_dw[0];
    for (int t = 1; t < d->ntaps; ++t) {
        dh0 = d->tap_dh[t] < dh0 ? d->tap_dh[t] : dh0; dh1 = d->tap_dh[t] > dh1 ? d->tap_dh[t] : dh1;
        dw0 = d->tap_dw[t] < dw0 ? d->tap_dw[t] : dw0; dw1 = d->tap_dw[t] > dw1 ? d->tap_dw[t] : dw1;
    }
    NG_REQUIRE(d->in_oh + dh0 >= 0 && (d->OH - 1) * d->in_stride + d->in_oh + dh1 < d->in_hp, "conv: input rows out of range");
    NG_REQUIRE(d->in_ow + dw0 >= 0 && int64_t((d->OW - 1) * d->in_stride + d->in_ow + dw1) * d->in_cs + d->run <= int64_t(d->in_wp) * d->in_cs, "conv: input columns out of range");
    NG_REQUIRE(d->out_oh >= 0 && (d->OH - 1) * d->out_stride + d->out_oh < d->out_hp && d->out_ow >= 0 && (d->OW - 1) * d->out_stride + d->out_ow + (span_view ? 0 : span - 1) < d->out_wp, "conv: output window out of range");

    p.in = d->in; p.w = d->w; p.bias = d->bias; p.out = d->out; p.zero = d->zero_page;
    p.in_cs = d->in_cs; p.in_row = d->in_wp * d->in_cs; p.in_img = d->in_hp * p.in_row;
    p.run = d->run; p.in_stride = d->in_stride; p.in_org = d->in_oh * p.in_row + d->in_ow * d->in_cs;
    p.ntaps = d->ntaps;
    for (int t = 0; t < NIRGAN_MAX_TAPS; ++t) p.tap_off[t] = t < d->ntaps ? d->tap_dh[t] * p.in_row + d->tap_dw[t] * d->in_cs : 0;
    p.K = d->ntaps * d->run;
    p.out_cs = d->out_cs; p.out_row = d->out_wp * d->out_cs; p.out_img = d->out_hp * p.out_row;
    p.out_stride = d->out_stride; p.out_org = d->out_oh * p.out_row + d->out_ow * d->out_cs;
    p.OW = d->OW; p.OHW = d->OH * d->OW;
    const int64_t M = int64_t(d->B) * p.OHW;
    NG_REQUIRE(M < (int64_t(1) << 31), "conv: too many output pixels");
    p.M = int(M); p.N = d->N; p.ch = ch;
    p.mtiles = (p.M + 127) / 128;
    p.ntiles = d->N > 64 ? (d->N + 127) / 128 : 1;
    NG_REQUIRE(d->precision >= 0 && d->precision <= 3, "conv: precision=%d (0 fp32, 1 bf16, 2 bf16x3, 3 fp32 as three bf16 terms)", d->precision);
    p.prec = d->precision;
    p.w3 = static_cast<const unsigned short*>(d->w_x3);
    p.w3_plane = d->w_x3_plane;
    NG_REQUIRE(p.w3 == nullptr || (ng_aligned16(p.w3) && d->w_x3_plane >= int64_t(d->N) * d->ntaps * d->run && d->w_x3_plane % 8 == 0), "conv: w_x3 misaligned or w_x3_plane too small");
    p.w_bf16 = d->w_bf16 ? 1 : 0;
    NG_REQUIRE(!p.w_bf16 || (d->precision == 1 && d->run % 8 == 0), "conv: bf16-stored weights need precision 1 and run %% 8 == 0 (run=%d)", d->run);
    p.in_bf16 = d->in_bf16 ? 1 : 0;
    NG_REQUIRE(!p.in_bf16 || (p.w_bf16 && d->in_cs % 8 == 0), "conv: bf16 activations need bf16-stored weights and in_cs %% 8 == 0 (in_cs=%d)", d->in_cs);
    p.off32 = (d->in_elems * (p.in_bf16 ? 2 : 4) < (int64_t(1) << 32) && d->w_elems * (p.w_bf16 ? 2 : 4) < (int64_t(1) << 32)) ? 1 : 0;
    p.algo = d->algo;
    p.dbg = nullptr;
    p.ksplit = 1;
    p.split_ws = nullptr;
    p.out16 = d->out_bf16 ? 1 : 0;
    p.f_y16 = d->fuse_y_bf16 ? 1 : 0;
    NG_REQUIRE(!p.out16 || (d->N % 4 == 0 && d->out_cs % 4 == 0 && d->ksplit <= 1), "conv: a bf16 output needs N %% 4 == 0, out_cs %% 4 == 0 and no split-K (N=%d)", d->N);
    p.stats = nullptr; p.stats_chunk0 = 0; p.stats_cps = 0;
    if (d->stats_ws != nullptr) {
        NG_REQUIRE(d->ksplit <= 1 && p.OHW % 128 == 0, "conv: the instance-norm partial sums need OH*OW %% 128 == 0 and no split-K (OH*OW=%d)", p.OHW);
        NG_REQUIRE(d->stats_chunk0 >= 0 && d->stats_chunk0 + p.OHW / 64 * span <= d->stats_chunks, "conv: stats_chunk0 + OH*OW/64 (x out_span) exceeds stats_chunks");
        NG_REQUIRE(d->stats_ws_elems >= int64_t(d->B) * d->stats_chunks * 4 * ch, "conv: stats_ws too small (B * stats_chunks * 4 * channels floats)");
        p.stats = d->stats_ws; p.stats_chunk0 = d->stats_chunk0; p.stats_cps = d->stats_chunks;
    }
    p.f_y = nullptr; p.f_mean = nullptr; p.f_rstd = nullptr; p.f_part = nullptr;
    p.f_img = p.f_row = p.f_org = p.f_chunk0 = p.f_cps = 0; p.f_act = NIRGAN_ACT_NONE; p.f_slope = 0.f;
    if (d->fuse_y != nullptr) {
        NG_REQUIRE(d->ksplit <= 1 && p.OHW % 128 == 0 && d->N % 4 == 0 && d->bias == nullptr, "conv: the fused instance-norm backward sums need OH*OW %% 128 == 0, N %% 4 == 0, no split-K and no bias (OH*OW=%d N=%d)", p.OHW, d->N);
        NG_REQUIRE(d->fuse_mean && d->fuse_rstd && d->fuse_part && ng_aligned16(d->fuse_y) && ng_aligned16(d->fuse_mean) && ng_aligned16(d->fuse_rstd) && ng_aligned16(d->fuse_part), "conv: fuse_mean / fuse_rstd / fuse_part missing or misaligned");
        NG_REQUIRE(d->fuse_oh >= 0 && d->fuse_ow >= 0 && (d->OH - 1) * d->out_stride + d->fuse_oh < d->fuse_h && (d->OW - 1) * d->out_stride + d->fuse_ow + span - 1 < d->fuse_w, "conv: fused window out of y's %d x %d extent", d->fuse_h, d->fuse_w);
        NG_REQUIRE(int64_t(d->B) * d->fuse_h * d->fuse_w * ch < (int64_t(1) << 31), "conv: fuse_y must be < 2^31 floats");
        NG_REQUIRE(d->fuse_chunk0 >= 0 && d->fuse_chunk0 + p.OHW / 128 * span <= d->fuse_chunks, "conv: fuse_chunk0 + OH*OW/128 (x out_span) exceeds fuse_chunks");
        NG_REQUIRE(d->fuse_part_elems >= int64_t(d->B) * d->fuse_chunks * 2 * ch, "conv: fuse_part too small");
        NG_REQUIRE(d->fuse_act == NIRGAN_ACT_NONE || d->fuse_act == NIRGAN_ACT_RELU || d->fuse_act == NIRGAN_ACT_LRELU, "conv: fuse_act=%d", d->fuse_act);
        p.f_y = d->fuse_y; p.f_mean = d->fuse_mean; p.f_rstd = d->fuse_rstd; p.f_part = d->fuse_part;
        p.f_row = d->fuse_w * ch; p.f_img = d->fuse_h * p.f_row; p.f_org = d->fuse_oh * p.f_row + d->fuse_ow * ch;
        p.f_act = d->fuse_act; p.f_slope = d->fuse_slope; p.f_chunk0 = d->fuse_chunk0; p.f_cps = d->fuse_chunks;
    }
    if (d->ksplit > 1) {
        const int nk = d->ntaps * ((d->run + 31) / 32);
        NG_REQUIRE(d->split_ws != nullptr && ng_aligned16(d->split_ws), "conv: split_ws missing or misaligned");
        NG_REQUIRE(d->N % 4 == 0, "conv: split-K needs N %% 4 == 0");
        NG_REQUIRE(d->split_ws_elems >= int64_t(d->ksplit) * M * d->N, "conv: split_ws too small");
        const int per = (nk + d->ksplit - 1) / d->ksplit;
        NG_REQUIRE(per * (d->ksplit - 1) < nk, "conv: ksplit=%d too large for %d K-steps", d->ksplit, nk);
        p.ksplit = d->ksplit;
        p.split_ws = d->split_ws;
    }
    return NIRGAN_OK;
}

inline int build_wgrad_params(const nirgan_wgrad_desc* d, WgradParams& p) {
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

    p.p = d->p; p.q = d->q; p.slabs = d->slabs; p.zero = d->zero_page;
    p.p_cs = d->p_cs; p.p_row = d->p_wp * d->p_cs; p.p_img = d->p_hp * p.p_row; p.p_org = d->p_oh * p.p_row + d->p_ow * d->p_cs;
    p.q_cs = d->q_cs; p.q_row = d->q_wp * d->q_cs; p.q_img = d->q_hp * p.q_row; p.q_stride = d->q_stride;
    p.q_org = d->q_oh * p.q_row + d->q_ow * d->q_cs;
    p.run = d->run; p.ntaps = d->ntaps;
    for (int t = 0; t < NIRGAN_MAX_TAPS; ++t) p.tap_off[t] = t < d->ntaps ? d->tap_dh[t] * p.q_row + d->tap_dw[t] * d->q_cs : 0;
    p.K = K; p.OW = d->OW; p.OH = d->OH; p.OHW = d->OH * d->OW; p.M = int(M); p.N = d->N;
    p.rows_per_split = d->rows_per_split; p.nsplit = d->nsplit;
    p.ntiles_k = (K + 127) / 128;
    p.ntiles_n = d->N > 64 ? (d->N + 127) / 128 : 1;
    NG_REQUIRE(d->precision >= 0 && d->precision <= 3, "wgrad_igemm: precision=%d (0 fp32, 1 bf16, 2 bf16x3, 3 fp32 as three bf16 terms)", d->precision);
    p.prec = d->precision;
    p.nplanes = d->nplanes > 1 ? d->nplanes : 1;
    p.p_plane = d->p_plane; p.q_plane = d->q_plane;
    NG_REQUIRE(p.nplanes == 1 || ((d->precision == 0 || d->precision == 3) && !d->pq_bf16 && d->p_plane > 0 && d->q_plane > 0), "wgrad_igemm: planes need the fp32 tile and positive plane strides");
    NG_REQUIRE(d->slab_elems >= int64_t(p.nplanes) * d->nsplit * d->N * K, "wgrad_igemm: slab_elems too small for %d planes", p.nplanes);
    NG_REQUIRE(d->p_elems >= (p.nplanes - 1) * d->p_plane + int64_t(d->B) * d->p_hp * d->p_wp * d->p_cs && d->q_elems >= (p.nplanes - 1) * d->q_plane + int64_t(d->B) * d->q_hp * d->q_wp * d->q_cs, "wgrad_igemm: p/q too small for the planes");
    p.pq_bf16 = d->pq_bf16 ? 1 : 0;
    p.dbg = nullptr;
    {
        bool taps_ok = true;
        for (int t = 0; t < d->ntaps; ++t) taps_ok = taps_ok && (d->tap_dh[t] * p.q_row + d->tap_dw[t] * d->q_cs >= 0);
        const int ms = d->pq_bf16 ? 64 : 32, es = d->pq_bf16 ? 2 : 4;          // pixels per K-step, bytes per stored element
        p.fast32_bytes = (d->p_elems * es < (int64_t(1) << 32) && d->q_elems * es < (int64_t(1) << 32)) ? 1 : 0;
        p.fast32 = (taps_ok && d->OW % ms == 0 && p.M % ms == 0 && d->rows_per_split % ms == 0
                    && d->p_elems * es < (int64_t(1) << 32) && d->q_elems * es < (int64_t(1) << 32)) ? 1 : 0;
    }
    NG_REQUIRE(!p.pq_bf16 || (d->precision == 1 && d->N > 64 && d->N % 8 == 0 && d->run % 8 == 0 && d->p_cs % 8 == 0 && d->q_cs % 8 == 0),
               "wgrad_igemm: bf16 twins need precision 1, N > 64 and N, run, p_cs, q_cs multiples of 8");
    return NIRGAN_OK;
}

}  // namespace ng
