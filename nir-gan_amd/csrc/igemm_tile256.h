// The 256 x 256 x 64 implicit-GEMM tile of the bf16 operand mode (round 4): both operands stored as bf16 (a producer's twin, the
// pack kernel's bf16 weights), fp32 accumulate.  ONE workgroup of eight waves per CU, 128 KB of LDS in one array, the K loop as
// eight phases per two K-tiles with the LDS-DMA of later K-tiles in flight ACROSS the barriers (counted vmcnt, raw s_barrier) --
// the structure cdna_hip_programming.md section 5 describes ("The 256^2 8-phase template"), here with the A operand gathered from a
// halo'd NHWC activation buffer through the tap table of an nirgan_conv_desc (model/networks.py:405-427: the ResnetBlock
// convolutions; their data gradients as flipped-weight correlations).
//
//   waves        8 = 2 (wr: rows) x 4 (wc: columns); wave (wr, wc) owns the four 64 x 32 quadrants
//                rows i * 128 + wr * 64 .. + 63, columns wc * 64 + j * 32 .. + 31 (i, j = 0, 1) of the 256 x 256 block tile
//   LDS          [2 K-tile buffers][A half 0, A half 1, B half 0, B half 1] x 16 KB; a half = 128 rows x 64 k bf16 in 128-byte rows,
//                16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (applied on the DMA's source side and on the read side)
//                B half j, row r holds weight row n0 + (r >> 5) * 64 + j * 32 + (r & 31): a wave's two column quadrants are adjacent
//   MFMA         v_mfma_f32_16x16x32_bf16: quadrant x 64 k = 4 x 2 tiles x 2 k-steps = 16 per phase
//   phases       K-tile k (buffer b): .1 read A half 0, C00;  .2 read B half 1, C01;  .3 read A half 1, C10;  .4 read B half 0 of k+1, C11
//   LDS-DMA      one half-tile (2 pieces per wave) per phase: .1 A1(k+1)  .2 B0(k+2)  .3 A0(k+2)  .4 B1(k+2); every phase ends with
//                vmcnt(10): all but the five youngest halves have landed, i.e. the half issued five phases ago, read from the NEXT phase on
//   stagger      waves 4-7 run one barrier behind waves 0-3: on every SIMD one wave is in its MFMA segment while its partner issues
//                fragment reads and DMA pieces
// Hazards (the guide's placement rules): a half is read one phase AFTER the wait that retires its DMA, and restaged two phases after
// its last fragment read (t256_kloop's table).
#pragma once
#include "igemm_tiles.h"
#include <type_traits>

namespace ng {

constexpr int T256_HALF = 128 * 128;          // bytes of one half-tile image
constexpr int T256_LDS = 8 * T256_HALF;       // 128 KB: the ring of 8 slots (and the epilogue's 8 wave-private staging areas)
constexpr int T256_LDS10 = 10 * T256_HALF;    // 160 KB: the ring of 10 slots

__device__ __forceinline__ void t256_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// The K loop of the eight-phase structure, shared by the convolution tile and the weight-gradient tile.  The callables work on the
// caller's registers: issueA / issueB(slot, h) start the LDS-DMA of half h (rows h * 128 .. of the tile) of the cursor's K-tile into
// ring slot `slot`, advance() moves the cursor to the next K-tile, readA(slot) / readB(which, slot) issue the fragment reads (B into
// register set `which`), mma(i, j) waits for them and accumulates quadrant (i, j) from the A registers and B set j.
//
// The half-tiles of the K loop form ONE sequence h = 4 k + {0: B half 0, 1: A half 0, 2: B half 1, 3: A half 1} of K-tile k, and the
// LDS holds a ring of S slots of 16 KB, half h in slot h mod S.  Global phase g = 4 k + part does three things:
//     fragment reads of half g + 1            (part 0: A half 0, 1: B half 1, 2: A half 1, 3: B half 0 of K-tile k + 1)
//     16 MFMAs of quadrant part               (C00 = A0 B0, C01 = A0 B1, C10 = A1 B0, C11 = A1 B1)
//     LDS-DMA of half g + S - 1 into the slot that half g - 1 left two phases ago, then vmcnt(2 (S - 3)): all but the S - 3 youngest
//     halves have landed, i.e. half g + 2 -- read in the NEXT phase
// Every slot lives one ring period: issued, retired by the wait S - 3 phases later, read in the phase after that, restaged two phases
// after the read.  S = 8 (128 KB) keeps five halves in flight, S = 10 (all 160 KB of a CU's LDS) seven.  The weight-gradient tile
// streams every operand byte from beyond L2 once (the convolution tile re-reads its patch nine times from L2) and takes 3 010 cycles
// per K-tile against 2 690 with the operands pinned in L2 (profiles/r04_tile256_stamps.txt) -- but NOT for want of prefetch distance:
// the ring of 10 measures 3 039 (profiles/r04_tile256_ring10.txt).  What the farther operand costs is the ISSUE of the LDS-DMA pieces
// (the load segment's length), not the wait for them: the CU's outstanding-request capacity, not the ring, bounds what is in flight.
// The slot numbers are literals: the loop is unrolled over lcm(4, S) phases (2 K-tiles for S = 8, 5 for S = 10).  The fragment reads are
// spread 2 : 1 : 2 : 1 over the phases; the first form of this loop (round 4, first day) read A half 0 and B half 0 together and waited
// vmcnt(6) once per K-tile.  In the last K-tiles, where fewer than S - 3 younger halves exist, the waits drain (vmcnt(0)).
template <int S> __device__ __forceinline__ void t256_wait_ring() {
    static_assert(S == 8 || S == 10, "ring of 8 or 10 half-tile slots");
    if constexpr (S == 8) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
}

template <int S, int G, class IA, class IB, class ADV, class RA, class RB, class MMA>
__device__ __forceinline__ void t256_phase(const int k, const int nk, IA& issueA, IB& issueB, ADV& advance, RA& readA, RB& readB, MMA& mma) {
    constexpr int part = G & 3;
    constexpr int jr = (G + 1) & 3, sr = (G + 1) % S;                  // the half whose fragments this phase reads: type, slot
    constexpr int ji = (G + S - 1) & 3, si = (G + S - 1) % S;          // the half this phase stages
    if (k + (part + 1) / 4 < nk) {
        if constexpr (jr == 0) readB(0, sr);
        else if constexpr (jr == 1) readA(sr);
        else if constexpr (jr == 2) readB(1, sr);
        else readA(sr);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (the LDS-DMA issued BEFORE the phase's fragment reads was measured: 3 150 against 3 010 cycles per K-tile for the weight-gradient
    // tile, 2 810 against 2 575 for the convolution tile -- the reads' latency then shows at the head of the MFMA segment)
    if (k + (part + S - 1) / 4 < nk) {
        if constexpr (ji == 0) issueB(si, 0);
        else if constexpr (ji == 1) issueA(si, 0);
        else if constexpr (ji == 2) issueB(si, 1);
        else { issueA(si, 1); advance(); }
        t256_wait_ring<S>();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    t256_bar();
    mma(part >> 1, part & 1);
    t256_bar();
}

// K-tiles KT .. of one unrolled period; returns when the K-tiles run out
template <int S, int KT, class IA, class IB, class ADV, class RA, class RB, class MMA>
__device__ __forceinline__ void t256_period(const int k0, const int nk, IA& issueA, IB& issueB, ADV& advance, RA& readA, RB& readB, MMA& mma) {
    constexpr int KTS = (S == 8 ? 8 : 20) / 4;
    if constexpr (KT < KTS) {
        if (k0 + KT >= nk) return;
        t256_phase<S, 4 * KT + 0>(k0 + KT, nk, issueA, issueB, advance, readA, readB, mma);
        t256_phase<S, 4 * KT + 1>(k0 + KT, nk, issueA, issueB, advance, readA, readB, mma);
        t256_phase<S, 4 * KT + 2>(k0 + KT, nk, issueA, issueB, advance, readA, readB, mma);
        t256_phase<S, 4 * KT + 3>(k0 + KT, nk, issueA, issueB, advance, readA, readB, mma);
        t256_period<S, KT + 1>(k0, nk, issueA, issueB, advance, readA, readB, mma);
    }
}

template <int S, int H, class IA, class IB, class ADV>
__device__ __forceinline__ void t256_prologue_issue(const int nk, IA& issueA, IB& issueB, ADV& advance) {
    if constexpr (H < S - 1) {
        if ((H >> 2) < nk) {
            if constexpr ((H & 3) == 0) issueB(H, 0);
            else if constexpr ((H & 3) == 1) issueA(H, 0);
            else if constexpr ((H & 3) == 2) issueB(H, 1);
            else { issueA(H, 1); advance(); }
        }
        t256_prologue_issue<S, H + 1>(nk, issueA, issueB, advance);
    }
}

template <int S, class IA, class IB, class ADV, class RA, class RB, class MMA>
__device__ __forceinline__ void t256_kloop(const int nk, const int wr, IA&& issueA, IB&& issueB, ADV&& advance, RA&& readA, RB&& readB, MMA&& mma) {
    // prologue: halves 0 .. S - 2, then "phase -1": the fragment reads of half 0 (B half 0 of K-tile 0)
    t256_prologue_issue<S, 0>(nk, issueA, issueB, advance);
    const bool full = 4 * nk >= S - 1;        // all S - 1 halves exist: the counted waits apply
    if (full) {
        if constexpr (S == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // half 0: all but the S - 2 younger ones
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    t256_bar();                               // ... from every wave
    if (wr == 1) t256_bar();                  // the stagger: waves 4-7 run one barrier behind from here on
    readB(0, 0);
    if (full) t256_wait_ring<S>();            // half 1
    t256_bar();
    t256_bar();                               // (the empty MFMA segment of that phase: keeps the two wave groups half a phase apart)
    constexpr int KTS = (S == 8 ? 8 : 20) / 4;
    for (int k0 = 0; k0 < nk; k0 += KTS) t256_period<S, 0>(k0, nk, issueA, issueB, advance, readA, readB, mma);
    if (wr == 0) t256_bar();                  // waves 0-3 wait for the staggered half: every fragment read and every DMA is done
}

// in-kernel stamps of the diagnostic build (scripts/diag/tile256_stamp.hip defines NG_DIAG256); none executes in the product build
#ifdef NG_DIAG256
#define T256_STAMP(i) { __builtin_amdgcn_sched_barrier(0); ng_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define T256_STAMP(i)
#endif

// F32 = true: the SAME structure in exact fp32 (v_mfma_f32_32x32x2_f32, the parity path's arithmetic): a K-tile is 32 k (the 128-byte
// rows hold 32 floats), a phase is 2 x 1 tiles of 32 x 32 x 16 k-steps of 2 = 32 MFMAs of 64 cycles, so the load segment of the partner
// wave (the same LDS-DMA pieces and fragment reads as in bf16) hides under 2 048 instead of 256 cycles of matrix work.  Fragments as in
// conv_tile: one ds_read_b128 hands a lane 4 consecutive k of its row, lanes 0-31 chunk 2g, lanes 32-63 chunk 2g + 1, MFMA j of group g
// contracts k = 8g + j and 8g + 4 + j.  `in_base / w_base / out_base`: base overrides for plane-batched launches (csrc/wino6.hip).
template <bool F32 = false, int S = 8>
__device__ __forceinline__ void conv_tile256(const ConvParams& p, const int id, char* lds,
                                             const float* in_base = nullptr, const float* w_base = nullptr, float* out_base = nullptr) {
    const float* const p_in = in_base ? in_base : p.in;
    const float* const p_w = w_base ? w_base : p.w;
    float* const p_out = out_base ? out_base : p.out;
    constexpr int ES = F32 ? 4 : 2;             // bytes per operand element
    constexpr int KT = F32 ? 32 : 64;           // k per K-tile (one 128-byte row)
    constexpr int CW = 16 / ES;                 // elements per 16-byte chunk
#ifdef NG_DIAG256
    unsigned long long ng_t[8];
    const unsigned long long ng_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    T256_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int mt256 = (p.M + 255) >> 8, nt256 = (p.N + 255) >> 8;
    const int n0 = (id % nt256) * 256, m0 = (id / nt256) * 256;

    // ---------------- loader state: per half-tile a wave issues pieces 2 wave, 2 wave + 1 (8 rows x 128 B each)
    unsigned a_boff[2][2], b_boff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + h * 128 + row;
            m = m < p.M ? m : p.M - 1;
            const int b = m / p.OHW, r = m - b * p.OHW;
            const int oh = r / p.OW, ow = r - oh * p.OW;
            a_boff[h][i] = unsigned(b * p.in_img + oh * p.in_stride * p.in_row + ow * p.in_stride * p.in_cs + p.in_org + lc * CW) * unsigned(ES);
            int n = n0 + (row >> 5) * 64 + h * 32 + (row & 31);
            n = n < p.N ? n : 0;                 // (columns past N are never stored and never summed)
            b_boff[h][i] = unsigned(n * p.K + lc * CW) * unsigned(ES);
        }
    // K-tiles in slice-major order: 64-channel slice cc of the run, all taps -- a slice of the tile's input patch (one 128-byte line per
    // pixel) stays in L2 across its taps.  The tap offsets sit in one VGPR (lane t holds tap t): a cursor step costs no memory access.
    const int tapv = p.tap_off[lane & (NIRGAN_MAX_TAPS - 1)];
    const int nk = p.ntaps * (p.run / KT);
    int ct = 0, cc = 0;
    auto advance = [&]() {
        ++ct;
        if (ct == p.ntaps) { ct = 0; cc += KT; }
    };
    const char* const in8 = reinterpret_cast<const char*>(p_in);
    const char* const w8 = reinterpret_cast<const char*>(p_w);
    auto issueA = [&](const int slot, const int h) {
#ifdef NG_DIAG_SHARE_A          // diagnostic build only (wrong results): the A stage of tap dw = 0 stands for the taps dw = 1, 2 of its kernel row --
        if (ct % 3 != 0) return; // an upper bound of what one A stage per kernel row would buy the K loop
#endif
        const int toff = __builtin_amdgcn_readlane(tapv, ct);
        const char* base = ng_uniform_ptr(in8 + (long long)(toff + cc) * ES);
        char* dst = lds + slot * T256_HALF + wave * 2048;
        ng_glds16_so(base, a_boff[h][0], dst);
        ng_glds16_so(base, a_boff[h][1], dst + 1024);
    };
    auto issueB = [&](const int slot, const int h) {
        const char* base = ng_uniform_ptr(w8 + (long long)(ct * p.run + cc) * ES);
        char* dst = lds + slot * T256_HALF + wave * 2048;
        ng_glds16_so(base, b_boff[h][0], dst);
        ng_glds16_so(base, b_boff[h][1], dst + 1024);
    };

    // ---------------- compute state
    const int key = (lane & 15) >> 1;
    const int x0 = ((lane >> 4) ^ key) << 4, x1 = (((lane >> 4) | 4) ^ key) << 4;
    const int a_rd0 = (wr * 64 + (lane & 15)) * 128 + x0, a_rd1 = (wr * 64 + (lane & 15)) * 128 + x1;
    const int b_rd0 = (wc * 32 + (lane & 15)) * 128 + x0, b_rd1 = (wc * 32 + (lane & 15)) * 128 + x1;
    bf16x8 A[4][2], B0[2][2], B1[2][2];
    f32x4 acc[2][2][4][2];
    // fp32: rows wr * 64 + mt * 32 + (lane & 31) of an A half, wc * 32 + (lane & 31) of a B half; group g reads chunk 2g + half
    const int half = lane >> 5, keyf = (lane >> 1) & 7;
    int f_rd[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) f_rd[g] = (lane & 31) * 128 + (((2 * g + half) ^ keyf) << 4);
    f32x4 Af[2][4], B0f[4], B1f[4];
    f32x16 accF[2][2][2];
    if constexpr (F32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accF[i][j][mt][r] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto readA = [&](const int slot) {
        const char* s = lds + slot * T256_HALF;
        if constexpr (F32) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) Af[mt][g] = *reinterpret_cast<const f32x4*>(s + (wr * 64 + mt * 32) * 128 + f_rd[g]);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                A[mt][0] = *reinterpret_cast<const bf16x8*>(s + a_rd0 + mt * 2048);
                A[mt][1] = *reinterpret_cast<const bf16x8*>(s + a_rd1 + mt * 2048);
            }
        }
    };
    auto readB = [&](const int which, const int slot) {
        const char* s = lds + slot * T256_HALF;
        if constexpr (F32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(s + (wc * 32) * 128 + f_rd[g]);
                if (which == 0) B0f[g] = v; else B1f[g] = v;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(s + b_rd0 + nt * 2048), v1 = *reinterpret_cast<const bf16x8*>(s + b_rd1 + nt * 2048);
                if (which == 0) { B0[nt][0] = v0; B0[nt][1] = v1; } else { B1[nt][0] = v0; B1[nt][1] = v1; }
            }
        }
    };
    auto mma = [&](const int i, const int j) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F32) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        accF[i][j][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[mt][g][q], j == 0 ? B0f[g][q] : B1f[g][q], accF[i][j][mt], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[i][j][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt][s], j == 0 ? B0[nt][s] : B1[nt][s], acc[i][j][mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    T256_STAMP(1)
    t256_kloop<S>(nk, wr, issueA, issueB, advance, readA, readB,
               [&](const int i, const int j) {         // (i, j are literals at every call site: the branches fold after inlining)
                   if (i == 0 && j == 0) mma(0, 0); else if (i == 0) mma(0, 1); else if (j == 0) mma(1, 0); else mma(1, 1);
               });
    T256_STAMP(2)

    // ---------------- partial sums for the instance norm that follows (nirgan_conv_desc.stats_ws): a wave's 64 rows x 64 columns leave,
    // per column, {k = the chunk's first row, sum (v - k), sum (v - k)^2, 64} -- same contract as conv_tile, chunk = 64 output pixels
    if (p.stats != nullptr) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mrow = m0 + i * 128 + wr * 64;
            if (mrow < p.M) {                 // (host: OH*OW % 128 == 0, so a 64-row chunk is whole and inside one sample)
                const int b = mrow / p.OHW;
                float* sp = p.stats + (size_t(b) * p.stats_cps + p.stats_chunk0 + ((mrow - b * p.OHW) >> 6)) * 4 * p.N;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (F32 && nt == 1) continue;             // fp32: one 32-column tile per (i, j), a column on lanes l and l + 32
                        float k0, s1 = 0.f, s2 = 0.f;
                        if constexpr (F32) {
                            k0 = __shfl(accF[i][j][0][0], lane & 31, 64);
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float v = accF[i][j][mt][r] - k0;
                                    s1 += v;
                                    s2 += v * v;
                                }
                        } else {
                            k0 = __shfl(acc[i][j][0][nt][0], lane & 15, 64);
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float v = acc[i][j][mt][nt][r] - k0;
                                    s1 += v;
                                    s2 += v * v;
                                }
                            s1 += __shfl_xor(s1, 16, 64);
                            s2 += __shfl_xor(s2, 16, 64);
                        }
                        s1 += __shfl_xor(s1, 32, 64);
                        s2 += __shfl_xor(s2, 32, 64);
                        const int col = n0 + wc * 64 + j * 32 + (F32 ? (lane & 31) : nt * 16 + (lane & 15));
                        if (lane < (F32 ? 32 : 16) && col < p.N) {
                            sp[col] = k0;
                            sp[p.N + col] = s1;
                            sp[2 * p.N + col] = s2;
                            sp[3 * p.N + col] = 64.f;
                        }
                    }
            }
        }
    }

    T256_STAMP(3)
    // ---------------- epilogue: each wave transposes its 64 x 64 (row half i) through its OWN 16 KB of LDS -- no workgroup barrier --
    // and stores whole 256-byte (fp32) / 128-byte (bf16) row segments, 16 bytes per lane either way (the store path is bound by
    // the number of store instructions: 8-byte stores of a bf16 output took 16 k cycles per tile, twice the 16-byte ones)
    float* const stg = reinterpret_cast<float*>(lds + wave * 16384);
    const int OH = p.OHW / p.OW;
    const bool fused = p.f_y != nullptr;
    const float fneg = p.f_act == NIRGAN_ACT_RELU ? 0.f : (p.f_act == NIRGAN_ACT_LRELU ? p.f_slope : 1.f);
    auto half_out = [&](const int i, auto wtag) {
        constexpr int W = decltype(wtag)::value;          // output channels per lane: 4 (fp32 output) or 8 (bf16 output)
        constexpr int LPR = 64 / W, RPP = 64 / LPR;       // lanes per row, rows per pass
        typedef float fW __attribute__((ext_vector_type(W)));
        typedef __bf16 bW __attribute__((ext_vector_type(W)));
        const int chunk = lane % LPR, lrow = lane / LPR;
        const int n = n0 + wc * 64 + chunk * W;
        const bool n_ok = n < p.N;                         // (host: N % 256 == 0)
        fW bv = {};
        if (p.bias != nullptr && n_ok) bv = *reinterpret_cast<const fW*>(p.bias + n);
        const int mbase = m0 + i * 128 + wr * 64;
        int m = mbase + lrow;
        const int mc = m < p.M ? m : p.M - 1;
        int b = mc / p.OHW;
        const int r0 = mc - b * p.OHW;
        int oh = r0 / p.OW, ow = r0 - oh * p.OW;
        const int fb = (mbase < p.M ? mbase : p.M - 1) / p.OHW;          // (fused: one sample per 128-row half, host: OH*OW % 128 == 0)
        fW fm = {}, fr = {}, s1 = {}, s2 = {};
        if (fused && n_ok) {
            fm = *reinterpret_cast<const fW*>(p.f_mean + size_t(fb) * p.N + n);
            fr = *reinterpret_cast<const fW*>(p.f_rstd + size_t(fb) * p.N + n);
        }
#pragma unroll 4
        for (int pass = 0; pass < 64 / RPP; ++pass) {
            if (m < p.M && n_ok) {
                fW v = *reinterpret_cast<const fW*>(stg + (pass * RPP + lrow) * 64 + chunk * W);
                v += bv;
                const int oidx = b * p.out_img + oh * p.out_stride * p.out_row + ow * p.out_stride * p.out_cs + p.out_org + n;
                if constexpr (W == 8) *reinterpret_cast<bW*>(reinterpret_cast<unsigned short*>(p_out) + oidx) = __builtin_convertvector(v, bW);
                else *reinterpret_cast<fW*>(p_out + oidx) = v;
                if (fused) {
                    const size_t yidx = size_t(b) * p.f_img + size_t(oh * p.out_stride) * p.f_row + size_t(ow * p.out_stride) * p.N + p.f_org + n;
                    fW y;
                    if (p.f_y16) y = __builtin_convertvector(*reinterpret_cast<const bW*>(reinterpret_cast<const unsigned short*>(p.f_y) + yidx), fW);
                    else y = *reinterpret_cast<const fW*>(p.f_y + yidx);
                    const fW z = (y - fm) * fr;
#pragma unroll
                    for (int q = 0; q < W; ++q) {
                        const float gz = z[q] > 0.f ? v[q] : v[q] * fneg;
                        s1[q] += gz;
                        s2[q] += gz * z[q];
                    }
                }
            }
            m += RPP;
            ow += RPP;
            while (ow >= p.OW) { ow -= p.OW; ++oh; }
            while (oh >= OH) { oh -= OH; ++b; }
        }
        if (fused) {
            // first pass of the consumer layer's instance-norm backward: this wave's 64 rows, then the two waves of a 128-row chunk
            // (wr = 0, 1) join through LDS in a fixed order.  All of a wave's staging reads are done (same wave, program order).
#pragma unroll
            for (int q = 0; q < W; ++q)
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
                    s1[q] += __shfl_xor(s1[q], o, 64);
                    s2[q] += __shfl_xor(s2[q], o, 64);
                }
            t256_bar();                                             // (uniform: `fused` is a launch constant) every wave's staging reads are done
            fW* const red = reinterpret_cast<fW*>(lds);             // 8 waves x 64 columns x 2 sums: 4 KB
            if (lane < LPR) {
                red[(wave * LPR + chunk) * 2] = s1;
                red[(wave * LPR + chunk) * 2 + 1] = s2;
            }
            t256_bar();
            if (wr == 0 && lane < LPR && n_ok && mbase < p.M) {
                const fW t1 = s1 + red[((wave + 4) * LPR + chunk) * 2], t2 = s2 + red[((wave + 4) * LPR + chunk) * 2 + 1];
                float* pp = p.f_part + (size_t(fb) * p.f_cps + p.f_chunk0 + ((m0 + i * 128 - fb * p.OHW) >> 7)) * 2 * p.N + n;
                *reinterpret_cast<fW*>(pp) = t1;
                *reinterpret_cast<fW*>(pp + p.N) = t2;
            }
            t256_bar();
        }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (F32) {                 // C/D of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stg[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + (lane & 31)] = accF[i][j][mt][r];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            stg[(mt * 16 + (lane >> 4) * 4 + r) * 64 + j * 32 + nt * 16 + (lane & 15)] = acc[i][j][mt][nt][r];
        }
        if (p.out16) half_out(i, std::integral_constant<int, 8>{});
        else half_out(i, std::integral_constant<int, 4>{});
    }
#ifdef NG_DIAG256
    T256_STAMP(4)
    if (lane == 0 && p.dbg != nullptr) {
        unsigned long long* o = p.dbg + (size_t(id) * 8 + wave) * 7;
        for (int q = 0; q < 5; ++q) o[q] = ng_t[q];
        o[5] = ng_r0;
        o[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// The weight gradient on the same structure: slab[split][n][J] = sum over the split's pixels m of P[m][n] * Q[m + tap(J)][c(J)], both
// operands from the producers' bf16 twins (model/networks.py:405-427 through autograd: dW of the ResnetBlock convolutions).  A unit =
// 256 rows n x 256 columns J x one split; a K-tile = 64 consecutive pixels.  The reduction index (the pixel) is the ROW of both
// images, the MFMA wants it contiguous per lane: the half-tiles are 64 pixel rows x 128 columns in 256-byte rows and the fragments are
// read with ds_read_b64_tr_b16 (a 16-lane group reads a 4-row x 16-column block, lane i receives column i): two reads = one
// 16x16x32 operand.  Chunk c of row r sits at chunk c ^ 2 * ((r & 3) | ((r >> 3) & 1) << 2): the 8 rows a 32-lane half touches land on
// 8 distinct 32-byte bank segments.  Q half j, LDS column x holds column j0 + (x >> 5) * 64 + j * 32 + (x & 31) of J (as the
// convolution tile's B halves).  Host: OW % 64 == 0 or 64 % OW == 0, OH * OW % 64 == 0, rows_per_split % 64 == 0 -- a K-tile is a
// fixed pattern of pixels relative to its first one, which walks in scalar registers.
template <int S = 8>
__device__ __forceinline__ void wgrad_tile256(const WgradParams& p, const int unit, char* lds) {
#ifdef NG_DIAG256
    unsigned long long ng_t[8];
    const unsigned long long ng_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    T256_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = p.N >> 8, ntj = p.K >> 8;
    const int nt_ = unit % ntn, rest = unit / ntn;
    const int jt = rest % ntj, split = rest / ntj;
    const int n0 = nt_ * 256, j0 = jt * 256;
    const int mstart = split * p.rows_per_split;
    int mend = mstart + p.rows_per_split;
    mend = mend < p.M ? mend : p.M;
    const int nk = mend > mstart ? (mend - mstart) >> 6 : 0;

    // ---------------- loader state
    const bool wide = (p.OW & 63) == 0;               // a K-tile is 64 pixels of one image row; otherwise 64 / OW whole rows
    unsigned p_boff[2][2], q_boff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + (lane >> 4);
            const int lc = (lane & 15) ^ (2 * ((row & 3) | (((row >> 3) & 1) << 2)));
            const int rr = wide ? 0 : row / p.OW, rc = wide ? row : row - rr * p.OW;
            p_boff[h][i] = unsigned(rr * p.p_row + rc * p.p_cs + h * 128 + lc * 8) * 2u;
            const int J = j0 + ((lc >> 2) * 8 + h * 4 + (lc & 3)) * 8;
            const int t = J / p.run;
            q_boff[h][i] = unsigned(rr * p.q_stride * p.q_row + rc * p.q_stride * p.q_cs + p.tap_off[t] + (J - t * p.run)) * 2u;
        }
    // the K-tile cursor: first pixel (soh, sow) of the tile and the two base addresses, advanced by additions in scalar registers
    const int sb0 = mstart / p.OHW;
    int soh = (mstart - sb0 * p.OHW) / p.OW;
    int sow = mstart - sb0 * p.OHW - soh * p.OW;
    const char* pcur = reinterpret_cast<const char*>(p.p) + ((long long)sb0 * p.p_img + (long long)soh * p.p_row + sow * p.p_cs + p.p_org + n0) * 2;
    const char* qcur = reinterpret_cast<const char*>(p.q) + ((long long)sb0 * p.q_img + (long long)soh * p.q_stride * p.q_row + sow * p.q_stride * p.q_cs + p.q_org) * 2;
    const int rows_step = wide ? 1 : 64 / p.OW;                         // image rows a K-tile advances when it wraps / always
    const long long p_step = wide ? 128ll * p.p_cs : 2ll * rows_step * p.p_row;
    const long long q_step = wide ? 128ll * p.q_stride * p.q_cs : 2ll * rows_step * p.q_stride * p.q_row;
    const long long p_wrap = 2ll * (p.p_row - p.OW * p.p_cs), q_wrap = 2ll * p.q_stride * (p.q_row - p.OW * p.q_cs);       // wide: end of an image row
    const long long p_img_wrap = 2ll * (p.p_img - (long long)p.OH * p.p_row), q_img_wrap = 2ll * (p.q_img - (long long)p.OH * p.q_stride * p.q_row);
    auto advance = [&]() {
#ifdef NG_DIAG_NOADVANCE          // diagnostic build only: every K-tile re-reads the first one (L2-resident operands: what does the memory side cost?)
        return;
#endif
        pcur += p_step;
        qcur += q_step;
        if (wide) {
            sow += 64;
            if (sow >= p.OW) { sow = 0; ++soh; pcur += p_wrap; qcur += q_wrap; }
        } else {
            soh += rows_step;
        }
        if (soh >= p.OH) { soh = 0; pcur += p_img_wrap; qcur += q_img_wrap; }
    };
    auto issueA = [&](const int slot, const int h) {
        const char* base = ng_uniform_ptr(pcur);
        char* dst = lds + slot * T256_HALF + wave * 2048;
        ng_glds16_so(base, p_boff[h][0], dst);
        ng_glds16_so(base, p_boff[h][1], dst + 1024);
    };
    auto issueB = [&](const int slot, const int h) {
        const char* base = ng_uniform_ptr(qcur);
        char* dst = lds + slot * T256_HALF + wave * 2048;
        ng_glds16_so(base, q_boff[h][0], dst);
        ng_glds16_so(base, q_boff[h][1], dst + 1024);
    };

    // ---------------- compute state
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fl = 2 * (q | ((g & 1) << 2));
    int a_ad[4], b_ad[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a_ad[mt] = (8 * g + q) * 256 + ((((wr * 4 + mt) * 2) | (pp >> 1)) ^ fl) * 16 + 8 * (pp & 1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b_ad[nt] = (8 * g + q) * 256 + ((((wc * 2 + nt) * 2) | (pp >> 1)) ^ fl) * 16 + 8 * (pp & 1);
    // The transposing reads are INLINE ASM: for the ds_read_tr builtin hipcc (ROCm 7.2) waits vmcnt(0) before every read while an
    // LDS-DMA is in flight (it cannot tell the read from the DMA's destination), which drains the ring every phase.  The compiler then
    // neither counts these reads nor waits for them: mma() waits lgkmcnt(0) itself, fences the scheduler behind the wait (MFMAs are
    // register-only and would otherwise be hoisted over it), and only THEN joins the two 8-byte halves of an operand -- any register
    // copy the join needs happens after the data has arrived.  Offsets are 16-bit: buffer 1 reads through a second address (+ 64 KB).
    const unsigned lds0 = unsigned(size_t((NG_LDS char*)lds));
    constexpr int NW = (S * T256_HALF + 65535) / 65536;       // 64 KB windows of the ring (the instruction's offset field has 16 bits)
    unsigned a_adw[4][NW], b_adw[2][NW];
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a_adw[mt][w_] = lds0 + unsigned(a_ad[mt]) + 65536u * w_;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b_adw[nt][w_] = lds0 + unsigned(b_ad[nt]) + 65536u * w_;
    }
    s16x4 Ar[4][2][2], B0r[2][2][2], B1r[2][2][2];           // [tile][k-step][8-byte half]
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#define T256_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
    // (slot is a literal at every call site: the window and the offset fold)
#define T256_TR4(R, ad, slot)                                                                                     \
    switch ((slot) & 3) {                                                                                          \
        case 0: T256_TR(R[0][0], ad, 0); T256_TR(R[0][1], ad, 1024); T256_TR(R[1][0], ad, 8192); T256_TR(R[1][1], ad, 8192 + 1024); break;                               \
        case 1: T256_TR(R[0][0], ad, 16384); T256_TR(R[0][1], ad, 16384 + 1024); T256_TR(R[1][0], ad, 16384 + 8192); T256_TR(R[1][1], ad, 16384 + 8192 + 1024); break;   \
        case 2: T256_TR(R[0][0], ad, 32768); T256_TR(R[0][1], ad, 32768 + 1024); T256_TR(R[1][0], ad, 32768 + 8192); T256_TR(R[1][1], ad, 32768 + 8192 + 1024); break;   \
        default: T256_TR(R[0][0], ad, 49152); T256_TR(R[0][1], ad, 49152 + 1024); T256_TR(R[1][0], ad, 49152 + 8192); T256_TR(R[1][1], ad, 49152 + 8192 + 1024); break;  \
    }
    auto readA = [&](const int slot) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const unsigned ad = a_adw[mt][slot >> 2];
            T256_TR4(Ar[mt], ad, slot)
        }
    };
    auto readB = [&](const int which, const int slot) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const unsigned ad = b_adw[nt][slot >> 2];
            if (which == 0) { T256_TR4(B0r[nt], ad, slot) } else { T256_TR4(B1r[nt], ad, slot) }
        }
    };
#undef T256_TR4
#undef T256_TR
    auto join = [](const s16x4 lo, const s16x4 hi) -> bf16x8 {
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mma = [&](f32x4 (&c)[4][2], const s16x4 (&Bj)[2][2][2]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    c[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(Ar[mt][s_][0], Ar[mt][s_][1]), join(Bj[nt][s_][0], Bj[nt][s_][1]), c[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    T256_STAMP(1)
    if (nk > 0)
        t256_kloop<S>(nk, wr, issueA, issueB, advance, readA, readB,
                      [&](const int i, const int j) { if (j == 0) mma(acc[i][0], B0r); else mma(acc[i][1], B1r); });

    T256_STAMP(2)
    T256_STAMP(3)
    // ---------------- the partial tile to its slab: rows n, 256-byte segments of J (through the wave's own 16 KB of LDS, as conv_tile256)
    float* const stg = reinterpret_cast<float*>(lds + wave * 16384);
    float* const slab = p.slabs + size_t(split) * p.N * p.K;
    const int chunk = lane & 15, lrow = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(mt * 16 + lrow * 4 + r) * 64 + j * 32 + nt * 16 + (lane & 15)] = acc[i][j][mt][nt][r];
        float* dst = slab + size_t(n0 + i * 128 + wr * 64 + lrow) * p.K + j0 + wc * 64 + chunk * 4;
#pragma unroll 4
        for (int pass = 0; pass < 16; ++pass)
            *reinterpret_cast<f32x4*>(dst + size_t(pass * 4) * p.K) = *reinterpret_cast<const f32x4*>(stg + (pass * 4 + lrow) * 64 + chunk * 4);
    }
#ifdef NG_DIAG256
    T256_STAMP(4)
    if (lane == 0 && p.dbg != nullptr) {
        unsigned long long* o = p.dbg + (size_t(unit) * 8 + wave) * 7;
        for (int q_ = 0; q_ < 5; ++q_) o[q_] = ng_t[q_];
        o[5] = ng_r0;
        o[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// whether wgrad_tile256 covers a problem (host)
inline bool wgrad_tile256_ok(const WgradParams& p) {
    if (!(p.prec == 1 && p.pq_bf16 && p.nplanes == 1)) return false;
    if (p.N % 256 != 0 || p.K % 256 != 0 || p.run % 8 != 0) return false;
    if (!((p.OW % 64 == 0) || (p.OW <= 64 && 64 % p.OW == 0)) || p.OHW % 64 != 0 || p.rows_per_split % 64 != 0) return false;
    for (int t = 0; t < p.ntaps; ++t)
        if (p.tap_off[t] < 0) return false;
    return p.fast32_bytes != 0;
}

// whether the 256-wide tile covers a problem (host): both operands stored as bf16, whole 64-channel slices, whole 256-column tiles,
// 32-bit offsets -- and enough tiles.  One workgroup per CU: a launch takes ceil(tiles / CUs) rounds whatever the last round holds, and a
// round is worth about 0.62 of what the 128-row tile (two workgroups per CU, 0.33 of the peak against 0.50) needs for a round's worth of
// work; so alone the tile is chosen when its rounds are at least 62 % full (128 tiles on 256 CUs: 55 us against 47; 273 tiles: 120 against
// 118 -- the fused launch fills such a tail with weight-gradient units and only asks for `pair` = at least half a round).
inline bool conv_tile256_ok(const ConvParams& p, const bool pair = false, const int cus = 256) {
    // exact fp32 on this tile only on request (NIRGAN_CONV_TILE256): measured within 1 % of the 128-row tile on long K loops (138.3 against
    // 136.7 TFLOP/s on the 3x3 256 -> 256 layer, profiles/r04_tile256_fp32_parts.txt) -- at 64 cycles per MFMA the load segment was
    // never the limit -- so the parity path keeps the kernels its fixtures were measured with
    const bool b16 = p.prec == 1 && p.in_bf16 && p.w_bf16, f32 = p.prec == 0 && !p.out16 && !p.f_y16 && p.algo == NIRGAN_CONV_TILE256;
    if (!((b16 || f32) && p.off32 && p.ksplit == 1)) return false;
    if (p.run % (b16 ? 64 : 32) != 0 || p.N % 256 != 0) return false;
    // a bf16 output leaves eight channels per lane (16-byte stores, 32-byte loads of bias / mean / rstd, 16-byte loads of a bf16 y): every
    // pixel's channel group must then be 16-byte aligned, as the 128-row tile requires for its eight-channel form
    if (p.out16 && ((p.out_cs | p.out_org | p.out_row | p.out_img) & 7) != 0) return false;
    if (p.f_y16 && ((p.f_img | p.f_row | p.f_org) & 7) != 0) return false;
    const long long tiles = (long long)((p.M + 255) >> 8) * (p.N >> 8);
    if (pair) return tiles >= cus / 2;
    const long long rounds = (tiles + cus - 1) / cus;
    return tiles * 100 >= rounds * cus * 62;
}

}  // namespace ng
