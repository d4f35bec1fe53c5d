// fp32-EQUIVALENT contraction on the bf16 matrix pipe (round 5; descriptor precision 3).
//
// On gfx950 an fp32 MFMA runs at the vector rate (157 TFLOP/s, 1/16 of bf16).  Here every fp32 operand is split into THREE bf16 terms
// x = h + m + l (each the round-to-nearest bf16 of what the previous ones left: 24+ significant bits, the split is exact) and a
// product a * b is contracted as six bf16 products with fp32 accumulation,
//     al bh + ah bl + am bm + am bh + ah bm + ah bh,
// i.e. every cross term down to 2^-16 of the leading one; the three dropped terms (am bl, al bm, al bl) are <= 2^-24 |a b| -- below the
// rounding of the fp32 product itself.  Ceiling 2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-equivalent work against 157.
// (model/networks.py:349,360-363,559-574 through train.py:29: the reference's arithmetic is fp32.)
//
// Tile 256 (M) x BN (N) x 32 (K), ONE workgroup of eight waves per CU (4 x 2, wave tile 64 x BN/2), v_mfma_f32_16x16x32_bf16.
//   A (activations / gradients, fp32 in HBM as every other kernel reads them -- no twin buffers, no producer changes): each thread
//     fetches 2 x 8 consecutive k of its two rows into registers one K-tile ahead (global_load_dwordx4), splits them (11 VALU per two
//     elements: v_cvt_pk_bf16_f32, shift / mask, subtract -- in the shadow of the MFMAs: an MFMA holds the vector issue port for 8 of
//     its 16 cycles) and writes the three bf16 images with ds_write_b128: every element is converted ONCE per workgroup, the fragment
//     reads are plain bf16 (no conversion in front of the MFMAs, one read feeds six of them).
//   B (packed weights): split once per optimizer step by nirgan_split3 into three bf16 planes [3][N][K]; staged by LDS-DMA
//     (one 1 KB piece per wave and term).
//   LDS images: per term [rows][32 k] in 64-byte rows, 16-byte chunk c of row r stored at chunk c ^ g((r >> 2) & 3), g = {0, 2, 3, 1}:
//     conflict-free for ds_read_b128's lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) with lane = 16 * chunk + row
//     (MI355X_MICROARCH.md, LDS table) and for the 8-lane groups of ds_write_b128 (two whole rows per group).
//   Two stages (2 x (48 + 3 BN / 16) KB = 144 KB at BN = 128), ONE barrier per K-tile: 96 MFMAs (1 536 cycles) per wave between barriers.
#pragma once
#include "igemm_tiles.h"

namespace ng {

constexpr int X3_A_TERM = 256 * 64;            // bytes of one term image of A (256 rows x 32 k bf16)
constexpr int X3_A_BYTES = 3 * X3_A_TERM;

__device__ __forceinline__ int x3_key(const int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__device__ __forceinline__ f32x4 ng_gld16_so(const char* base, unsigned off) {
    asm volatile("" : "+v"(off));
    return *reinterpret_cast<const NG_GLOBAL f32x4*>((const NG_GLOBAL char*)base + off);
}

// x = h + m + l, each term the RNE bf16 of the remainder (exact: the remainders are representable in fp32).  Per PAIR of elements:
// one v_cvt_pk_bf16_f32 gives both bf16 terms in one dword, a shift and a mask widen them back, two subtractions leave the remainders
// (5 VALU per level, 11 in all; hipcc's own bf16 -> f32 widening of a vector re-converts every element: 15)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned x3_pk(const float a, const float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (one v_sub_f32 per element: left to itself hipcc pairs the subtractions into v_pk_add_f32, which beside MFMAs costs more than the two
// scalar forms it replaces -- MI355X_MICROARCH.md, 'price of one filler beside MFMAs')
__device__ __forceinline__ float x3_sub(const float a, const float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void x3_split8(const f32x4 lo, const f32x4 hi, bf16x8& H, bf16x8& M, bf16x8& L) {
    const f32x8 x = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    u32x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        h[i] = x3_pk(x0, x1);
        const float r0 = x3_sub(x0, __builtin_bit_cast(float, h[i] << 16)), r1 = x3_sub(x1, __builtin_bit_cast(float, h[i] & 0xffff0000u));
        m[i] = x3_pk(r0, r1);
        const float s0 = x3_sub(r0, __builtin_bit_cast(float, m[i] << 16)), s1 = x3_sub(r1, __builtin_bit_cast(float, m[i] & 0xffff0000u));
        l[i] = x3_pk(s0, s1);
    }
    H = __builtin_bit_cast(bf16x8, h);
    M = __builtin_bit_cast(bf16x8, m);
    L = __builtin_bit_cast(bf16x8, l);
}

// Work of one persistent launch: up to four problems of one tile width (the sub-pixel phases of a stride-2 layer) or `nplanes`
// problems of one geometry (the plane GEMMs of a Winograd layer: plane i reads in + i in_plane, the weight planes + i w3_pstride and
// writes out + i out_plane).  Item i of the launch = tile i - first[k] of problem k.
struct X3Work {
    ConvParams p[4];
    int first[5];              // first item of each problem; first[n] = all items
    int n;
    int nplanes;               // > 1: p[0] only; item = plane * tiles + tile
    long long in_plane, w3_pstride, out_plane;
    int spread;                // problems of UNEQUAL K in one launch (sub-pixel phases with 1 / 2 / 2 / 4 taps): tile t of problem k belongs to
    int start[4];              // the workgroup at walk position (start[k] + t) mod gridDim.x, so every workgroup gets its share of every
                               // problem (host: ng_launch_conv_x3 places the problems longest first where the load is lowest).  Without
                               // it the items are walked problem by problem, XCD-contiguous: the XCDs that got the 4-tap phase ran 295 us
                               // of a 316 us launch whose balanced time is 177 (ConvTranspose2d(256, 128, 3, s2) at bs 16)
};

// PERSISTENT workgroups, one per CU: a workgroup walks items b, b + G, ... (XCD-contiguous through ng_xcd_remap).  What one tile per
// workgroup left exposed with a single workgroup per CU -- 7-10 us of prologue (index arithmetic, the first fetch's latency) and
// epilogue per tile, as much as the K loop of a sub-pixel phase with 4-16 K-tiles -- overlaps here: behind the K loop's last barrier
// the NEXT item's first K-tile is fetched (registers + LDS-DMA into stage 0), then this item's accumulators are stored, then the
// fetched rows are converted.  The epilogue's transpose goes through stage 1 only (in two halves): stage 0 is the next item's.
// (Measured and dropped: the MFMA operands swapped so that a lane's four accumulator registers are four consecutive channels of one
// pixel and leave as one 16-byte store without LDS -- adjacent lanes then hold different pixels, the stores do not coalesce: +8 us per tile.)
template <int BN>
__device__ __forceinline__ void conv_x3_persist(const NG_CONST X3Work* const wp, char* sA0, char* sA1, char* sB0, char* sB1, char* sRed) {
    static_assert(BN == 128 || BN == 64, "256 x 128 or 256 x 64 block tiles");
    constexpr int B_TERM = BN * 64;            // bytes of one term image of B
    constexpr int NT = BN / 32;                // 16-column MFMA tiles per wave (wave tile 64 x BN/2)
    constexpr int CW = BN / 2;                 // columns per wave
    constexpr int BPT = BN / 16;               // 1 KB LDS-DMA pieces per term image of B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int total = wp->first[wp->n];
    const int G = gridDim.x;

    // ---------------- constants of the launch
    int a_wr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = j * 128 + (tid >> 2), c = tid & 3;
        a_wr[j] = row * 64 + ((c ^ x3_key(row)) << 4);
    }
    const int b_idx = wave % BPT;               // piece of a term image this wave stages (the same for every term it serves)
    const int b_row = b_idx * 16 + (lane >> 2), b_c = (lane & 3) ^ x3_key(b_row);
    const int swz = ((lane >> 4) ^ x3_key(lane & 15)) << 4;
    const int a_rd = (wr * 64 + (lane & 15)) * 64 + swz;
    const int b_rd = (wc * CW + (lane & 15)) * 64 + swz;

    // ---------------- per-item state: `L` the item whose K-tiles are being fetched, `E` the item whose accumulators are being stored
    struct Item {
        const NG_CONST ConvParams* p;
        const char* in8; const char* w8; float* out;
        int m0, n0, nk;
        int ntaps, run;            // (copies: a read through `p` behind a barrier or an asm memory clobber is a scalar load again)
        long long w3_plane;
    };
    auto locate = [&](const int item, Item& t) {
        int k = 0, id = 0;
        if (wp->spread) {
            // this workgroup's m-th item (item = blockIdx.x + m G): walk the problems, count the tiles of each that sit at this position.
            // Positions are XCD-contiguous (ng_xcd_remap of the workgroup id): neighbouring row tiles, and ALL phases of one row tile,
            // are fetched through one L2
            int m = (item - int(blockIdx.x)) / G;
            const int pos = ng_xcd_remap(int(blockIdx.x), G);
            k = -1;
            for (int q = 0; q < wp->n; ++q) {
                const int Tq = wp->first[q + 1] - wp->first[q];
                int t0 = pos - wp->start[q];
                t0 += t0 < 0 ? G : 0;
                const int cnt = t0 < Tq ? (Tq - 1 - t0) / G + 1 : 0;
                if (k < 0) {
                    if (m < cnt) { k = q; id = t0 + m * G; }
                    else m -= cnt;
                }
            }
            if (k < 0) { t.nk = -1; return; }                             // no such item: this workgroup is done
        } else {
            const int id0 = ng_xcd_remap(item, total);
            if (id0 >= wp->first[1]) k = 1;
            if (id0 >= wp->first[2]) k = 2;
            if (id0 >= wp->first[3]) k = 3;
            id = id0 - wp->first[k];
        }
        const NG_CONST ConvParams* p = &wp->p[k];
        int plane = 0;
        if (wp->nplanes > 1) {
            const int per = total / wp->nplanes;
            plane = id / per;
            id -= plane * per;
        }
        const int ntn = p->N / BN;
        t.p = p;
        t.n0 = (id % ntn) * BN;
        t.m0 = (id / ntn) * 256;
        t.ntaps = p->ntaps;
        t.run = p->run;
        t.w3_plane = p->w3_plane;
        t.nk = t.ntaps * (t.run >> 5);
        t.in8 = reinterpret_cast<const char*>(p->in + (long long)plane * wp->in_plane);
        t.w8 = reinterpret_cast<const char*>(p->w3 + (long long)plane * wp->w3_pstride);
        t.out = p->out + (long long)plane * wp->out_plane;
    };
    Item L, E;
    unsigned a_goff[2], b_goff;
    int tapv, ct = 0, cc = 0;                  // K-tiles in slice-major order: 32-channel slice cc of the run, all taps
    auto begin = [&]() {                       // loader state of item L
        const NG_CONST ConvParams& p = *L.p;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = L.m0 + j * 128 + (tid >> 2);
            m = m < p.M ? m : p.M - 1;
            const int b = m / p.OHW, r = m - b * p.OHW;
            const int oh = r / p.OW, ow = r - oh * p.OW;
            a_goff[j] = unsigned(b * p.in_img + oh * p.in_stride * p.in_row + ow * p.in_stride * p.in_cs + p.in_org + (tid & 3) * 8) * 4u;
        }
        b_goff = unsigned((L.n0 + b_row) * p.K + b_c * 8) * 2u;
        tapv = p.tap_off[lane & (NIRGAN_MAX_TAPS - 1)];
        ct = 0;
        cc = 0;
    };
    auto advance = [&]() {
        ++ct;
        if (ct == L.ntaps) { ct = 0; cc += 32; }
    };
    f32x4 ra[4];
    auto loadA = [&]() {
        const int toff = __builtin_amdgcn_readlane(tapv, ct);
        const char* base = ng_uniform_ptr(L.in8 + (long long)(toff + cc) * 4);
        ra[0] = ng_gld16_so(base, a_goff[0]);
        ra[1] = ng_gld16_so(base, a_goff[0] + 16u);
        ra[2] = ng_gld16_so(base, a_goff[1]);
        ra[3] = ng_gld16_so(base, a_goff[1] + 16u);
    };
    auto issueB = [&](char* sB) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 8 * i;     // (uniform) BN = 128: term i, piece `wave` of it
            if (BN == 128 || piece < 3 * BPT) {
                const int term = piece / BPT;
                const char* base = ng_uniform_ptr(L.w8 + ((long long)term * L.w3_plane + ct * L.run + cc) * 2);
                ng_glds16_so(base, b_goff, sB + term * B_TERM + b_idx * 1024);
            }
        }
    };
    auto commitA = [&](char* sA, const int j) {
        bf16x8 H, M, Lo;
#ifdef NG_X3_DIAG              // diagnostic build only: 0x400 = no conversion (raw bits stored), 0x800 = no LDS stores of the converted rows
        if (L.p->algo & 0x800) return;
        if (L.p->algo & 0x400) {
            H = __builtin_bit_cast(bf16x8, ra[2 * j]); M = __builtin_bit_cast(bf16x8, ra[2 * j + 1]); Lo = H;
        } else
#endif
        x3_split8(ra[2 * j], ra[2 * j + 1], H, M, Lo);
        *reinterpret_cast<bf16x8*>(sA + a_wr[j]) = H;
        *reinterpret_cast<bf16x8*>(sA + X3_A_TERM + a_wr[j]) = M;
        *reinterpret_cast<bf16x8*>(sA + 2 * X3_A_TERM + a_wr[j]) = Lo;
    };

    // ---------------- compute: acc[mt][nt][r] = out[row mt * 16 + 4 (lane >> 4) + r][column nt * 16 + (lane & 15)] of the wave tile
    f32x4 acc[4][NT];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // one K-tile.  The fragment reads are INLINE ASM with counted lgkmcnt waits, ordered by hand against the MFMAs:
    //   * written as plain loads, hipcc puts s_waitcnt vmcnt(0) in front of the first fragment read of a step once the next stage's LDS-DMA
    //     has been issued (it cannot tell the read from the DMA's destination): the whole fetch latency at the head of every step;
    //   * left to the scheduler, all 18 reads of the first half are issued up front and the first MFMA waits for most of them while all
    //     eight waves read at once (~600 cycles of the LDS pipe with the matrix pipe idle).
    // Here the first MFMA group needs six reads (A row tile 0, B column tile 0); the next column tile's terms are read under the current
    // group's MFMAs, the next row tile's under the last group's; no scalar-memory operation is issued inside the loop (lgkmcnt counts
    // LDS operations in order only while none is outstanding).  `mid()` = the conversion + LDS stores of the NEXT K-tile's A rows: behind
    // the first half (their fetch has landed by then), interleaved by the scheduler with the second half's MFMAs, whose operands are all
    // in registers (two VALU per MFMA gap: an MFMA holds the vector issue port for 8 of its 16 cycles).
    const unsigned a_ad0 = unsigned(size_t((NG_LDS char*)sA0)) + unsigned(a_rd), a_ad1 = unsigned(size_t((NG_LDS char*)sA1)) + unsigned(a_rd);
    const unsigned b_ad0 = unsigned(size_t((NG_LDS char*)sB0)) + unsigned(b_rd), b_ad1 = unsigned(size_t((NG_LDS char*)sB1)) + unsigned(b_rd);
#define X3_DSR(dst, ad, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(off) : "memory")
#define X3_RA(a, ad, MT) { X3_DSR(a[0], ad, (MT) * 1024); X3_DSR(a[1], ad, X3_A_TERM + (MT) * 1024); X3_DSR(a[2], ad, 2 * X3_A_TERM + (MT) * 1024); }
#define X3_RB(ad, NTI) { X3_DSR(B[NTI][0], ad, (NTI) * 1024); X3_DSR(B[NTI][1], ad, B_TERM + (NTI) * 1024); X3_DSR(B[NTI][2], ad, 2 * B_TERM + (NTI) * 1024); }
#define X3_WAIT(n) { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    auto compute = [&](const unsigned aad, const unsigned bad, auto more_tag, auto&& pre, auto&& mid) {
        constexpr bool MORE = decltype(more_tag)::value;
        bf16x8 B[NT][3], A[2][3];
        auto mma = [&](const int mt, const bf16x8 (&a)[3], const int nt) {
            f32x4 c = acc[mt][nt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], B[nt][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], B[nt][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], B[nt][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], B[nt][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], B[nt][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], B[nt][0], c, 0, 0, 0);
            acc[mt][nt] = c;
        };
        __builtin_amdgcn_sched_barrier(0);
        X3_RA(A[0], aad, 0)
        X3_RB(bad, 0)
        pre();                                    // the next K-tile's fetch (registers + LDS-DMA): issued behind the first reads
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NT == 4) {
            X3_RB(bad, 1) X3_WAIT(3) mma(0, A[0], 0); __builtin_amdgcn_sched_barrier(0);
            X3_RB(bad, 2) X3_WAIT(3) mma(0, A[0], 1); __builtin_amdgcn_sched_barrier(0);
            X3_RB(bad, 3) X3_WAIT(3) mma(0, A[0], 2); __builtin_amdgcn_sched_barrier(0);
            X3_RA(A[1], aad, 1) X3_WAIT(3) mma(0, A[0], 3); __builtin_amdgcn_sched_barrier(0);
        } else {
            X3_RB(bad, 1) X3_WAIT(3) mma(0, A[0], 0); __builtin_amdgcn_sched_barrier(0);
            X3_RA(A[1], aad, 1) X3_WAIT(3) mma(0, A[0], 1); __builtin_amdgcn_sched_barrier(0);
        }
        X3_RA(A[0], aad, 2) X3_WAIT(3)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mma(1, A[1], nt);
        __builtin_amdgcn_sched_barrier(0);
        // second half: row tile 3's terms are read into the registers row tile 1 has just left, under row tile 2's MFMAs and the first
        // half of the conversion; the wait in front of row tile 3 counts the LDS stores issued behind that read (three per converted row
        // set) instead of draining them.  Row tile 2's own terms are waited for HERE (all but the three reads just issued): they were
        // issued a whole MFMA group ago and have long landed on an idle LDS pipe, but nothing else orders mma(2) behind them -- without
        // this wait a workgroup whose LDS pipe is contended (two launches on two streams) multiplied a stale low term once in ~10^5 tiles
        X3_RA(A[1], aad, 3) X3_WAIT(3)
        mid(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mma(2, A[0], nt);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) { X3_WAIT(3) } else { X3_WAIT(0) }
        mid(1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mma(3, A[1], nt);
    };
#undef X3_WAIT
#undef X3_RB
#undef X3_RA
#undef X3_DSR
    // (two distinct stage objects, unrolled by two: the compiler sees that the stores and the LDS-DMA of stage s + 1 do not alias the
    // fragment reads of stage s)
    // `behind_stores`: the first step behind a full tile's epilogue.  vmcnt counts in issue order and this item's first LDS-DMA pieces
    // were issued BEFORE that epilogue's ESTORES output stores, so "all but the ESTORES youngest" covers them and leaves the stores in
    // flight -- with vmcnt(0) every workgroup would sit out the drain of the 32 MB all 256 of them have just written (measured: +8 us
    // per tile against one tile per workgroup, where the next workgroup's loop runs under the previous one's stores).
    constexpr int ESTORES = 2 * (32 / (64 / (CW / 4)));       // store instructions of a full tile's epilogue, per lane
    auto step = [&](const int cur, char* nA, char* nB, const bool more, const bool behind_stores = false) {
        if (behind_stores) {
            if constexpr (ESTORES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const unsigned aad = cur == 0 ? a_ad0 : a_ad1, bad = cur == 0 ? b_ad0 : b_ad1;
        if (more) {
#ifdef NG_X3_DIAG              // 0x1000 = no fetch of the next K-tile (the loop runs on stale LDS images); 0x800 changes the store count: drain
            if (L.p->algo & 0x1800) { compute(aad, bad, std::false_type{}, [&]() { if (!(L.p->algo & 0x1000)) { issueB(nB); loadA(); } advance(); }, [&](const int j) { commitA(nA, j); }); return; }
#endif
            compute(aad, bad, std::true_type{}, [&]() { issueB(nB); loadA(); advance(); }, [&](const int j) { commitA(nA, j); });
        } else {
            compute(aad, bad, std::false_type{}, []() {}, [](const int) {});
        }
    };

    // ---------------- epilogue of item E.  The accumulators go through the SECOND stage (idle until the next item's second K-tile lands:
    // the next item's first fetch, into stage 0, is in flight meanwhile) in two halves of 32 rows, each wave through its own 32 x CW
    // floats, and leave as whole row segments, 16 bytes per lane.
    auto epilogue = [&]() -> bool {
        const NG_CONST ConvParams& p = *E.p;
#ifdef NG_X3_DIAG              // diagnostic build only (scripts/diag/x3_parts.sh): what do the epilogue / its stores cost a plane GEMM?
        if (p.algo & 0x200) return E.m0 + 256 <= p.M;
#endif
        // (copies: the stores below make the compiler read every field again through `p`)
        const int pM = p.M, pN = p.N, OHW = p.OHW, OW = p.OW, OH = p.OHW / p.OW, out_img = p.out_img, out_row = p.out_row * p.out_stride,
                  out_px = p.out_cs * p.out_stride, out_org = p.out_org, f_img = p.f_img, f_row = p.f_row * p.out_stride, f_px = p.ch * p.out_stride,
                  f_org = p.f_org, pC = p.ch, span = p.N / p.ch;
        // (pC < pN -- out_span = 2: column n is channel n - pC of the row's SECOND pixel from pC on; the per-channel records of the
        // statistics / the fused sums then come in pairs per chunk, pixel parity 0 first)
        const float* const f_y = p.f_y;
        const int mbase = E.m0 + wr * 64;
        const bool full = E.m0 + 256 <= pM;
        // partial sums for the instance norm that follows (nirgan_conv_desc.stats_ws): the wave's 64 rows leave, per column,
        // {k = the chunk's first row, sum (v - k), sum (v - k)^2, 64} -- the contract of conv_tile / conv_tile256
        if (p.stats != nullptr && mbase < pM) {                     // (host: OH*OW % 128 == 0: a 64-row chunk is whole and inside one sample)
            const int b = mbase / OHW;
            float* sp = p.stats + (size_t(b) * p.stats_cps + p.stats_chunk0 + ((mbase - b * OHW) >> 6) * span) * 4 * pC;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float k0 = __shfl(acc[0][nt][0], lane & 15, 64);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[mt][nt][r] - k0;
                        s1 += v;
                        s2 += v * v;
                    }
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                int col = E.n0 + wc * CW + nt * 16 + (lane & 15);
                if (col >= pC) col += 3 * pC;                   // the second pixel's record follows the first one's four rows
                if (lane < 16) {
                    sp[col] = k0;
                    sp[pC + col] = s1;
                    sp[2 * pC + col] = s2;
                    sp[3 * pC + col] = 64.f;
                }
            }
        }
        constexpr int LPR = CW / 4, RPP = 64 / LPR;         // lanes per row (4 channels each), rows per pass
        // staging rows are CW floats = a whole number of bank rounds: the four 16-lane groups of an accumulator store (rows 4 apart,
        // same columns) would hit the same banks -- 16-column block b of row r is stored at block b ^ ((r >> 2) & SWM)
        constexpr int SWM = CW / 16 - 1;
        const int sw_wr = ((lane >> 4) & SWM) << 4;
        float* const stg = reinterpret_cast<float*>(wave < 6 ? sA1 + wave * 8192 : sB1 + (wave - 6) * 8192);
        const bool fused = f_y != nullptr;
        const float fneg = p.f_act == NIRGAN_ACT_RELU ? 0.f : (p.f_act == NIRGAN_ACT_LRELU ? p.f_slope : 1.f);
        const int chunk = lane % LPR, lrow = lane / LPR;
        const int n = E.n0 + wc * CW + chunk * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
        int m = mbase + lrow;
        const int mc = m < pM ? m : pM - 1;
        int b = mc / OHW;
        const int r0 = mc - b * OHW;
        int oh = r0 / OW, ow = r0 - oh * OW;
        const int fb = (mbase < pM ? mbase : pM - 1) / OHW;           // (fused: one sample per 128-row chunk, host: OH*OW % 128 == 0)
        f32x4 fm = {0.f, 0.f, 0.f, 0.f}, fr = fm, s1 = fm, s2 = fm;
        const int nq = n >= pC ? 1 : 0, nc = n - nq * pC;           // pixel of the row and channel of this lane's four columns
        if (fused) {
            fm = *reinterpret_cast<const f32x4*>(p.f_mean + size_t(fb) * pC + nc);
            fr = *reinterpret_cast<const f32x4*>(p.f_rstd + size_t(fb) * pC + nc);
        }
        // (FULL: a tile whose 256 rows all exist issues every store instruction, unrolled and without a predicate -- step() counts them,
        // and so can the compiler when it waits for the next item's fetched rows in front of their conversion)
        auto halves = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int P = 32 / RPP;                       // passes (store instructions per lane) of a half
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stg[(q * 16 + (lane >> 4) * 4 + r) * CW + ((nt * 16 + (lane & 15)) ^ sw_wr)] = acc[h * 2 + q][nt][r];
                // the half's y values (fused first pass) are fetched BEFORE its stores: vmcnt counts loads and stores in one order, so a
                // load issued behind a store is not "done" before that store is acknowledged -- a y load per pass behind the previous
                // pass's store sat out a store round trip sixteen times per tile
                int ooff[P];
                bool ok[P];
                f32x4 yv[P];
#pragma unroll
                for (int pass = 0; pass < P; ++pass) {
                    ok[pass] = FULL || m < pM;
                    ooff[pass] = b * out_img + oh * out_row + ow * out_px + out_org + n;
                    if (fused && ok[pass]) yv[pass] = *reinterpret_cast<const f32x4*>(f_y + (size_t(b) * f_img + size_t(oh) * f_row + size_t(ow) * f_px + f_org + n));
                    m += RPP;
                    ow += RPP;
                    while (ow >= OW) { ow -= OW; ++oh; }
                    while (oh >= OH) { oh -= OH; ++b; }
                }
#pragma unroll
                for (int pass = 0; pass < P; ++pass) {
                    if (ok[pass]) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(stg + (pass * RPP + lrow) * CW + ((chunk * 4) ^ ((((pass * RPP + lrow) >> 2) & SWM) << 4)));
                        v += bv;
#ifdef NG_X3_DIAG
                        if (!(p.algo & 0x100))
#endif
#ifdef NG_X3_NT_STORES         // (A/B switch, scripts/diag/ab_build.sh: non-temporal stores -- the consumer's reads then miss: -0.84 % on the step)
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(E.out + ooff[pass]));
#else
                        *reinterpret_cast<f32x4*>(E.out + ooff[pass]) = v;
#endif
                        if (fused) {
                            const f32x4 z = (yv[pass] - fm) * fr;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float gz = z[q] > 0.f ? v[q] : v[q] * fneg;
                                s1[q] += gz;
                                s2[q] += gz * z[q];
                            }
                        }
                    }
                }
            }
        };
        if (full) halves(std::true_type{}); else halves(std::false_type{});
        if (fused) {
            // first pass of the consumer layer's instance-norm backward: this wave's 64 rows, then the two waves of a 128-row chunk
            // (wr = 2 c, 2 c + 1) join through a small LDS array of their own in a fixed order
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
                    s1[q] += __shfl_xor(s1[q], o, 64);
                    s2[q] += __shfl_xor(s2[q], o, 64);
                }
            f32x4* const red = reinterpret_cast<f32x4*>(sRed);        // 8 waves x LPR x 2 sums
            if (lane < LPR) {
                red[(wave * LPR + chunk) * 2] = s1;
                red[(wave * LPR + chunk) * 2 + 1] = s2;
            }
            __syncthreads();                    // (uniform: `fused` is a constant of the item's problem, the items of a launch agree)
            if ((wr & 1) == 0 && lane < LPR && mbase < pM) {
                const f32x4 t1 = s1 + red[((wave + 2) * LPR + chunk) * 2], t2 = s2 + red[((wave + 2) * LPR + chunk) * 2 + 1];
                float* pp = p.f_part + (size_t(fb) * p.f_cps + p.f_chunk0 + ((mbase - fb * OHW) >> 7) * span + nq) * 2 * pC + nc;
                *reinterpret_cast<f32x4*>(pp) = t1;
                *reinterpret_cast<f32x4*>(pp + pC) = t2;
            }
            __syncthreads();                    // the array is free for the next item
        }
        return full;
    };

    // ---------------- the walk
    int item = blockIdx.x;
    if (item >= total) return;
    locate(item, L);
    if (L.nk < 0) return;                       // (spread walk: no tile of any problem sits at this workgroup's position)
    begin();
    loadA();
    issueB(sB0);
    advance();
    commitA(sA0, 0);
    commitA(sA0, 1);
    bool behind = false;
    while (true) {
        zero();
        const int nk = L.nk;
        int k = 0;
        if (nk >= 2) {
            if (behind) step(0, sA1, sB1, true, true); else step(0, sA1, sB1, true);
            step(1, sA0, sB0, 2 < nk);
            k = 2;
        }
        for (; k + 2 <= nk; k += 2) {
            step(0, sA1, sB1, true);
            step(1, sA0, sB0, k + 2 < nk);
        }
        if (k < nk) step(0, sA1, sB1, false);
        E = L;
        item += G;
        bool more = wp->spread ? true : item < total;      // (spread walk: locate says when this workgroup's items are through)
        __syncthreads();                        // every wave's fragment reads of the last K-tile are done: stage 0 is free
        if (more) {
            locate(item, L);
            more = L.nk >= 0;
        }
        if (more) {
            begin();
            loadA();
            issueB(sB0);
            advance();
        }
        behind = epilogue();
        if (!more) break;
        commitA(sA0, 0);
        commitA(sA0, 1);
    }
}

// one persistent launch over 1..4 problems of one tile width, or over nplanes problems of p[0]'s geometry (defined in igemm_conv.hip)
int ng_launch_conv_x3(const ConvParams* ps, int n, int bn, int nplanes, long long in_plane, long long w3_pstride, long long out_plane,
                      hipStream_t st, const char* what);
int ng_cu_count_conv();

// whether the split tile covers a problem (host): precision 3 with the weight planes present, 32-channel slices, whole 64-column tiles,
// fp32 tensors on both sides, 32-bit offsets, no split-K
inline bool conv_x3_ok(const ConvParams& p) {
    if (!(p.prec == 3 && p.w3 != nullptr && p.off32 && p.ksplit == 1)) return false;
    if (p.run % 32 != 0 || p.N % 64 != 0 || p.in_bf16 || p.w_bf16 || p.out16 || p.f_y16) return false;
    if (((p.out_cs | p.out_org | p.out_row | p.out_img) & 3) != 0) return false;
    return (long long)p.N * p.K * 2 < (1ll << 32);
}
// 128-column tiles unless N has no such tiles, the caller pins the 64-column tile (NIRGAN_CONV_X3_BN64: A/B), or the problem has fewer
// 256 x 128 tiles than three quarters of the CUs (one workgroup per CU: twice as many 64-column tiles fill the chip)
inline int conv_x3_tiles(const ConvParams& p, const int bn) { return ((p.M + 255) >> 8) * (p.N / bn); }
inline int conv_x3_bn(const ConvParams& p, const int cus = 256) {
    if (p.N % 128 != 0 || p.algo == NIRGAN_CONV_X3_BN64) return 64;
    const long long tiles = (long long)((p.M + 255) >> 8) * (p.N / 128);
    return tiles * 4 >= 3ll * cus ? 128 : 64;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The weight gradient in the same arithmetic: slab[split][n][J] = sum over the split's pixels m of P[m][n] * Q[m + tap(J)][c(J)] with
// both operands fp32 in memory (dY and the layer input as every other kernel reads them), split into three bf16 terms on the way into
// LDS.  Unit = TN rows n x 128 columns J x one split; a K-tile = 32 consecutive pixels of ONE image row (host: fast32 -- OW, M and the
// split length multiples of 32 -- so the tile's first pixel walks in scalar registers and every per-lane offset is a constant of the
// unit).  The reduction index (the pixel) is the ROW of both images, the MFMA wants it contiguous per lane: the images are
// [pixel][channel] in rows of 2 TN / 256 bytes and the fragments are read with ds_read_b64_tr_b16 (as wgrad_tile256: a 16-lane group
// reads a 4-row x 16-column block, lane i receives column i; two reads = one 16x16x32 operand); 16-byte chunk c of row r sits at
// chunk c ^ 2 ((r & 3) | ((r >> 3) & 1) << 2).  Eight waves as 4 (n) x 2 (J): wave tile TN/4 x 64.
template <int TN>
__device__ __forceinline__ void wgrad_tile_x3(const WgradParams& p, const int unit, char* sP0, char* sP1, char* sQ0, char* sQ1) {
    static_assert(TN == 128 || TN == 256, "128 or 256 rows n per unit");
    constexpr int PRS = TN * 2;                 // bytes of an image row of P (one pixel, TN channels bf16)
    constexpr int P_TERM = 32 * PRS, Q_TERM = 32 * 256;
    constexpr int MT = TN / 64;                 // 16-row MFMA tiles per wave (wave tile TN/4 x 64)
    constexpr int PT = TN / 128;                // loader tasks per thread for P (8 channels of one pixel each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = p.N / TN, ntj = (p.K + 127) >> 7;
    const int per_plane = ntn * ntj * p.nsplit;
    const int plane = unit / per_plane, u = unit - plane * per_plane;       // (planes: the transform-domain weight gradient of a Winograd layer)
    const int nt_ = u % ntn, rest = u / ntn;
    const int jt = rest % ntj, split = rest / ntj;
    const int n0 = nt_ * TN, j0 = jt * 128;
    const int mstart = split * p.rows_per_split;
    int mend = mstart + p.rows_per_split;
    mend = mend < p.M ? mend : p.M;
    const int nk = mend > mstart ? (mend - mstart + 31) >> 5 : 0;          // (a partly filled last K-tile only in the plane-matrix form)
    const float* const Pp = p.p + (long long)plane * p.p_plane;
    const float* const Qp = p.q + (long long)plane * p.q_plane;

    // ---------------- loader state: thread -> pixel row tid >> 4 of the K-tile, 8 channels / columns from (tid & 15) * 8
    const int lrow = tid >> 4, lch = tid & 15;
    const int lkey = 2 * ((lrow & 3) | (((lrow >> 3) & 1) << 2));
    unsigned p_goff[PT], q_goff;
    int p_wr[PT], q_wr;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        p_goff[i] = unsigned(lrow * p.p_cs + i * 128 + lch * 8) * 4u;
        p_wr[i] = lrow * PRS + (((i * 16 + lch) ^ lkey) << 4);
    }
    {
        int J = j0 + lch * 8;
        J = J < p.K ? J : 0;                    // (columns past K are never stored and never summed: any valid address does)
        const int t = J / p.run;
        q_goff = unsigned(lrow * p.q_stride * p.q_cs + p.tap_off[t] + (J - t * p.run)) * 4u;
        q_wr = lrow * 256 + ((lch ^ lkey) << 4);
    }
    // the K-tile cursor: first pixel (sb, soh, sow) of the tile, advanced by additions in scalar registers
    int sb = __builtin_amdgcn_readfirstlane(mstart / p.OHW);
    int soh = __builtin_amdgcn_readfirstlane((mstart - sb * p.OHW) / p.OW);
    int sow = __builtin_amdgcn_readfirstlane(mstart - sb * p.OHW - soh * p.OW);
    f32x4 rp[PT][2], rq[2];
    int mcur = mstart;
    auto load = [&]() {
        const char* pb = ng_uniform_ptr(reinterpret_cast<const char*>(Pp + (size_t(sb) * p.p_img + size_t(soh) * p.p_row + sow * p.p_cs + p.p_org + n0)));
        const char* qb = ng_uniform_ptr(reinterpret_cast<const char*>(Qp + (size_t(sb) * p.q_img + size_t(soh) * p.q_stride * p.q_row + sow * p.q_stride * p.q_cs + p.q_org)));
        if (mcur + 32 <= mend) {
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                rp[i][0] = ng_gld16_so(pb, p_goff[i]);
                rp[i][1] = ng_gld16_so(pb, p_goff[i] + 16u);
            }
            rq[0] = ng_gld16_so(qb, q_goff);
            rq[1] = ng_gld16_so(qb, q_goff + 16u);
        } else {
            // the split's last, partly filled K-tile (plane-matrix form: one image row of M pixels, M no multiple of 32): rows past the
            // end read the last valid row and contribute zeros
            const int last = mend - 1 - mcur, back = lrow > last ? lrow - last : 0;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const unsigned o = p_goff[i] - unsigned(back * p.p_cs) * 4u;
                rp[i][0] = ng_gld16_so(pb, o);
                rp[i][1] = ng_gld16_so(pb, o + 16u);
                if (back) rp[i][0] = rp[i][1] = z4;
            }
            const unsigned o = q_goff - unsigned(back * p.q_stride * p.q_cs) * 4u;
            rq[0] = ng_gld16_so(qb, o);
            rq[1] = ng_gld16_so(qb, o + 16u);
            if (back) rq[0] = rq[1] = z4;
        }
        mcur += 32;
        sow += 32;
        if (sow >= p.OW) { sow = 0; ++soh; }
        if (soh >= p.OH) { soh = 0; ++sb; }
    };
    auto commit = [&](char* sP, char* sQ) {
        bf16x8 H, M, L;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            x3_split8(rp[i][0], rp[i][1], H, M, L);
            *reinterpret_cast<bf16x8*>(sP + p_wr[i]) = H;
            *reinterpret_cast<bf16x8*>(sP + P_TERM + p_wr[i]) = M;
            *reinterpret_cast<bf16x8*>(sP + 2 * P_TERM + p_wr[i]) = L;
        }
        x3_split8(rq[0], rq[1], H, M, L);
        *reinterpret_cast<bf16x8*>(sQ + q_wr) = H;
        *reinterpret_cast<bf16x8*>(sQ + Q_TERM + q_wr) = M;
        *reinterpret_cast<bf16x8*>(sQ + 2 * Q_TERM + q_wr) = L;
    };

    // ---------------- compute state: lane (g, q, pp) reads 8 bytes of row 8 g + q (+ 4) -- see wgrad_tile256
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fl = 2 * (q | ((g & 1) << 2));
    int a_ad[MT], b_ad[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_ad[mt] = (8 * g + q) * PRS + ((((wr * MT + mt) * 2) | (pp >> 1)) ^ fl) * 16 + 8 * (pp & 1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b_ad[nt] = (8 * g + q) * 256 + ((((wc * 4 + nt) * 2) | (pp >> 1)) ^ fl) * 16 + 8 * (pp & 1);
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto frag = [&](const char* img, const int ad, const int rs) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NG_LDS s16x4*)(img + ad));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((NG_LDS s16x4*)(img + ad + 4 * rs));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    // (reads ordered against the MFMAs as in conv_x3_persist: the first group needs one row tile and one column tile, the next tile's
    // terms are read under the current group's MFMAs; this kernel issues no LDS-DMA, the compiler's own lgkmcnt bookkeeping applies)
    auto compute = [&](const char* sP, const char* sQ, auto&& mid) {
        bf16x8 B[4][3], A[2][3];
        auto readA = [&](const int mt, bf16x8 (&a)[3]) {
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = frag(sP + t * P_TERM, a_ad[mt], PRS);
        };
        auto readB = [&](const int nt) {
#pragma unroll
            for (int t = 0; t < 3; ++t) B[nt][t] = frag(sQ + t * Q_TERM, b_ad[nt], 256);
        };
        auto mma = [&](const int mt, const bf16x8 (&a)[3], const int nt) {
            f32x4 c = acc[mt][nt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], B[nt][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], B[nt][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], B[nt][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], B[nt][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], B[nt][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], B[nt][0], c, 0, 0, 0);
            acc[mt][nt] = c;
        };
        __builtin_amdgcn_sched_barrier(0);
        readA(0, A[0]);
        readB(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt + 1 < 4) readB(nt + 1); else readA(1, A[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma(0, A[0], nt);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MT == 2) {
            mid();                                  // behind the first half of the MFMAs: the next K-tile's operands have arrived
        } else {
            readA(2, A[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) mma(1, A[1], nt);
            __builtin_amdgcn_sched_barrier(0);
            readA(3, A[1]);
            mid();
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) mma(2, A[0], nt);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) mma(MT - 1, A[1], nt);
    };
    auto step = [&](const char* cP, const char* cQ, char* nP, char* nQ, const bool more) {
        __syncthreads();
        if (more) {
            load();
            compute(cP, cQ, [&]() { commit(nP, nQ); });
        } else {
            compute(cP, cQ, []() {});
        }
    };
    if (nk > 0) {
        load();
        commit(sP0, sQ0);
        int k = 0;
        for (; k + 2 <= nk; k += 2) {
            step(sP0, sQ0, sP1, sQ1, true);
            step(sP1, sQ1, sP0, sQ0, k + 2 < nk);
        }
        if (k < nk) step(sP0, sQ0, sP1, sQ1, false);
    }

    // ---------------- the partial tile to its slab: rows n, 256-byte segments of J, through the wave's own 16 / 8 KB of the idle stages
    __syncthreads();
    // (a P stage holds three regions of MT x 4 KB, a Q stage -- 24 KB -- one)
    char* const reg = wave < 3 ? sP0 + wave * (MT * 4096) : (wave < 6 ? sP1 + (wave - 3) * (MT * 4096) : (wave == 6 ? sQ0 : sQ1));
    float* const stg = reinterpret_cast<float*>(reg);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(mt * 16 + (lane >> 4) * 4 + r) * 64 + nt * 16 + (lane & 15)] = acc[mt][nt][r];
    float* const slab = p.slabs + (size_t(plane) * p.nsplit + split) * p.N * p.K;
    const int chunk = lane & 15, srow = lane >> 4;
    const int jj = j0 + wc * 64 + chunk * 4;
    if (jj < p.K) {
        float* dst = slab + size_t(n0 + wr * (TN / 4) + srow) * p.K + jj;
#pragma unroll 4
        for (int pass = 0; pass < MT * 4; ++pass)
            *reinterpret_cast<f32x4*>(dst + size_t(pass * 4) * p.K) = *reinterpret_cast<const f32x4*>(stg + (pass * 4 + srow) * 64 + chunk * 4);
    }
}

// whether wgrad_tile_x3 covers a problem (host)
inline bool wgrad_x3_ok(const WgradParams& p) {
    // the scalar pixel walk: K-tiles of 32 pixels inside one image row (fast32), or the plane-matrix form (one row of M pixels, any M)
    const bool matrix = p.ntaps == 1 && p.tap_off[0] == 0 && p.q_stride == 1 && p.OH == 1 && p.OW == p.M && p.fast32_bytes;
    if (!(p.prec == 3 && !p.pq_bf16 && (p.fast32 || matrix))) return false;
    if (p.nplanes > 1 && !matrix) return false;
    if (p.N % 128 != 0 || p.run % 8 != 0 || p.K % 4 != 0) return false;
    return true;
}
inline int wgrad_x3_tn(const WgradParams& p) { return p.N % 256 == 0 ? 256 : 128; }

}  // namespace ng
