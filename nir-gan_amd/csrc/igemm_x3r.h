// The three-term split tile of igemm_x3.h as ONE WAVE PER SIMD with the activation operand fed from REGISTERS (round 6).
//
// Same arithmetic, same operand order, same accumulation order per output element as conv_x3_persist (the results are bitwise those of
// the eight-wave tile); what changed is where the operands live and how the stream is scheduled:
//   * 256 (M) x BN (N) x 32 (K) per K-tile, FOUR waves as 4 x 1: a wave owns 64 rows and ALL BN columns (wave tile 64 x 128 at BN = 128:
//     128 accumulator registers, the wave has the SIMD's 512 registers to itself).
//   * A (activations, fp32 in HBM) never touches LDS.  The MFMA A operand of v_mfma_f32_16x16x32_bf16 is "lane (row r, chunk c) holds
//     k = 8 c .. 8 c + 7 of row r": exactly what two global_load_dwordx4 of that lane fetch from the row's 32-channel slice.  A row of
//     the block tile is consumed by ONE wave, so that wave fetches it, splits it into the three bf16 terms in registers and multiplies:
//     no ds_write of converted rows, no fragment reads of A, no barrier between conversion and use.  LDS traffic per K-tile and CU:
//     96 KB of B fragment reads + 24 KB of B stores against 192 KB + 72 KB of the eight-wave tile.
//   * B (the three bf16 planes of the packed weights) is staged through REGISTERS into two LDS stages: six global_load_dwordx4 per wave
//     and K-tile, one tile ahead, six ds_write_b128 a tile later (an LDS-DMA piece costs the issuing wave ~180 cycles among MFMAs --
//     measured with in-kernel stamps, scripts/diag/x3r_stamps.* -- and with one wave per SIMD nothing else feeds the matrix pipe
//     meanwhile).  ONE barrier per K-tile in the MIDDLE of the tile (it publishes K-tile j + 1), none at the tile boundaries: the MFMA
//     stream runs from one K-tile into the next without a stop.
//   * mt-outer order with ROLLING operands: a K-tile is four blocks (one 16-row tile mt each) of NT x 6 MFMAs; all NT B fragments of
//     the tile stay in registers (96 at BN = 128) and are replaced one by one behind their last use in block 3; A[mt] of the next
//     K-tile is converted behind block mt (one raw buffer, refilled right behind its conversion).
//   * The hot loop is written as asm REGIONS of twelve MFMAs (X3R_PAIR): the conversion's VALU and the tile's memory instructions sit
//     in the gaps behind the MFMAs, by hand -- with one wave per SIMD an instruction that does not fit a gap is paid in full
//     (scripts/diag/x3r_knockout.sh measures what each part costs).
//   * The loader is a cursor TWO K-tiles ahead of the MFMAs over the workgroup's items as one stream (it is in the next item while the
//     current one is multiplied); the epilogue follows the item's last K-tile: four slices of 16 rows per wave through wave-private
//     staging halves, LDS writes and global stores alternating (full plain tiles: asm; partial tiles, statistics, fused pass: C++).
// Every vector-memory operation of the loop is issued unconditionally and in a fixed order (a loader past the end of its work re-reads
// its last tile), so every s_waitcnt vmcnt(N) below is a constant of the schedule; an item's first K-tile adds the stores its
// epilogue issued behind the fetches it waits for.
#pragma once
#include "igemm_x3.h"

namespace ng {

// diagnostic build only (scripts/diag/x3r_stamps.sh, -DNG_X3R_STAMP): per-wave cycle sums of the segments of a K-tile and of the epilogue
#ifdef NG_X3R_STAMP
__device__ unsigned long long ng_x3r_stamps[1024 * 4 * 16];
#define X3R_STAMP(k) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                       __builtin_amdgcn_sched_barrier(0); st_sum[k] += unsigned(t_) - st_last; st_last = unsigned(t_); }
#else
#define X3R_STAMP(k)
#endif

template <int BN>
struct X3R {
    static constexpr int NT = BN / 16;                 // 16-column MFMA tiles of the wave tile (64 x BN)
    static constexpr int B_TERM = BN * 64;             // bytes of one term image of B ([BN rows][32 k] bf16)
    static constexpr int STAGE = 3 * B_TERM;
    static constexpr int NSTAGE = 2;
    static constexpr int RING = NSTAGE * STAGE;        // 48 KB at BN = 128
    static constexpr int STG = 2 * 16 * BN * 4;        // staging of one wave: two halves of 16 rows x BN floats (a slice each)
    static constexpr int BPT = BN / 16;                // 1 KB pieces (one wave instruction) per term image
    static constexpr int PIECES = 3 * BPT / 4;         // pieces per wave and K-tile (6 / 3)
    static constexpr int LPR = BN / 4;                 // lanes per output row (4 channels each)
    static constexpr int RPP = 64 / LPR;               // rows per store pass
    static constexpr int SP = 16 / RPP;                // store passes per slice (8 / 4)
};

#define X3R_VALU 0x6        // sched_barrier mask: VALU and SALU instructions may cross, MFMAs, memory operations and inline asm may not
#define X3R_GLD(dst, off, base, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(off), "s"(base), "n"(imm) : "memory")
// (the B fragments live in the accumulator half of the register file: ds_read writes AGPRs directly, MFMA reads them as operands --
// the vector half is left to what VALU instructions touch: the raw rows, the split terms, the epilogue)
#define X3R_DSR(dst, ad, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(ad), "n"(off) : "memory")

// One REGION of the K loop as ONE asm block: the twelve MFMAs of a row tile against TWO column tiles -- per accumulator in the order of
// conv_x3_persist::mma (al bh, ah bl, am bm, am bh, ah bm, ah bh) -- interleaved by hand with the conversion of one pair of raw values into
// the three bf16 terms (11 VALU: x3_split8's rule), optionally behind a counted lgkmcnt wait, optionally with the six fragment reads that replace the two column tiles' B terms.
// With ONE wave per SIMD every instruction costs its issue cycles (4 per VALU / SALU / memory instruction, 8 per MFMA; an MFMA occupies
// the pipe for 16): what does not fit the two free slots behind each MFMA adds its full cost to the tile (measured with knock-out builds,
// scripts/diag/x3r_knockout.sh).  Hence regions of
// twelve: hipcc puts an `s_nop 0` between two inline asm statements that share any register operand (a hazard it cannot rule out), and
// consecutive six-MFMA regions shared the conversion's temporaries -- 36 of them per K-tile.  The temporaries here are FIXED registers
// (clobbers), two sets used alternately, so consecutive regions name no common register.
// The accumulators are "+a": they live in the accumulator half of the register file for the whole kernel (left to itself hipcc keeps a
// third of them in VGPRs and moves them back and forth every K-tile).  Hazards: dependent MFMAs on one accumulator back to back are
// interlocked by the hardware; an MFMA's A / B operands are read in its first passes (the VALU writes here go to OTHER registers: the
// next K-tile's terms); what reads an accumulator behind the loop waits out the last MFMA explicitly (X3R_MFMA_DRAIN) -- the compiler
// does not know these are MFMAs.
#define X3R_MC(c, a, b) "v_mfma_f32_16x16x32_bf16 %[" #c "], %[" #a "], %[" #b "], %[" #c "]\n\t"
#define X3R_MZ(c, a, b) "v_mfma_f32_16x16x32_bf16 %[" #c "], %[" #a "], %[" #b "], 0\n\t"
// (diagnostic builds only, scripts/diag/x3r_knockout.sh: NG_X3R_KO is a bit mask of parts of the K loop left out -- wrong results, timing
// of what remains: 1 the conversion's VALU, 2 its v_cvt_pk replaced by v_and, 4 the raw rows' fetches, 8 the B pieces' fetches and
// stores, 16 the fragment reads of block 3, 32 the full epilogue's staging writes and reads, 64 its global stores)
#ifndef NG_X3R_KO
#define NG_X3R_KO 0
#endif
#if NG_X3R_KO & 1
#define X3R_V(s)
#else
#define X3R_V(s) s
#endif
#if NG_X3R_KO & 2
#define X3R_CVT "v_and_b32 "
#else
#define X3R_CVT "v_cvt_pk_bf16_f32 "
#endif
#if NG_X3R_KO & 16
#define X3R_RD(s)
#else
#define X3R_RD(s) s
#endif
// (v_pk_add_f32 for the two subtractions of a step -- 9 instead of 11 VALU per pair -- is SLOWER: + 100 cycles per block of 48 MFMAs,
// stamps, profiles/r06_x3r_knockout.txt: the packed add costs about three plain ones here)
#define X3R_SUB2(R0, R1, X0, X1, T0, T1) X3R_V("v_sub_f32 " R0 ", " X0 ", " T0 "\n\t") X3R_V("v_sub_f32 " R1 ", " X1 ", " T1 "\n\t")
// MA / MB: the first product of each accumulator (X3R_MC, or X3R_MZ in an item's first K-tile: C = 0, nothing zeroes 128 registers per
// item); T0 T1 TP / R0 R1 RP: the temporaries (a register pair each, by halves and whole); RDA / RDB: the reads replacing the first /
// second column tile's fragments (block 3), behind the last MFMA that names them
// G8 .. G12: what else is issued in the gaps behind MFMAs 8 .. 12, which carry no conversion VALU: the fragment reads of block 3 (those
// of the PREVIOUS region's column tiles: its own are still MFMA operands), the B pieces' LDS stores of block 0, the fetches of blocks
// 0-2 -- two instructions per gap at most (an MFMA leaves two issue slots free): as statements of their own between two regions
// every one of them was paid in full
#define X3R_PAIR(MA, MB, T0, T1, TP, R0, R1, RP, G8, G9, G10, G11, G12) \
    MA(c0, al, b0) X3R_V(X3R_CVT "%[h], %[x0], %[x1]\n\t") \
    X3R_MC(c0, ah, b2) X3R_V("v_lshlrev_b32 " T0 ", 16, %[h]\n\t") X3R_V("v_and_b32 " T1 ", 0xffff0000, %[h]\n\t") \
    X3R_MC(c0, am, b1) X3R_SUB2(R0, R1, "%[x0]", "%[x1]", T0, T1) \
    X3R_MC(c0, am, b0) X3R_V(X3R_CVT "%[m], " R0 ", " R1 "\n\t") \
    X3R_MC(c0, ah, b1) X3R_V("v_lshlrev_b32 " T0 ", 16, %[m]\n\t") X3R_V("v_and_b32 " T1 ", 0xffff0000, %[m]\n\t") \
    X3R_MC(c0, ah, b0) X3R_SUB2(R0, R1, R0, R1, T0, T1) \
    MB(c1, al, d0) X3R_V(X3R_CVT "%[l], " R0 ", " R1 "\n\t") \
    X3R_MC(c1, ah, d2) G8 \
    X3R_MC(c1, am, d1) G9 \
    X3R_MC(c1, am, d0) G10 \
    X3R_MC(c1, ah, d1) G11 \
    X3R_MC(c1, ah, d0) G12
#define X3R_TA "v248", "v249", "v[248:249]", "v250", "v251", "v[250:251]"
#define X3R_TB "v252", "v253", "v[252:253]", "v254", "v255", "v[254:255]"
#define X3R_CLOB_A "memory", "v248", "v249", "v250", "v251"
#define X3R_CLOB_B "memory", "v252", "v253", "v254", "v255"
#define X3R_WAITL "s_waitcnt lgkmcnt(%[w])\n\t"
#if NG_X3R_KO & 4
#define X3R_KA(s) ""
#else
#define X3R_KA(s) s
#endif
#if NG_X3R_KO & 8
#define X3R_KB(s) ""
#else
#define X3R_KB(s) s
#endif
// the previous region's six fragment reads (block 3), two per gap
#define X3R_PR0 X3R_RD("ds_read_b128 %[q0], %[bad] offset:%[o0]\n\t" "ds_read_b128 %[q1], %[bad] offset:%[o1]\n\t")
#define X3R_PR1 X3R_RD("ds_read_b128 %[q2], %[bad] offset:%[o2]\n\t" "ds_read_b128 %[q3], %[bad] offset:%[o3]\n\t")
#define X3R_PR2 X3R_RD("ds_read_b128 %[q4], %[bad] offset:%[o4]\n\t" "ds_read_b128 %[q5], %[bad] offset:%[o5]\n\t")
// two B pieces into the other LDS stage / two B pieces fetched / a raw row set fetched
#define X3R_ST0 X3R_KB("ds_write_b128 %[sa], %[sd0] offset:%[so0]\n\t")
#define X3R_ST1 X3R_KB("ds_write_b128 %[sa], %[sd1] offset:%[so1]\n\t")
#define X3R_LB0 X3R_KB("global_load_dwordx4 %[ld0], %[lo0], %[lb]\n\t")
#define X3R_LB1 X3R_KB("global_load_dwordx4 %[ld1], %[lo1], %[lb]\n\t")
#define X3R_LA0 X3R_KA("global_load_dwordx4 %[ld0], %[lo0], %[lb]\n\t")
#define X3R_LA1 X3R_KA("global_load_dwordx4 %[ld1], %[lo0], %[lb] offset:16\n\t")
#define X3R_MFMA_DRAIN asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

// KIND 2 (GEN): the launch's problems run the fused first backward pass (their epilogue branches and needs
// the accumulators in VGPRs: a kernel of its own, so that its register pressure is not the plain kernel's)
template <int BN, int KIND>
__device__ __forceinline__ void conv_x3r_persist(const NG_CONST X3Work* const wp, char* const ring, char* const stg_all, char* const sRed) {
    static_assert(BN == 128 || BN == 64, "256 x 128 or 256 x 64 block tiles");
    // KIND 0: plain problems; 1: problems that leave instance-norm partial sums (the plain kernel + the sums from the accumulators and NT
    // stores per item); 2: the fused first backward pass (its epilogue branches per element: the C++ path)
    constexpr bool GEN = KIND == 2, STATS = KIND == 1;
    using T = X3R<BN>;
    constexpr int NT = T::NT, B_TERM = T::B_TERM, STAGE = T::STAGE, BPT = T::BPT, PIECES = T::PIECES, LPR = T::LPR, RPP = T::RPP, SP = T::SP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = wp->first[wp->n];
    const int G = gridDim.x;
#ifdef NG_X3R_STAMP
    unsigned st_sum[16], st_last = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) st_sum[i] = 0;
    unsigned long long st_c0, st_r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c0), "=s"(st_r0) :: "memory");
#endif

    // ---------------- which tile is item i of this workgroup (as conv_x3_persist::locate)
    struct Item {
        const NG_CONST ConvParams* p;
        const char* in8; const char* w8; float* out;
        int m0, n0, nk, ntaps, run, k;
        long long w3_plane;
    };
    // the tap tables of the launch's (at most four) problems, lane i of tapq[k] = tap i of problem k: fetched ONCE, here -- per item they
    // cost sixteen scalar loads that hipcc issues one behind the other through one SGPR, each with its own wait: 1 400 of the 2 060 cycles
    // between two items, measured (scripts/diag/x3r_knockout.sh)
    int tapq[4];
    {
        const unsigned to = unsigned(lane & (NIRGAN_MAX_TAPS - 1)) * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* tb = ng_uniform_ptr(reinterpret_cast<const char*>((const int*)wp->p[q].tap_off));
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(tapq[q]) : "v"(to), "s"(tb) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(tapq[0]), "+v"(tapq[1]), "+v"(tapq[2]), "+v"(tapq[3]) :: "memory");
    }
    auto locate = [&](const int item, Item& t) {
        int k = 0, id = 0;
        if (wp->spread) {
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
            if (k < 0) { t.nk = -1; return; }
        } else {
            if (item >= total) { t.nk = -1; return; }
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
        t.k = k;
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

    // ---------------- the loader: a cursor TWO K-tiles ahead of the MFMAs, over the workgroup's items as one stream of K-tiles.  From
    // the cursor itself: the rows of row tiles 0-2 and the B pieces of K-tile j + 2 (while tile j is multiplied); row tile 3 of K-tile
    // j + 1 (converted one block later than the rest of its tile) through a SNAPSHOT of the cursor one tile old (a scalar base and the
    // lane's row offset).  One raw buffer per operand, refilled right behind its last use: one K-tile of latency budget.
    // The cursor's step (step_early / step_cursor below): a K-tile further inside the item, or -- on the item's last K-tile -- onto the first
    // K-tile of the NEXT item, whose state (`N`) was prepared outside the K loop, behind the previous item's epilogue (host: every
    // problem has at least three K-tiles, so the cursor crosses one item boundary per multiplied item).  Past the end of the
    // workgroup's items `N` is the last item once more: the fetches go on (their NUMBER is what the counted waits rely on), to valid
    // addresses, and are never used.
    struct Cur {
        const char* in8; const char* w8[3];
        int ntaps, run2, nk;
        unsigned goff[4], b0, b1;
        int tapv;
    };
    // m = (b OH + oh) OW + ow by two float reciprocals and a correction step each (m < 2^24: the estimate is off by at most one) instead
    // of two integer divisions (~50 VALU) per row: this runs once per item and row tile, with the matrix pipe idle
    auto split_row = [&](const int m, const int OHW, const int OW, const float r_ohw, const float r_ow, int& b, int& oh, int& ow) {
        b = int(float(m) * r_ohw);
        int r = m - b * OHW;
        b += r < 0 ? -1 : (r >= OHW ? 1 : 0);
        r = m - b * OHW;
        oh = int(float(r) * r_ow);
        ow = r - oh * OW;
        oh += ow < 0 ? -1 : (ow >= OW ? 1 : 0);
        ow = r - oh * OW;
    };
    auto prepare = [&](const Item& t, Cur& c) {
        const NG_CONST ConvParams& p = *t.p;
        const int OHW = p.OHW, OW = p.OW, pM = p.M;
        const float r_ohw = 1.0f / float(OHW), r_ow = 1.0f / float(OW);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int m = t.m0 + wave * 64 + mt * 16 + (lane & 15);
            m = m < pM ? m : pM - 1;
            int b, oh, ow;
            split_row(m, OHW, OW, r_ohw, r_ow, b, oh, ow);
            c.goff[mt] = unsigned(b * p.in_img + oh * p.in_stride * p.in_row + ow * p.in_stride * p.in_cs + p.in_org + (lane >> 4) * 8) * 4u;
        }
        const int pK = p.K;
        // piece q = wave + 4 i of a K-tile's 3 BPT pieces (term q / BPT, 16-row group q % BPT): the lane's row of group `wave` and, at
        // BN = 128, of group `wave + 4`
        const int b_c = (lane & 3) ^ x3_key(lane >> 2);
        c.b0 = unsigned((t.n0 + wave * 16 + (lane >> 2)) * pK + b_c * 8) * 2u;
        c.b1 = c.b0 + unsigned(64 * pK) * 2u;
        // the tap table in a VGPR (lane i holds tap i: the cursor picks with v_readlane)
        c.tapv = t.k == 0 ? tapq[0] : (t.k == 1 ? tapq[1] : (t.k == 2 ? tapq[2] : tapq[3]));
        c.in8 = ng_uniform_ptr(t.in8);
#pragma unroll
        for (int q = 0; q < 3; ++q) c.w8[q] = ng_uniform_ptr(t.w8 + (long long)q * t.w3_plane * 2);
        c.ntaps = t.ntaps;
        c.run2 = t.run * 2;
        c.nk = t.nk;
    };
    Cur L, N;                                   // the cursor's item / the item behind it
    int itemN = blockIdx.x;                     // the item `N` stands for
    int left = 0, ct = 0, cc = 0, kb = 0;       // K-tiles left in the cursor's item (this one included), its tap, its slice, (ct run + cc) 2
    const char* baseC = nullptr;                // the cursor's K-tile (j + 2 while tile j is multiplied)
    const char* baseP = nullptr;                // snapshot one tile old: K-tile j + 1
    const char* baseW[3] = {nullptr, nullptr, nullptr};      // the cursor's K-tile in the three weight planes
    unsigned goff3P = 0;
    auto cursor_bases = [&]() {
        const int toff = __builtin_amdgcn_readlane(L.tapv, ct);
        baseC = L.in8 + (long long)(toff + cc) * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) baseW[q] = L.w8[q] + kb;
    };
    bool crossS = false;
    // The cursor moves on: what it stood on becomes the snapshot.  In the K loop the step is taken in two parts: step_early() computes where
    // the cursor goes INSIDE its item (the common case) into shadow variables, in the middle of the tile -- hipcc spreads those ~25 scalar
    // instructions through the gaps between the regions behind it -- and step_cursor() at the tile's end commits them, or, on the item's
    // last K-tile (once per item: a real branch; as selects these were ~20 instructions of every K-tile, each paid in full with one
    // wave per SIMD; the empty asm keeps hipcc from turning the branch back into selects), puts the cursor on the next item's first K-tile.
    int n_ct = 0, n_cc = 0, n_kb = 0;
    const char* n_baseC = nullptr;
    const char* n_baseW[3] = {nullptr, nullptr, nullptr};
    bool n_valid = false;
    // (four pieces, each pinned where it is called by an empty volatile asm on what it computed: without the pins hipcc sinks all of it
    // to the commit at the tile's end)
    auto step_early = [&](auto piece_tag) __attribute__((always_inline)) {
        constexpr int PIECE = decltype(piece_tag)::value;
        if constexpr (PIECE < 0 || PIECE == 0) {
            // the next tap of this slice, or the first tap of the next slice
            const int ct1 = ct + 1;
            const bool wrap = ct1 == L.ntaps;
            n_cc = wrap ? cc + 32 : cc;
            n_kb = wrap ? n_cc * 2 : kb + L.run2;
            n_ct = wrap ? 0 : ct1;
            if constexpr (PIECE == 0) asm volatile("" : "+s"(n_cc), "+s"(n_kb), "+s"(n_ct));
        }
        if constexpr (PIECE < 0 || PIECE == 1) {
            if constexpr (PIECE == 1) asm volatile("" : "+s"(n_ct), "+s"(n_cc));      // (... and not earlier than here: hipcc hoisted pieces 1-3 to the tile's barrier)
            const int toff = __builtin_amdgcn_readlane(L.tapv, n_ct);
            n_baseC = L.in8 + (long long)(toff + n_cc) * 4;
            if constexpr (PIECE == 1) asm volatile("" : "+s"(n_baseC));
        }
        if constexpr (PIECE < 0 || PIECE == 2) {
            if constexpr (PIECE == 2) asm volatile("" : "+s"(n_kb));
            n_baseW[0] = L.w8[0] + n_kb;
            n_baseW[1] = L.w8[1] + n_kb;
            if constexpr (PIECE == 2) asm volatile("" : "+s"(n_baseW[0]), "+s"(n_baseW[1]));
        }
        if constexpr (PIECE < 0 || PIECE == 3) {
            if constexpr (PIECE == 3) asm volatile("" : "+s"(n_kb));
            n_baseW[2] = L.w8[2] + n_kb;
            if constexpr (PIECE == 3) asm volatile("" : "+s"(n_baseW[2]));
            n_valid = true;
        }
    };
    auto step_cursor = [&]() __attribute__((always_inline)) {
        baseP = baseC;
        goff3P = L.goff[3];
        crossS = left == 1;                     // (uniform)
        if (__builtin_expect(crossS, 0)) {
            asm volatile("" ::: "memory");
            ct = 0; cc = 0; kb = 0;
            left = N.nk;
            L.in8 = N.in8;
#pragma unroll
            for (int q = 0; q < 3; ++q) L.w8[q] = N.w8[q];
            L.ntaps = N.ntaps;
            L.run2 = N.run2;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) L.goff[mt] = N.goff[mt];
            L.b0 = N.b0;
            L.b1 = N.b1;
            L.tapv = N.tapv;
            cursor_bases();
        } else {
            if (!n_valid) step_early(std::integral_constant<int, -1>{});         // (the prologue's steps: nothing was computed ahead)
            ct = n_ct; cc = n_cc; kb = n_kb;
            left = left - 1;
            baseC = n_baseC;
#pragma unroll
            for (int q = 0; q < 3; ++q) baseW[q] = n_baseW[q];
        }
        n_valid = false;
    };
    f32x4 F[4][2];                              // raw rows: F[mt] = this lane's 8 k of its row of 16-row tile mt
    f32x4 Braw[PIECES];                         // raw B pieces (bf16 bits): 16 bytes per lane and piece
    u32x4 A[4][3];                              // the three terms of the CURRENT K-tile's rows, two bf16 per dword (rolling: A[mt] is replaced behind block mt)
    bf16x8 Bf[NT][3];                           // the current K-tile's B fragments (rolling: replaced behind their last use in block 3)
    // (inline asm with hand-counted waits: left to the compiler, the wait in front of a raw row set that was fetched in the PREVIOUS
    // iteration of the loop comes out as vmcnt(0) -- every K-tile would sit out the fetches issued half a tile ago.  The destination
    // registers are tied to the wait ("+v"), so nothing that reads them can be scheduled in front of it; scripts/check_x3r_asm.py
    // checks in the built code object that no instruction touches them between the load and its wait)
    auto loadA = [&](f32x4 (&f)[2], const char* base, const unsigned goff) {
        X3R_GLD(f[0], goff, base, 0);
        X3R_GLD(f[1], goff, base, 16);
    };
    // `stores_behind`: the K-tile right behind a full tile's plain epilogue -- its 4 SP = 32 output stores were issued behind every fetch
    // this tile waits for, and vmcnt counts loads and stores in one order: the waits leave them in flight (without that every item
    // would sit out the acknowledgement of the 32 KB it has just written: ~18 % of a plane GEMM's item, measured)
    int stores_behind = 0;                      // 0: none; 1: EST (a plain epilogue); 2: EST + NT (... and the NT stores of a statistics record)
    constexpr int EST = 4 * SP;
    static_assert(12 + EST + NT <= 63, "vmcnt is a six-bit counter");
    // (ONE asm statement per wait, the choice between its two counts a scalar branch INSIDE it: two statements in the arms of a C++ `if`
    // let the compiler merge the tied registers with copies in front of one of them -- copies of a destination whose load has not landed)
#define X3R_WAITV2(N) "s_cmp_eq_u32 %[sb], 0\n\ts_cbranch_scc1 .Lx3rw%=\n\ts_cmp_eq_u32 %[sb], 1\n\ts_cbranch_scc1 .Lx3ru%=\n\ts_waitcnt vmcnt(%[n2])\n\ts_branch .Lx3rv%=\n" \
                      ".Lx3ru%=:\n\ts_waitcnt vmcnt(%[n1])\n\ts_branch .Lx3rv%=\n.Lx3rw%=:\n\ts_waitcnt vmcnt(" #N ")\n.Lx3rv%=:"
    // (only an item's FIRST K-tile can have stores behind its fetches: the steady tile's waits are plain counts, no compare and branch --
    // five waits per tile, four instructions each, every one of them paid in full with one wave per SIMD)
    auto wait_raw = [&](f32x4 (&f)[2], auto n_tag, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        if constexpr (FIRST) {
            const int sb = __builtin_amdgcn_readfirstlane(stores_behind);
            if constexpr (decltype(n_tag)::value == 12) asm volatile(X3R_WAITV2(12) : "+v"(f[0]), "+v"(f[1]) : [sb] "s"(sb), [n1] "n"(12 + EST), [n2] "n"(12 + EST + NT) : "memory", "scc");
            else asm volatile(X3R_WAITV2(6) : "+v"(f[0]), "+v"(f[1]) : [sb] "s"(sb), [n1] "n"(6 + EST), [n2] "n"(6 + EST + NT) : "memory", "scc");
        } else {
            if constexpr (decltype(n_tag)::value == 12) asm volatile("s_waitcnt vmcnt(12)" : "+v"(f[0]), "+v"(f[1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" : "+v"(f[0]), "+v"(f[1]) :: "memory");
        }
    };
    // pair i (k = 2 i, 2 i + 1 of the lane's eight) of a raw row set into dword i of the three terms: 11 VALU (x3_split8's rule)
    auto convert_pair = [&](const f32x4 (&f)[2], u32x4 (&a)[3], const int i) {
        const float x0 = f[i >> 1][(2 * i) & 3], x1 = f[i >> 1][(2 * i + 1) & 3];
        const unsigned h = x3_pk(x0, x1);
        const float r0 = x3_sub(x0, __builtin_bit_cast(float, h << 16)), r1 = x3_sub(x1, __builtin_bit_cast(float, h & 0xffff0000u));
        const unsigned m = x3_pk(r0, r1);
        const float s0 = x3_sub(r0, __builtin_bit_cast(float, m << 16)), s1 = x3_sub(r1, __builtin_bit_cast(float, m & 0xffff0000u));
        a[0][i] = h;
        a[1][i] = m;
        a[2][i] = x3_pk(s0, s1);
    };
    // B piece i of the cursor's K-tile into its raw register set / from there into an LDS stage (image: igemm_x3.h -- 64-byte rows,
    // 16-byte chunk c of row r at chunk c ^ key(r): the swizzle is applied on the global side, the LDS side is lane-linear)
    auto loadB = [&](const int i) {              // (i is a constant of the unrolled caller; wave < 4: the term of piece wave + 4 i does not depend on the wave)
        const int term = i / (BPT / 4);         // BN = 128: 0 0 1 1 2 2; BN = 64: 0 1 2
        if (term == 0) { if (BPT == 8 && (i & 1)) X3R_GLD(Braw[i], L.b1, baseW[0], 0); else X3R_GLD(Braw[i], L.b0, baseW[0], 0); }
        else if (term == 1) { if (BPT == 8 && (i & 1)) X3R_GLD(Braw[i], L.b1, baseW[1], 0); else X3R_GLD(Braw[i], L.b0, baseW[1], 0); }
        else { if (BPT == 8 && (i & 1)) X3R_GLD(Braw[i], L.b1, baseW[2], 0); else X3R_GLD(Braw[i], L.b0, baseW[2], 0); }
    };
    const unsigned ring0 = unsigned(size_t((NG_LDS char*)ring));
    const unsigned b_wr = ring0 + unsigned(wave * 1024 + lane * 16);
    auto storeB = [&](const unsigned stage_off, const int i) {
        // piece q = wave + 4 i -> term (q / BPT), group (q % BPT): byte offset term * B_TERM + (q % BPT) * 1024 - wave * 1024 from b_wr
        const unsigned ad = b_wr + stage_off;
        if (BPT == 8) {
            if (i == 0) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[0]), "n"(0) : "memory");
            else if (i == 1) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[1]), "n"(4096) : "memory");
            else if (i == 2) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[2]), "n"(B_TERM) : "memory");
            else if (i == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[3 % PIECES]), "n"(B_TERM + 4096) : "memory");
            else if (i == 4) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[4 % PIECES]), "n"(2 * B_TERM) : "memory");
            else asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[5 % PIECES]), "n"(2 * B_TERM + 4096) : "memory");
        } else {
            if (i == 0) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[0]), "n"(0) : "memory");
            else if (i == 1) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[1]), "n"(B_TERM) : "memory");
            else asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ad), "v"(Braw[2]), "n"(2 * B_TERM) : "memory");
        }
    };

    // ---------------- compute state
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swz = ((lane >> 4) ^ x3_key(lane & 15)) << 4;
    const unsigned b_rd = unsigned((lane & 15) * 64 + swz);
    int sj = 0;                                 // LDS stage of the K-tile being multiplied
    // one region (see X3R_PAIR above): row tile MT against column tiles 2 P and 2 P + 1; pair P of raw set SRC becomes dword P of A[SRC]'s
    // three terms; WAIT >= 0: behind a counted lgkmcnt wait (block 0); READ: both column tiles' fragments of the NEXT K-tile behind their
    // last MFMAs (block 3)
    auto region = [&](auto first_tag, auto mt_tag, auto p_tag, auto src_tag, auto wait_tag, const unsigned bad, const unsigned st_ad, const char* base, const unsigned goff) __attribute__((always_inline)) {
        constexpr int MT = decltype(mt_tag)::value, P = decltype(p_tag)::value, SRC = decltype(src_tag)::value, WAIT = decltype(wait_tag)::value;
        constexpr int N0 = 2 * P, N1 = 2 * P + 1;
        constexpr bool FIRST = decltype(first_tag)::value;
        // what rides in the region's free gaps (see X3R_PAIR): ST -- B pieces 2 (P - 1), 2 (P - 1) + 1 of K-tile j + 1 into the other LDS stage
        // (block 0, regions 1-3); LB -- pieces 2 P, 2 P + 1 of K-tile j + 2 fetched (block 1, regions 0-2: behind the stores that emptied
        // their registers); LA -- the block's raw row set fetched again (last region of blocks 0-2; block 3's follows its region: the
        // operand list is full); PR -- the fragments of the PREVIOUS region's column tiles read for the next K-tile (block 3)
        constexpr bool ST = MT == 0 && P >= 1, LB = MT == 1 && P < 3, LA = P == 3 && MT < 3, PR = MT == 3 && P >= 1;
        static_assert((WAIT >= 0) == (MT == 0), "block 0's regions wait for their fragments");
        static_assert(PIECES == 6 && BPT == 8, "B pieces: two per term");
        f32x4& c0 = acc[MT][N0];
        f32x4& c1 = acc[MT][N1];
        const float x0 = F[SRC][P >> 1][(2 * P) & 3], x1 = F[SRC][P >> 1][(2 * P + 1) & 3];
        unsigned h, m, l;
        constexpr int I0 = ST ? 2 * (P - 1) : (LB ? 2 * P : 0), I1 = I0 + 1;       // the region's two B pieces
        constexpr int SO0 = (I0 / 2) * B_TERM, SO1 = SO0 + 4096;                   // piece i in its stage: term i / 2, row group i % 2
        constexpr int NP0 = PR ? N0 - 2 : 0, NP1 = PR ? N1 - 2 : 1;
#define X3R_ACC [c0] "+a"(c0), [c1] "+a"(c1)
#define X3R_B_INOUT [b0] "+a"(Bf[N0][0]), [b1] "+a"(Bf[N0][1]), [b2] "+a"(Bf[N0][2]), [d0] "+a"(Bf[N1][0]), [d1] "+a"(Bf[N1][1]), [d2] "+a"(Bf[N1][2])
#define X3R_B_IN [b0] "a"(Bf[N0][0]), [b1] "a"(Bf[N0][1]), [b2] "a"(Bf[N0][2]), [d0] "a"(Bf[N1][0]), [d1] "a"(Bf[N1][1]), [d2] "a"(Bf[N1][2])
#define X3R_TERMS [h] "=&v"(h), [m] "=&v"(m), [l] "=&v"(l)
#define X3R_A_IN [ah] "v"(A[MT][0]), [am] "v"(A[MT][1]), [al] "v"(A[MT][2]), [x0] "v"(x0), [x1] "v"(x1)
#define X3R_PR_OUT [q0] "+a"(Bf[NP0][0]), [q1] "+a"(Bf[NP0][1]), [q2] "+a"(Bf[NP0][2]), [q3] "+a"(Bf[NP1][0]), [q4] "+a"(Bf[NP1][1]), [q5] "+a"(Bf[NP1][2])
#define X3R_PR_IN [bad] "v"(bad), [o0] "n"(NP0 * 1024), [o1] "n"(B_TERM + NP0 * 1024), [o2] "n"(2 * B_TERM + NP0 * 1024), \
                  [o3] "n"(NP1 * 1024), [o4] "n"(B_TERM + NP1 * 1024), [o5] "n"(2 * B_TERM + NP1 * 1024)
#define X3R_ST_IN [sa] "v"(st_ad), [sd0] "v"(Braw[I0]), [sd1] "v"(Braw[I1]), [so0] "n"(SO0), [so1] "n"(SO1)
// (NOT early-clobber: the row set's old value is an INPUT of the same statement -- x0, x1 are two of its elements, read by the
// conversion long before the fetch is issued -- and with "=&v" the new value had to live in other registers than the old one: hipcc
// then copied it back at the loop's end, in front of the wait that covers the fetch (scripts/check_x3_asm.py))
#define X3R_LB_OUT [ld0] "=v"(Braw[I0]), [ld1] "=v"(Braw[I1])
#define X3R_LA_OUT [ld0] "=v"(F[SRC][0]), [ld1] "=v"(F[SRC][1])
#define X3R_EMIT(MA, T, CLOB) \
        if constexpr (ST && LA) { \
            const char* const lb = base; \
            asm volatile(X3R_WAITL X3R_PAIR(MA, MA, T, "", X3R_ST0, X3R_ST1, X3R_LA0, X3R_LA1) : X3R_ACC, X3R_B_INOUT, X3R_TERMS, X3R_LA_OUT \
                         : X3R_A_IN, [w] "n"(WAIT), X3R_ST_IN, [lo0] "v"(goff), [lb] "s"(lb) : CLOB); \
        } else if constexpr (ST) { \
            asm volatile(X3R_WAITL X3R_PAIR(MA, MA, T, "", "", X3R_ST0, X3R_ST1, "") : X3R_ACC, X3R_B_INOUT, X3R_TERMS : X3R_A_IN, [w] "n"(WAIT), X3R_ST_IN : CLOB); \
        } else if constexpr (WAIT >= 0) { \
            asm volatile(X3R_WAITL X3R_PAIR(MA, MA, T, "", "", "", "", "") : X3R_ACC, X3R_B_INOUT, X3R_TERMS : X3R_A_IN, [w] "n"(WAIT) : CLOB); \
        } else if constexpr (LB) { \
            const char* const lb = P == 0 ? baseW[0] : (P == 1 ? baseW[1] : baseW[2]); \
            const unsigned lo0 = L.b0, lo1 = L.b1; \
            asm volatile(X3R_PAIR(MA, MA, T, "", "", X3R_LB0, X3R_LB1, "") : X3R_ACC, X3R_TERMS, X3R_LB_OUT : X3R_A_IN, X3R_B_IN, [lo0] "v"(lo0), [lo1] "v"(lo1), [lb] "s"(lb) : CLOB); \
        } else if constexpr (LA) { \
            const char* const lb = base; \
            asm volatile(X3R_PAIR(MA, MA, T, "", "", X3R_LA0, X3R_LA1, "") : X3R_ACC, X3R_TERMS, X3R_LA_OUT : X3R_A_IN, X3R_B_IN, [lo0] "v"(goff), [lb] "s"(lb) : CLOB); \
        } else if constexpr (PR) { \
            asm volatile(X3R_PAIR(MA, MA, T, X3R_PR0, X3R_PR1, X3R_PR2, "", "") : X3R_ACC, X3R_TERMS, X3R_PR_OUT : X3R_A_IN, X3R_B_IN, X3R_PR_IN : CLOB); \
        } else { \
            asm volatile(X3R_PAIR(MA, MA, T, "", "", "", "", "") : X3R_ACC, X3R_TERMS : X3R_A_IN, X3R_B_IN : CLOB); \
        }
        if constexpr (P & 1) {
            if constexpr (FIRST) { X3R_EMIT(X3R_MZ, X3R_TB, X3R_CLOB_B) } else { X3R_EMIT(X3R_MC, X3R_TB, X3R_CLOB_B) }
        } else {
            if constexpr (FIRST) { X3R_EMIT(X3R_MZ, X3R_TA, X3R_CLOB_A) } else { X3R_EMIT(X3R_MC, X3R_TA, X3R_CLOB_A) }
        }
#undef X3R_EMIT
#undef X3R_ACC
#undef X3R_B_INOUT
#undef X3R_B_IN
#undef X3R_TERMS
#undef X3R_A_IN
#undef X3R_PR_OUT
#undef X3R_PR_IN
#undef X3R_ST_IN
#undef X3R_LB_OUT
#undef X3R_LA_OUT
        A[SRC][0][P] = h;
        A[SRC][1][P] = m;
        A[SRC][2][P] = l;
    };
#define X3R_RB(NTI, ad) { X3R_DSR(Bf[NTI][0], ad, (NTI) * 1024); X3R_DSR(Bf[NTI][1], ad, B_TERM + (NTI) * 1024); X3R_DSR(Bf[NTI][2], ad, 2 * B_TERM + (NTI) * 1024); }

    // ---------------- epilogue of the item whose K-tiles are being multiplied (`E`), in four slices of 16 rows per wave
    // (what outlives an item's lookup: the epilogue's view of it -- seven scalar registers per slot, three slots)
    struct Slot {
        const NG_CONST ConvParams* p;
        float* out;
        int m0, n0, nk;
    };
    Slot E;
    char* const stg8 = stg_all + wave * T::STG;
    float* const stg = reinterpret_cast<float*>(stg8);
    const unsigned stg_lds = unsigned(size_t((NG_LDS char*)stg8));
    const int chunk = lane % LPR, lrow = lane / LPR;
    struct Epi {
        int pM, OHW, OW, OH, out_img, out_row, out_px, out_org, f_img, f_row, f_px, f_org, pC, span;
        const float* f_y;
        bool fused, stats;
        float fneg;
        int n, nq, nc, m, b, oh, ow, fb, mbase;
        f32x4 bv, fm, fr, s1, s2;
        float k0[NT], t1[NT], t2[NT];
    };
    auto epi_begin = [&](Epi& e) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) e.k0[nt] = e.t1[nt] = e.t2[nt] = 0.f;
        const NG_CONST ConvParams& p = *E.p;
        e.pM = p.M; e.OHW = p.OHW; e.OW = p.OW; e.OH = p.OHW / p.OW; e.out_img = p.out_img; e.out_row = p.out_row * p.out_stride;
        e.out_px = p.out_cs * p.out_stride; e.out_org = p.out_org; e.f_img = p.f_img; e.f_row = p.f_row * p.out_stride; e.f_px = p.ch * p.out_stride;
        e.f_org = p.f_org; e.pC = p.ch; e.span = p.N / p.ch;
        e.f_y = p.f_y;
        e.mbase = E.m0 + wave * 64;
        e.fused = e.f_y != nullptr;
        e.stats = p.stats != nullptr && e.mbase < e.pM;
        e.fneg = p.f_act == NIRGAN_ACT_RELU ? 0.f : (p.f_act == NIRGAN_ACT_LRELU ? p.f_slope : 1.f);
        e.n = E.n0 + chunk * 4;
        e.bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
            // (inline asm with its own wait: a compiler-visible load inside the loop makes hipcc guard its destination registers with
            // s_waitcnt vmcnt(0) at the top of EVERY K-tile -- the fetches of the previous tile would be drained each time)
            const char* bb = ng_uniform_ptr(reinterpret_cast<const char*>(p.bias));
            const unsigned bo = unsigned(e.n) * 4u;
            asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(e.bv) : "v"(bo), "s"(bb) : "memory");
        }
        e.m = e.mbase + lrow;
        const int mc = e.m < e.pM ? e.m : e.pM - 1;
        split_row(mc, e.OHW, e.OW, 1.0f / float(e.OHW), 1.0f / float(e.OW), e.b, e.oh, e.ow);
        e.fb = (e.mbase < e.pM ? e.mbase : e.pM - 1) / e.OHW;
        e.nq = e.n >= e.pC ? 1 : 0;
        e.nc = e.n - e.nq * e.pC;
        e.fm = e.fr = e.s1 = e.s2 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e.fused) {
            e.fm = *reinterpret_cast<const f32x4*>(p.f_mean + size_t(e.fb) * e.pC + e.nc);
            e.fr = *reinterpret_cast<const f32x4*>(p.f_rstd + size_t(e.fb) * e.pC + e.nc);
        }
    };
    // slice mt: the wave's rows 16 mt .. 16 mt + 15.  The instance-norm partial sums (nirgan_conv_desc.stats_ws) are taken from the
    // accumulators as they stand (the contract and the order of conv_x3_persist: per column {k = the chunk's first row, sum (v - k),
    // sum (v - k)^2, 64} over the wave's 64 rows), the rows then go through the staging block and leave 16 bytes per lane
    // the instance-norm partial sums of slice mt from the accumulators as they stand: per column sum (v - k), sum (v - k)^2 over the lane's
    // four rows, rows in order (the contract and the order of conv_x3_persist).  ONE asm statement per column tile, straight from the
    // accumulator registers: written in C++ hipcc reads all 128 accumulators of an item into VGPRs up front -- loop-carried values went
    // to scratch, and every reload behind the epilogue's stores is a `s_waitcnt vmcnt(0)` that sits out their acknowledgement
    auto stats_acc = [&](Epi& e, auto mt_tag) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_tag)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (mt == 0) {
                e.k0[nt] = __shfl(acc[0][nt][0], lane & 15, 64);
                e.t1[nt] = 0.f;
                e.t2[nt] = 0.f;
            }
            const f32x4 c = acc[mt][nt];
            const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            float tmp;
            asm volatile("v_accvgpr_read_b32 %[t], %[c0]\n\tv_sub_f32 %[t], %[t], %[k]\n\tv_add_f32 %[s1], %[s1], %[t]\n\tv_fmac_f32 %[s2], %[t], %[t]\n\t"
                         "v_accvgpr_read_b32 %[t], %[c1]\n\tv_sub_f32 %[t], %[t], %[k]\n\tv_add_f32 %[s1], %[s1], %[t]\n\tv_fmac_f32 %[s2], %[t], %[t]\n\t"
                         "v_accvgpr_read_b32 %[t], %[c2]\n\tv_sub_f32 %[t], %[t], %[k]\n\tv_add_f32 %[s1], %[s1], %[t]\n\tv_fmac_f32 %[s2], %[t], %[t]\n\t"
                         "v_accvgpr_read_b32 %[t], %[c3]\n\tv_sub_f32 %[t], %[t], %[k]\n\tv_add_f32 %[s1], %[s1], %[t]\n\tv_fmac_f32 %[s2], %[t], %[t]"
                         : [t] "=&v"(tmp), [s1] "+v"(e.t1[nt]), [s2] "+v"(e.t2[nt]) : [c0] "a"(c0), [c1] "a"(c1), [c2] "a"(c2), [c3] "a"(c3), [k] "v"(e.k0[nt]));
        }
    };
    auto slice = [&](Epi& e, auto mt_tag, auto mode_tag) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_tag)::value;
        constexpr bool PLAIN = decltype(mode_tag)::value == 1;          // no statistics, no fused pass: straight-line code (it is interleaved with MFMAs)
        if (!PLAIN && e.stats) stats_acc(e, mt_tag);
        if constexpr (PLAIN) {
            // the staging traffic in inline asm with its own waits (compiler-visible LDS accesses next to hand-counted ones get a full
            // drain in front of them): 32 ds_write_b32 -- rows 4 apart of one column per instruction: two lanes per bank, free on a
            // store -- then 8 ds_read_b128 of whole row segments
            const unsigned sw = stg_lds + unsigned(((lane >> 4) * 4 * BN + (lane & 15)) * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (straight from the accumulator registers: ds_write takes AGPR data; a copy into VGPRs first -- the compiler hoists all
                    // 128 of an item -- is what pushed loop-carried values into scratch.  Offset (r BN + 16 nt) floats, one case per r:
                    // the immediate must be a constant of the instruction)
                    const float v = acc[mt][nt][r];
                    if (nt == 0) {
                        if (r == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(0) : "memory");
                        else if (r == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(BN * 4) : "memory");
                        else if (r == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(2 * BN * 4) : "memory");
                        else asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(3 * BN * 4) : "memory");
                    }
#define X3R_STW(NTI) else if (nt == NTI) { \
                        if (r == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(NTI * 64) : "memory"); \
                        else if (r == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(BN * 4 + NTI * 64) : "memory"); \
                        else if (r == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(2 * BN * 4 + NTI * 64) : "memory"); \
                        else asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(sw), "a"(v), "n"(3 * BN * 4 + NTI * 64) : "memory"); }
                    X3R_STW(1) X3R_STW(2) X3R_STW(3) X3R_STW(4) X3R_STW(5) X3R_STW(6) X3R_STW(7)
#undef X3R_STW
                }
            int ooff[SP];
            bool ok[SP];
#pragma unroll
            for (int pass = 0; pass < SP; ++pass) {
                ok[pass] = e.m < e.pM;
                ooff[pass] = e.b * e.out_img + e.oh * e.out_row + e.ow * e.out_px + e.out_org + e.n;
                // (selects, no loops: host -- OW >= RPP, see conv_x3r_ok)
                e.m += RPP;
                e.ow += RPP;
                const bool wrap_w = e.ow >= e.OW;
                e.ow -= wrap_w ? e.OW : 0;
                e.oh += wrap_w ? 1 : 0;
                const bool wrap_h = e.oh >= e.OH;
                e.oh -= wrap_h ? e.OH : 0;
                e.b += wrap_h ? 1 : 0;
            }
            f32x4 v[SP];
            const unsigned sr = stg_lds + unsigned((lrow * BN + chunk * 4) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int pass = 0; pass < SP; ++pass) asm volatile("ds_read_b128 %0, %1" : "=v"(v[pass]) : "v"(sr + unsigned(pass * RPP * BN * 4)) : "memory");
            if (SP == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4 % SP]), "+v"(v[5 % SP]), "+v"(v[6 % SP]), "+v"(v[7 % SP]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
            for (int pass = 0; pass < SP; ++pass)
                if (ok[pass]) *reinterpret_cast<f32x4*>(E.out + ooff[pass]) = v[pass] + e.bv;
            return;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[((lane >> 4) * 4 + r) * BN + nt * 16 + (lane & 15)] = acc[mt][nt][r];
        int ooff[SP];
        bool ok[SP];
        f32x4 yv[SP];
#pragma unroll
        for (int pass = 0; pass < SP; ++pass) {
            ok[pass] = e.m < e.pM;
            ooff[pass] = e.b * e.out_img + e.oh * e.out_row + e.ow * e.out_px + e.out_org + e.n;
            if (!PLAIN && e.fused && ok[pass]) yv[pass] = *reinterpret_cast<const f32x4*>(e.f_y + (size_t(e.b) * e.f_img + size_t(e.oh) * e.f_row + size_t(e.ow) * e.f_px + e.f_org + e.n));
            // (selects, no loops: host -- OW >= RPP, see conv_x3r_ok)
            e.m += RPP;
            e.ow += RPP;
            const bool wrap_w = e.ow >= e.OW;
            e.ow -= wrap_w ? e.OW : 0;
            e.oh += wrap_w ? 1 : 0;
            const bool wrap_h = e.oh >= e.OH;
            e.oh -= wrap_h ? e.OH : 0;
            e.b += wrap_h ? 1 : 0;
        }
#pragma unroll
        for (int pass = 0; pass < SP; ++pass) {
            if (ok[pass]) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + (pass * RPP + lrow) * BN + chunk * 4);
                v += e.bv;
                *reinterpret_cast<f32x4*>(E.out + ooff[pass]) = v;
                if (!PLAIN && e.fused) {
                    const f32x4 z = (yv[pass] - e.fm) * e.fr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float gz = z[q] > 0.f ? v[q] : v[q] * e.fneg;
                        e.s1[q] += gz;
                        e.s2[q] += gz * z[q];
                    }
                }
            }
        }
    };
    // A FULL tile's plain epilogue.  It is ISSUE-bound (one wave per SIMD: every instruction of it is exposed), so it is kept short:
    //   W (per slice): 16 ds_write2_b32 straight from the accumulator registers -- two rows of one column per instruction (rows 4 apart
    //     per 16-lane group: two lanes per bank, free on a store); the accumulators are NOT zeroed (the next item's first K-tile
    //     starts them with C = 0);
    //   R: 8 ds_read_b128 of whole row segments, no wait in front (a wave's LDS operations execute in issue order);
    //   S: 8 stores, scalar base + 32-bit offset, straight from the registers the reads landed in (+ the bias only where there is one);
    //   the byte offsets of the lane's rows: ONE add per slice while the slice's 16 rows do not cross an image row (a scalar test);
    //     the select walk otherwise.
    // Two staging halves, order  W0 R0 [S0 | W1] R1 [S1 | W2] R2 [S2 | W3] R3 S3  ([S | W]: a store, a column block, a store, ...).
    // column blocks of slice mt from the accumulators into the staging half (mt & 1) (only >= 0: column block `only` alone)
    auto stage_w = [&](auto mt_tag, const int only = -1) __attribute__((always_inline)) {        // (only >= 0: column block `only` alone)
        constexpr int mt = decltype(mt_tag)::value;
        const unsigned sw = stg_lds + unsigned(((lane >> 4) * 4 * BN + (lane & 15)) * 4) + (mt & 1) * (16 * BN * 4), sw2 = sw + 2 * BN * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if ((only >= 0 && nt != only) || (NG_X3R_KO & 32)) continue;
            const f32x4 c = acc[mt][nt];
            const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            // (offsets in dwords: column block nt, the second row BN dwords on)
#define X3R_STW4(NTI) if (nt == NTI) asm volatile("ds_write2_b32 %0, %2, %3 offset0:%6 offset1:%7\n\tds_write2_b32 %1, %4, %5 offset0:%6 offset1:%7" \
                    :: "v"(sw), "v"(sw2), "a"(c0), "a"(c1), "a"(c2), "a"(c3), "n"(NTI * 16), "n"(NTI * 16 + BN) : "memory");
            X3R_STW4(0) X3R_STW4(1) X3R_STW4(2) X3R_STW4(3) X3R_STW4(4) X3R_STW4(5) X3R_STW4(6) X3R_STW4(7)
#undef X3R_STW4
        }
    };
    auto epilogue_full = [&](Epi& e, auto bias_tag, auto&& mid) __attribute__((always_inline)) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        const unsigned sr0 = stg_lds + unsigned((lrow * BN + chunk * 4) * 4);
        constexpr unsigned HALF = 16 * BN * 4;
        const char* const obase = ng_uniform_ptr(reinterpret_cast<const char*>(E.out));
        // byte steps of the row walk: RPP pixels on; a row wrap; a sample wrap
        const int d0 = RPP * e.out_px * 4, dW = (e.out_row - e.OW * e.out_px) * 4, dH = (e.out_img - e.OH * e.out_row) * 4;
        int off = (e.b * e.out_img + e.oh * e.out_row + e.ow * e.out_px + e.out_org + e.n) * 4;
        // (uniform) the column of the slice's first row: the slice wraps iff its 16 rows cross the end of an image row
        int s_ow = __builtin_amdgcn_readfirstlane(e.ow);       // (lane 0 stands on the slice's first row)
        const int s_OW = e.OW;
        static_assert(7 * 16 + BN <= 255, "ds_write2_b32 offsets are 8-bit dword counts");
        f32x4 v0[SP], v1[SP];
        auto R = [&](auto mt_tag, f32x4 (&vv)[SP]) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_tag)::value;
            const unsigned sr = sr0 + (mt & 1) * HALF;
#pragma unroll
            for (int pass = 0; pass < SP; ++pass) {
                if (NG_X3R_KO & 32) { asm volatile("" : "=v"(vv[pass]) :: "memory"); continue; }
                if (pass == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(vv[0]) : "v"(sr) : "memory");
#define X3R_RDP(P) else if (pass == P) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vv[P % SP]) : "v"(sr), "n"(P * RPP * BN * 4) : "memory");
                X3R_RDP(1) X3R_RDP(2) X3R_RDP(3) X3R_RDP(4) X3R_RDP(5) X3R_RDP(6) X3R_RDP(7)
#undef X3R_RDP
            }
        };
        // S with the NEXT slice's W woven in, a column block behind every store (NEXT < 4): the LDS store path and the global store
        // path each move 32 KB per wave and slice -- back to back they add up, alternating they overlap
        // the byte offsets of a slice's eight row segments (in front of the slice's reads: nothing but the wait stands between the reads
        // and the stores -- and no branch, which scripts/check_x3_asm.py's linear walk could not follow)
        auto Aoff = [&](int (&ooff)[SP]) __attribute__((always_inline)) {
            if (s_ow + 16 <= s_OW) {
                // the slice stays inside one image row: the lane's rows are d0 bytes apart
#pragma unroll
                for (int pass = 0; pass < SP; ++pass) ooff[pass] = off + pass * d0;
                off += SP * d0;
                e.ow += 16;
                // the lane's first row of the NEXT slice may lie behind the end of this image row (the slice ended at it, or -- odd OW --
                // one pixel short of it: the second of the lane pair is then already in the next row)
                const bool wrap_w = e.ow >= e.OW;
                e.ow -= wrap_w ? e.OW : 0;
                off += wrap_w ? dW : 0;
                e.oh += wrap_w ? 1 : 0;
                const bool wrap_h = e.oh >= e.OH;
                e.oh -= wrap_h ? e.OH : 0;
                off += wrap_h ? dH : 0;
                s_ow += 16;
                s_ow -= s_ow >= s_OW ? s_OW : 0;
            } else {
#pragma unroll
                for (int pass = 0; pass < SP; ++pass) {
                    ooff[pass] = off;
                    off += d0;
                    e.ow += RPP;
                    const bool wrap_w = e.ow >= e.OW;
                    e.ow -= wrap_w ? e.OW : 0;
                    off += wrap_w ? dW : 0;
                    e.oh += wrap_w ? 1 : 0;
                    const bool wrap_h = e.oh >= e.OH;
                    e.oh -= wrap_h ? e.OH : 0;
                    off += wrap_h ? dH : 0;
                }
                s_ow += 16;
                s_ow -= s_ow >= s_OW ? s_OW : 0;    // (host: OW >= 16 for this kernel's problems, see conv_x3r_ok)
            }
        };
        auto S = [&](auto next_tag, f32x4 (&vv)[SP], const int (&ooff)[SP]) __attribute__((always_inline)) {
            constexpr int NEXT = decltype(next_tag)::value;
            // the slice's reads have landed (behind them at most the 16 stores of the next slice's W: lgkmcnt counts to 15, the LDS
            // pipe returns in order -- all but the 15 youngest done means every read done)
            if constexpr (false) {
                if (SP == 8) asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4 % SP]), "+v"(vv[5 % SP]), "+v"(vv[6 % SP]), "+v"(vv[7 % SP]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]) :: "memory");
            } else {
                if (SP == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4 % SP]), "+v"(vv[5 % SP]), "+v"(vv[6 % SP]), "+v"(vv[7 % SP]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]) :: "memory");
            }
            const char* const ob = obase;           // (a local: clang refuses an asm operand that names the enclosing lambda's variable)
            // the bias on all eight row segments first, then the eight stores back to back from the registers the reads landed in, then
            // two wait states: a store of more than 8 bytes reads its data registers late -- a VALU write to them within two wait
            // states of the store corrupts what is stored (seen: the third / fourth dword of every segment whose registers the next
            // segment's v_pk_add reused), and the compiler's hazard recognizer does not see into the asm
            if constexpr (BIAS) {
#pragma unroll
                for (int pass = 0; pass < SP; ++pass) vv[pass] += e.bv;
            }
            static_assert(SP == NT, "one column block of the next slice per store");
#pragma unroll
            for (int pass = 0; pass < SP; ++pass) {
                const unsigned oo = unsigned(ooff[pass]);
                if (!(NG_X3R_KO & 64)) asm volatile("global_store_dwordx4 %1, %0, %2" : "+v"(vv[pass]) : "v"(oo), "s"(ob) : "memory");
                if constexpr (NEXT < 4) stage_w(next_tag, pass);
            }
            // (the segments stay LIVE up to here: "+v" alone lets hipcc hand a store's data register to the next value at once -- seen under
            // register pressure: a v_or_b32 of an address one wait state behind the store that still had to read the register)
            asm volatile("s_nop 1" :: "v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]), "v"(vv[4 % SP]), "v"(vv[5 % SP]), "v"(vv[6 % SP]), "v"(vv[7 % SP]) : "memory");
        };
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
        using J4 = std::integral_constant<int, 4>;
        // (slice 0 is in the staging block already: written in front of the epilogue's set-up, whose ~400 cycles of address arithmetic and
        // scalar loads then run beside the LDS writes)
        int oo[SP];
        Aoff(oo); R(J0{}, v0); S(J1{}, v0, oo);             // (the reads are waited for with lgkmcnt(0): nothing is behind them yet)
        Aoff(oo); R(J1{}, v1); S(J2{}, v1, oo);
        // the item behind the next one is looked up and its loader state prepared HERE: ~1 100 cycles, most of them latency of dependent
        // scalar loads, while the store path works off the 16 stores just issued (a CU takes 33 B per cycle, scripts/diag/x3r_knockout.sh:
        // the epilogue's time is that of its 128 KB of stores)
        mid();
        Aoff(oo); R(J2{}, v0); S(J3{}, v0, oo);
        Aoff(oo); R(J3{}, v1); S(J4{}, v1, oo);
    };
    // the wave's statistics record of its 64 rows as NT store instructions: lane (field = lane >> 4, column = lane & 15) writes field
    // {k, sum (v - k), sum (v - k)^2, 64} of its column (the four fields are pC floats apart; every lane holds every sum after the two
    // shuffles).  (The full tiles' fast path: their number is part of the next item's counted waits)
    auto stats_flush = [&](Epi& e) __attribute__((always_inline)) {
        const NG_CONST ConvParams& p = *E.p;
        const int b = e.mbase / e.OHW;
        float* sp = p.stats + (size_t(b) * p.stats_cps + p.stats_chunk0 + ((e.mbase - b * e.OHW) >> 6) * e.span) * 4 * e.pC;
        const char* const sb8 = ng_uniform_ptr(reinterpret_cast<const char*>(sp));
        const int field = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s1 = e.t1[nt], s2 = e.t2[nt];
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            int col = E.n0 + nt * 16 + (lane & 15);
            if (col >= e.pC) col += 3 * e.pC;
            const float val = field == 0 ? e.k0[nt] : (field == 1 ? s1 : (field == 2 ? s2 : 64.f));
            const unsigned so = unsigned(field * e.pC + col) * 4u;
            asm volatile("global_store_dword %0, %1, %2" :: "v"(so), "v"(val), "s"(sb8) : "memory");
        }
    };
    auto epi_finish = [&](Epi& e) {
        const NG_CONST ConvParams& p = *E.p;
        if (e.stats) {
            const int b = e.mbase / e.OHW;
            float* sp = p.stats + (size_t(b) * p.stats_cps + p.stats_chunk0 + ((e.mbase - b * e.OHW) >> 6) * e.span) * 4 * e.pC;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float s1 = e.t1[nt], s2 = e.t2[nt];
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                int col = E.n0 + nt * 16 + (lane & 15);
                if (col >= e.pC) col += 3 * e.pC;
                if (lane < 16) {
                    sp[col] = e.k0[nt];
                    sp[e.pC + col] = s1;
                    sp[2 * e.pC + col] = s2;
                    sp[3 * e.pC + col] = 64.f;
                }
            }
        }
        if (e.fused) {
            // first pass of the consumer layer's instance-norm backward: this wave's 64 rows, then the two waves of a 128-row chunk
            // (waves 2 c, 2 c + 1) join through a small LDS array in a fixed order
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
                    e.s1[q] += __shfl_xor(e.s1[q], o, 64);
                    e.s2[q] += __shfl_xor(e.s2[q], o, 64);
                }
            f32x4* const red = reinterpret_cast<f32x4*>(sRed);        // 4 waves x LPR x 2 sums
            if (lane < LPR) {
                red[(wave * LPR + chunk) * 2] = e.s1;
                red[(wave * LPR + chunk) * 2 + 1] = e.s2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // (uniform: `fused` is a constant of the item's problem, the items of a launch agree)
            asm volatile("" ::: "memory");
            if ((wave & 1) == 0 && lane < LPR && e.mbase < e.pM) {
                const f32x4 t1 = e.s1 + red[((wave + 1) * LPR + chunk) * 2], t2 = e.s2 + red[((wave + 1) * LPR + chunk) * 2 + 1];
                float* pp = p.f_part + (size_t(e.fb) * p.f_cps + p.f_chunk0 + ((e.mbase - e.fb * e.OHW) >> 7) * e.span + e.nq) * 2 * e.pC + e.nc;
                *reinterpret_cast<f32x4*>(pp) = t1;
                *reinterpret_cast<f32x4*>(pp + e.pC) = t2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // the array is free for the next item
            asm volatile("" ::: "memory");
        }
    };

    // ---------------- one K-tile.
    // Vector-memory operations of a tile, in order: L3 (2 loads, K-tile j + 1; last region of block 0), the B pieces (6, K-tile j + 2;
    // one per region of block 1), L0 (2, K-tile j + 2; last region of block 1), L1 (2), L2 (2) = 14.  A raw set is waited for one tile
    // after its loads: the B pieces in the first region of block 0 (6 behind them: L0, L1, L2), row tile 3 there too (12), row tile 0
    // in front of this tile's B pieces (6), row tiles 1 and 2 with 12 behind them.
    // The schedule is written region by region, each region one asm block (X3R_PAIR): the loop's instruction stream is what is
    // written here, in this order (every statement of the loop is a volatile asm; hipcc allocates the registers).
    auto tile = [&](auto first_tag) __attribute__((always_inline)) {
        const unsigned bnext = ring0 + unsigned((sj ^ 1) * STAGE) + b_rd;
        const unsigned stage_next = unsigned((sj ^ 1) * STAGE);
        // regions of one block: row tile MT; raw set SRC -> A[SRC]; `base / goff` = where SRC is fetched again
        auto block = [&](auto mt_tag, auto src_tag, const char* base, const unsigned goff) __attribute__((always_inline)) {
            constexpr int MT = decltype(mt_tag)::value, SRC = decltype(src_tag)::value;
            // the conversion's four pairs: NT = 8 -- a half per region (pair nt / 2); NT = 4 -- a pair per region (both halves: two calls)
            if (MT == 0) {
                // K-tile j + 1's pieces of B have landed (fetched in block 1 of the previous tile)
                if constexpr (decltype(first_tag)::value) {
                    const int sb = __builtin_amdgcn_readfirstlane(stores_behind);
                    if (PIECES == 6) asm volatile(X3R_WAITV2(6) : "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]), "+v"(Braw[3 % PIECES]), "+v"(Braw[4 % PIECES]), "+v"(Braw[5 % PIECES]) : [sb] "s"(sb), [n1] "n"(6 + EST), [n2] "n"(6 + EST + NT) : "memory", "scc");
                    else asm volatile(X3R_WAITV2(6) : "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]) : [sb] "s"(sb), [n1] "n"(6 + EST), [n2] "n"(6 + EST + NT) : "memory", "scc");
                } else {
                    if (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]), "+v"(Braw[3 % PIECES]), "+v"(Braw[4 % PIECES]), "+v"(Braw[5 % PIECES]) :: "memory");
                    else asm volatile("s_waitcnt vmcnt(6)" : "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]) :: "memory");
                }
            }
            wait_raw(F[SRC], std::integral_constant<int, MT == 1 ? 6 : 12>{}, first_tag);
            auto one = [&](auto p_tag) __attribute__((always_inline)) {
                constexpr int P = decltype(p_tag)::value, nt = 2 * P + 1;       // (nt: the second column tile of the region)
                static_assert(NT == 8, "the 64-column form needs its own region");
                // block 0: the second column tile's three reads were issued behind the previous tile's block 3, followed by those of
                // the fragments after it (the B stores of this block, issued behind earlier regions, only make the wait stricter)
                constexpr int behind = 3 * (NT - 1 - nt);
                using W = std::integral_constant<int, MT == 0 ? (behind > 15 ? 15 : behind) : -1>;
                region(first_tag, mt_tag, p_tag, src_tag, W{}, bnext, b_wr + stage_next, base, goff);
                if constexpr (MT == 2 && P < 3) step_early(std::integral_constant<int, P + 1>{});      // (the cursor's next position, a piece per gap)
                if constexpr (MT == 3 && P == 3) {
                    // the last region's own column tiles: their fragments of the next K-tile, then the block's fetch
                    X3R_RB(6, bnext)
                    X3R_RB(7, bnext)
                    if (!(NG_X3R_KO & 4)) loadA(F[SRC], base, goff);
                }
                if (MT == 3 && P == 3) { sj ^= 1; step_cursor(); }     // (the cursor's step behind the tile's last MFMAs)
            };
            static_assert(PIECES == 6, "six B pieces per wave and K-tile: two behind each of three regions");
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        X3R_STAMP(5)                                     // (what lies between two tiles: the loop's overhead, an epilogue's tail)
        block(I0{}, I3{}, baseP, goff3P);                // THIS tile's row tile 3 is converted (needed in block 3), refilled from K-tile j + 1
        X3R_STAMP(0)
        block(I1{}, I0{}, baseC, L.goff[0]);              // the NEXT tile's row tile 0 into A[0] (block 0 has issued its last use), K-tile j + 2
        X3R_STAMP(1)
        step_early(std::integral_constant<int, 0>{});
        // the tile's barrier: every wave's pieces of K-tile j + 1 are in LDS (stored in block 0: long done)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        X3R_STAMP(2)
        block(I2{}, I1{}, baseC, L.goff[1]);
        X3R_STAMP(3)
        block(I3{}, I2{}, baseC, L.goff[2]);
        X3R_STAMP(4)
#ifdef NG_X3R_STAMP
        ++st_sum[15];
#endif
    };

    // ---------------- the walk.  In front of K-tile 0 the loader issues what the tile before it would have, in its order, so that the
    // counted waits of the first tile find the operations they assume
    auto keep = [&](const Item& t, Slot& q) { q.p = t.p; q.out = t.out; q.m0 = t.m0; q.n0 = t.n0; q.nk = t.nk; };
    {
        Item t;
        locate(itemN, t);
        if (t.nk < 0) return;
        prepare(t, N);
        keep(t, E);
    }
    L = N;
    left = L.nk;
    // an item is located ONCE: `E` = the item being multiplied, `Q1` = the item behind it, `Q2` = the one behind that, located and prepared
    // (`N`) during E's epilogue
    Slot Q1, Q2;
    auto prepare_next = [&]() {                 // `N` = the item behind the cursor's (past the end: the last state once more, never used)
        Item t;
        locate(itemN + G, t);
        Q2.nk = t.nk;
        if (t.nk >= 0) {
            itemN += G;
            prepare(t, N);
            keep(t, Q2);
        }
    };
    prepare_next();
    Q1 = Q2;
    cursor_bases();                             // the cursor on K-tile 0
    loadA(F[0], baseC, L.goff[0]);
    loadA(F[1], baseC, L.goff[1]);
    loadA(F[2], baseC, L.goff[2]);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) loadB(i);
    if (PIECES == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]), "+v"(Braw[3 % PIECES]), "+v"(Braw[4 % PIECES]), "+v"(Braw[5 % PIECES]),
                                  "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]) :: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]), "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]) :: "memory");
#pragma unroll
    for (int i = 0; i < PIECES; ++i) storeB(0u, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { convert_pair(F[0], A[0], i); convert_pair(F[1], A[1], i); convert_pair(F[2], A[2], i); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const unsigned b0 = ring0 + b_rd;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt == 0) X3R_RB(0, b0) else if (nt == 1) X3R_RB(1, b0) else if (nt == 2) X3R_RB(2, b0) else if (nt == 3) X3R_RB(3, b0)
            else if (nt == 4) X3R_RB(4 % NT, b0) else if (nt == 5) X3R_RB(5 % NT, b0) else if (nt == 6) X3R_RB(6 % NT, b0) else X3R_RB(7 % NT, b0)
        }
    }
    loadA(F[3], baseC, L.goff[3]);              // K-tile 0's row tile 3: converted in block 0 of tile 0
    step_cursor();      // snapshot = K-tile 0, cursor on K-tile 1
#pragma unroll
    for (int i = 0; i < PIECES; ++i) loadB(i);
    loadA(F[0], baseC, L.goff[0]);
    loadA(F[1], baseC, L.goff[1]);
    loadA(F[2], baseC, L.goff[2]);
    step_cursor();      // snapshot = K-tile 1, cursor on K-tile 2: the state tile 0 expects
#ifdef NG_X3R_STAMP
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_last = unsigned(t_); }
#endif
    // (the multiplying side knows of its item only the number of K-tiles; the item itself is located again behind its last K-tile, for the
    // epilogue: nothing of the epilogue's state is live across the K loop)
    int nkC = E.nk;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    while (true) {
        // (the item's first K-tile starts its accumulators: C = 0 in the first product of each -- nothing zeroes them in between)
        tile(std::true_type{});
        stores_behind = 0;
        for (int k = 1; k < nkC; ++k) tile(std::false_type{});
        X3R_STAMP(9)
        X3R_MFMA_DRAIN;                          // the last MFMAs' results are in the accumulators
        if constexpr (GEN) {
            // KIND 2's epilogue is C++ under register pressure: hipcc spills and reloads what it likes, and it believes the raw row sets
            // and B pieces -- the next item's fetches, in flight -- valid.  One drain per item makes them so before it touches anything
            // (the other kinds' epilogues keep no value in scratch; scripts/check_x3_asm.py)
            static_assert(PIECES == 6, "six raw B pieces per wave");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1]),
                         "+v"(Braw[0]), "+v"(Braw[1]), "+v"(Braw[2]), "+v"(Braw[3 % PIECES]), "+v"(Braw[4 % PIECES]), "+v"(Braw[5 % PIECES]) :: "memory");
        }
        Epi e;
        bool full = false;
        if constexpr (KIND == 0) {
            full = E.m0 + 256 <= E.p->M;             // a full tile: every one of its 32 store instructions is issued
            if (full) stage_w(I0{});
        }
        epi_begin(e);
        X3R_STAMP(6)
        if constexpr (!GEN) {
            bool st = false;
            if constexpr (STATS) {
                // the wave's statistics record from the accumulators as they stand -- arithmetic, then NT stores -- in front of the
                // plain epilogue.  (Through the branching C++ path of KIND 2 every slice's stores were followed by reloads of spilled
                // registers, each behind a `s_waitcnt vmcnt(0)` that sat out the stores' acknowledgement: ~30 000 cycles per item
                // against the plain epilogue's 6 000 -- short-K launches with statistics lost on it)
                full = E.m0 + 256 <= e.pM;
                st = e.stats;
                if (st) { stats_acc(e, I0{}); stats_acc(e, I1{}); stats_acc(e, I2{}); stats_acc(e, I3{}); stats_flush(e); }
                if (full) stage_w(I0{});
            }
            if (full) {
                if (E.p->bias != nullptr) epilogue_full(e, std::true_type{}, prepare_next); else epilogue_full(e, std::false_type{}, prepare_next);
                stores_behind = st ? 2 : 1;
            } else {
                slice(e, I0{}, I1{});
                slice(e, I1{}, I1{});
                slice(e, I2{}, I1{});
                slice(e, I3{}, I1{});
                prepare_next();
            }
        } else {
            slice(e, I0{}, I2{});
            slice(e, I1{}, I2{});
            slice(e, I2{}, I2{});
            slice(e, I3{}, I2{});
            epi_finish(e);
            prepare_next();
        }
        X3R_STAMP(7)
#ifdef NG_X3R_STAMP
        ++st_sum[14];
#endif
        // the cursor crossed into Q1 while E was multiplied (two K-tiles ahead, at least three per item), which freed `N`: the item behind
        // Q1 was located and prepared above (inside the full epilogue, otherwise behind it)
        E = Q1;
        Q1 = Q2;
        if (E.nk < 0) break;
        nkC = E.nk;
        X3R_STAMP(8)
    }
#ifdef NG_X3R_STAMP
    {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        st_sum[12] = unsigned(c1 - st_c0);      // shader cycles of the whole kernel
        st_sum[13] = unsigned(r1 - st_r0);      // 100 MHz ticks of the whole kernel
    }
    if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ng_x3r_stamps[(blockIdx.x * 4 + wave) * 16 + i] = st_sum[i];
    }
#endif
#undef X3R_RB
}
#undef X3R_GLD
#undef X3R_WAITV2
#undef X3R_VALU
#undef X3R_DSR

inline bool conv_x3r_generic(const ConvParams& p) { return p.f_y != nullptr; }
inline bool conv_x3r_stats(const ConvParams& p) { return p.stats != nullptr; }
// whether the register-fed tile takes a launch the split tile covers (host): 128-column tiles, asked for by the descriptor (A/B switch)
// (problems whose epilogue leaves statistics / runs the fused pass -- the branching epilogue is not overlapped and not tuned -- only from
// 24 K-tiles on: measured inside the step, profiles/r06_x3r_per_op_ab.txt -- 1.04-1.14 x from 32 K-tiles, 0.68-0.91 x at 7-18)
#ifndef NG_X3R_GEN_MIN_NK
#define NG_X3R_GEN_MIN_NK 24
#endif
inline bool conv_x3r_ok(const ConvParams& p, const int bn) {
    const int nk = p.ntaps * (p.run >> 5);
    return bn == 128 && p.algo == NIRGAN_CONV_X3_R4 && p.OW >= 16 && nk >= (conv_x3r_generic(p) ? NG_X3R_GEN_MIN_NK : 3);
}

}  // namespace ng
