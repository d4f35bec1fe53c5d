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
//   phases       K-tile k (buffer b): .1 read B half 0 + A half 0, C00;  .2 read B half 1, C01;  .3 read A half 1, C10;  .4 C11
//   LDS-DMA      one half-tile (2 pieces per wave) per phase: .1 A1(k+1)  .2 B0(k+2)  .3 A0(k+2)  .4 B1(k+2), then vmcnt(6):
//                everything but the three youngest halves has landed = K-tile k+1 complete, read from the NEXT phase on
//   stagger      waves 4-7 run one barrier behind waves 0-3: on every SIMD one wave is in its MFMA segment while its partner issues
//                fragment reads and DMA pieces
// Hazards (the guide's placement rules): a half is read one phase AFTER the wait that retires its DMA; it is restaged two phases after
// its last fragment read, or one phase after where an lgkmcnt before the reading phase's barrier retired those reads (B half 0: the
// four B reads are issued first and retired by lgkmcnt(8)).
#pragma once
#include "igemm_tiles.h"

namespace ng {

constexpr int T256_HALF = 128 * 128;          // bytes of one half-tile image
constexpr int T256_LDS = 8 * T256_HALF;       // 128 KB

__device__ __forceinline__ void t256_bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void conv_tile256(const ConvParams& p, const int block_id, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int mt256 = (p.M + 255) >> 8, nt256 = (p.N + 255) >> 8;
    const int id = ng_xcd_remap(block_id, mt256 * nt256);
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
            a_boff[h][i] = unsigned(b * p.in_img + oh * p.in_stride * p.in_row + ow * p.in_stride * p.in_cs + p.in_org + lc * 8) * 2u;
            int n = n0 + (row >> 5) * 64 + h * 32 + (row & 31);
            n = n < p.N ? n : 0;                 // (columns past N are never stored and never summed)
            b_boff[h][i] = unsigned(n * p.K + lc * 8) * 2u;
        }
    // K-tiles in slice-major order: 64-channel slice cc of the run, all taps -- a slice of the tile's input patch (one 128-byte line per
    // pixel) stays in L2 across its taps.  The tap offsets sit in one VGPR (lane t holds tap t): a cursor step costs no memory access.
    const int tapv = p.tap_off[lane & (NIRGAN_MAX_TAPS - 1)];
    const int nk = p.ntaps * (p.run >> 6);
    int ct = 0, cc = 0;
    auto advance = [&]() {
        ++ct;
        if (ct == p.ntaps) { ct = 0; cc += 64; }
    };
    const char* const in8 = reinterpret_cast<const char*>(p.in);
    const char* const w8 = reinterpret_cast<const char*>(p.w);
    auto issueA = [&](const int buf, const int h) {
        const int toff = __builtin_amdgcn_readlane(tapv, ct);
        const char* base = ng_uniform_ptr(in8 + (long long)(toff + cc) * 2);
        char* dst = lds + (buf * 4 + h) * T256_HALF + wave * 2048;
        ng_glds16_so(base, a_boff[h][0], dst);
        ng_glds16_so(base, a_boff[h][1], dst + 1024);
    };
    auto issueB = [&](const int buf, const int h) {
        const char* base = ng_uniform_ptr(w8 + (long long)(ct * p.run + cc) * 2);
        char* dst = lds + (buf * 4 + 2 + h) * T256_HALF + wave * 2048;
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto readA = [&](const int buf, const int h) {
        const char* s = lds + (buf * 4 + h) * T256_HALF;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            A[mt][0] = *reinterpret_cast<const bf16x8*>(s + a_rd0 + mt * 2048);
            A[mt][1] = *reinterpret_cast<const bf16x8*>(s + a_rd1 + mt * 2048);
        }
    };
    auto readB = [&](bf16x8 (&Bj)[2][2], const int buf, const int h) {
        const char* s = lds + (buf * 4 + 2 + h) * T256_HALF;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            Bj[nt][0] = *reinterpret_cast<const bf16x8*>(s + b_rd0 + nt * 2048);
            Bj[nt][1] = *reinterpret_cast<const bf16x8*>(s + b_rd1 + nt * 2048);
        }
    };
    auto mma = [&](f32x4 (&c)[4][2], const bf16x8 (&Bj)[2][2]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    c[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt][s], Bj[nt][s], c[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---------------- prologue: K-tile 0 whole, three halves of K-tile 1 (its A half 1 goes out in phase 1)
    issueB(0, 0); issueA(0, 0); issueB(0, 1); issueA(0, 1);
    advance();
    if (nk > 1) {
        issueB(1, 0); issueA(1, 0); issueB(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    t256_bar();                               // every wave's pieces of K-tile 0 have landed
    if (wr == 1) t256_bar();                  // the stagger: waves 4-7 run one barrier behind from here on

    for (int k = 0; k < nk; k += 2) {
        // ======== K-tile k, buffer 0
        // phase 1
        readB(B0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        readA(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < nk) { issueA(1, 1); advance(); }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the four B reads are done: B half 0 may be restaged next phase
        t256_bar();
        mma(acc[0][0], B0);
        t256_bar();
        // phase 2
        readB(B1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < nk) issueB(0, 0);
        t256_bar();
        mma(acc[0][1], B1);
        t256_bar();
        // phase 3
        readA(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < nk) issueA(0, 0);
        t256_bar();
        mma(acc[1][0], B0);
        t256_bar();
        // phase 4
        if (k + 2 < nk) {
            issueB(0, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K-tile k+1 has landed (read from the next phase on)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        t256_bar();
        mma(acc[1][1], B1);
        t256_bar();
        if (k + 1 >= nk) break;
        // ======== K-tile k+1, buffer 1
        // phase 5
        readB(B0, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        readA(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < nk) { issueA(0, 1); advance(); }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        t256_bar();
        mma(acc[0][0], B0);
        t256_bar();
        // phase 6
        readB(B1, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 3 < nk) issueB(1, 0);
        t256_bar();
        mma(acc[0][1], B1);
        t256_bar();
        // phase 7
        readA(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 3 < nk) issueA(1, 0);
        t256_bar();
        mma(acc[1][0], B0);
        t256_bar();
        // phase 8
        if (k + 3 < nk) {
            issueB(1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        t256_bar();
        mma(acc[1][1], B1);
        t256_bar();
    }
    if (wr == 0) t256_bar();                  // waves 0-3 wait for the staggered half: every fragment read and every DMA is done

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
                        const float k0 = __shfl(acc[i][j][0][nt][0], lane & 15, 64);
                        float s1 = 0.f, s2 = 0.f;
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
                        s1 += __shfl_xor(s1, 32, 64);
                        s2 += __shfl_xor(s2, 32, 64);
                        const int col = n0 + wc * 64 + j * 32 + nt * 16 + (lane & 15);
                        if (lane < 16 && col < p.N) {
                            sp[col] = k0;
                            sp[p.N + col] = s1;
                            sp[2 * p.N + col] = s2;
                            sp[3 * p.N + col] = 64.f;
                        }
                    }
            }
        }
    }

    // ---------------- epilogue: each wave transposes its 64 x 64 (row half i) through its OWN 16 KB of LDS -- no workgroup barrier --
    // and stores whole 256-byte (fp32) / 128-byte (bf16) row segments, 16 / 8 bytes per lane
    float* const stg = reinterpret_cast<float*>(lds + wave * 16384);
    const int chunk = lane & 15, lrow = lane >> 4;
    const int n = n0 + wc * 64 + chunk * 4;
    const bool n_ok = n < p.N;                // (host: N % 4 == 0)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && n_ok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    const int OH = p.OHW / p.OW;
    const bool fused = p.f_y != nullptr;
    const float fneg = p.f_act == NIRGAN_ACT_RELU ? 0.f : (p.f_act == NIRGAN_ACT_LRELU ? p.f_slope : 1.f);
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
        const int mbase = m0 + i * 128 + wr * 64;
        int m = mbase + lrow;
        const int mc = m < p.M ? m : p.M - 1;
        int b = mc / p.OHW;
        const int r0 = mc - b * p.OHW;
        int oh = r0 / p.OW, ow = r0 - oh * p.OW;
        const int fb = (mbase < p.M ? mbase : p.M - 1) / p.OHW;          // (fused: one sample per 128-row half, host: OH*OW % 128 == 0)
        f32x4 fm = {0.f, 0.f, 0.f, 0.f}, fr = {1.f, 1.f, 1.f, 1.f}, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (fused && n_ok) {
            fm = *reinterpret_cast<const f32x4*>(p.f_mean + size_t(fb) * p.N + n);
            fr = *reinterpret_cast<const f32x4*>(p.f_rstd + size_t(fb) * p.N + n);
        }
#pragma unroll 4
        for (int pass = 0; pass < 16; ++pass) {
            if (m < p.M && n_ok) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + (pass * 4 + lrow) * 64 + chunk * 4);
                v += bv;
                const int oidx = b * p.out_img + oh * p.out_stride * p.out_row + ow * p.out_stride * p.out_cs + p.out_org + n;
                if (p.out16) {
                    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(p.out) + oidx) = __builtin_convertvector(v, bf16x4_t);
                } else {
                    *reinterpret_cast<f32x4*>(p.out + oidx) = v;
                }
                if (fused) {
                    const size_t yidx = size_t(b) * p.f_img + size_t(oh * p.out_stride) * p.f_row + size_t(ow * p.out_stride) * p.N + p.f_org + n;
                    f32x4 y4;
                    if (p.f_y16) {
                        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                        y4 = __builtin_convertvector(*reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const unsigned short*>(p.f_y) + yidx), f32x4);
                    } else {
                        y4 = *reinterpret_cast<const f32x4*>(p.f_y + yidx);
                    }
                    const f32x4 z = (y4 - fm) * fr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float gz = z[q] > 0.f ? v[q] : v[q] * fneg;
                        s1[q] += gz;
                        s2[q] += gz * z[q];
                    }
                }
            }
            m += 4;
            ow += 4;
            while (ow >= p.OW) { ow -= p.OW; ++oh; }
            while (oh >= OH) { oh -= OH; ++b; }
        }
        if (fused) {
            // first pass of the consumer layer's instance-norm backward: this wave's 64 rows, then the two waves of a 128-row chunk
            // (wr = 0, 1) join through LDS in a fixed order.  All of a wave's staging reads are done (same wave, program order).
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1[q] += __shfl_xor(s1[q], 16, 64);
                s2[q] += __shfl_xor(s2[q], 16, 64);
                s1[q] += __shfl_xor(s1[q], 32, 64);
                s2[q] += __shfl_xor(s2[q], 32, 64);
            }
            t256_bar();                                             // (uniform: `fused` is a launch constant) every wave's staging reads are done
            f32x4* const red = reinterpret_cast<f32x4*>(lds);       // 8 waves x 16 lanes x 2 x 16 B = 4 KB
            if (lane < 16) {
                red[(wave * 16 + chunk) * 2] = s1;
                red[(wave * 16 + chunk) * 2 + 1] = s2;
            }
            t256_bar();
            if (wr == 0 && lane < 16 && n_ok && mbase < p.M) {
                const f32x4 t1 = s1 + red[((wave + 4) * 16 + chunk) * 2], t2 = s2 + red[((wave + 4) * 16 + chunk) * 2 + 1];
                float* pp = p.f_part + (size_t(fb) * p.f_cps + p.f_chunk0 + ((m0 + i * 128 - fb * p.OHW) >> 7)) * 2 * p.N + n;
                *reinterpret_cast<f32x4*>(pp) = t1;
                *reinterpret_cast<f32x4*>(pp + p.N) = t2;
            }
            t256_bar();
        }
    }
}

// whether the 256-wide tile covers a problem (host): both operands stored as bf16, whole 64-channel slices, whole 256-column tiles,
// enough tiles to give most CUs one, 32-bit offsets
inline bool conv_tile256_ok(const ConvParams& p) {
    if (!(p.prec == 1 && p.in_bf16 && p.w_bf16 && p.off32 && p.ksplit == 1)) return false;
    if (p.run % 64 != 0 || p.N % 256 != 0) return false;
    const long long tiles = (long long)((p.M + 255) >> 8) * (p.N >> 8);
    return tiles >= 128;
}

}  // namespace ng
