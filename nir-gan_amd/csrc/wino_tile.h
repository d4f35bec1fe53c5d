// The frequency-folding GEMM tile of the Winograd F(2x2, 3x3) convolution (see winograd.hip), shared by the stand-alone
// kernel and the launch that fuses it horizontally with a weight-gradient problem.
#pragma once
#include "common.h"

namespace ng {

struct WinoG {
    const float* V; const float* U; const float* bias; float* y; const float* zero;
    int T, C, K, TH, TW, H, W;
    int mtiles, ntiles;
    int a;                       // frequencies per dimension: 4 = F(2x2,3x3), 5 = F(2x2,4x4)
    int fsplit;                  // workgroups per tile, each over a contiguous range of the a*a frequencies (1 = all)
    float* ws; long long ws_stride;   // fsplit > 1: partial outputs [split][B*H*W*K]
};

// coefficient of frequency f in output row `row` of A^T: [[1,1,1,0],[0,1,-1,-1]] for F(2,3), [[1,1,1,1,0],[0,1,-1,-1/2,1]] for F(2,4)
__device__ __forceinline__ float at_coef(int row, int f, int a) {
    if (row == 0) return f < a - 1 ? 1.f : 0.f;
    if (f == 0) return 0.f;
    if (f == 1) return 1.f;
    if (f == 2) return -1.f;
    if (a == 4) return -1.f;                     // F(2,3): f == 3
    return f == 3 ? -0.5f : 1.f;                 // F(2,4): f == 3, 4
}

constexpr int WINO_LDS_BYTES = 2 * (64 * 128 + 128 * 128);      // two stages of (64 tile rows + 128 channel rows) x 128 B = 48 KB

// one (64 tiles x 128 channels) block of the frequency-folding GEMM; lds: WINO_LDS_BYTES, 16-byte aligned
__device__ __forceinline__ void wino_tile(const WinoG& p, const int block_id, char* lds) {
    constexpr int BM = 64, BN = 128;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;      // 8 KB + 16 KB
    char* st0 = lds;
    char* st1 = lds + STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.mtiles * p.ntiles;
    const int rid = ng_xcd_remap(block_id, tiles * p.fsplit);
    const int sp = rid / tiles, id = rid - sp * tiles;
    const int n0 = (id % p.ntiles) * BN, m0 = (id / p.ntiles) * BM;
    const int nf = p.a * p.a;
    const int f_begin = sp * nf / p.fsplit, f_end = (sp + 1) * nf / p.fsplit;     // this workgroup's frequencies

    // ---------------- loader: wave w owns A pieces 2w, 2w+1 (8 tile rows each) and B pieces 4w .. 4w+3
    const int lrow = lane >> 3, lchunk = lane & 7;
    int a_base[2], b_base[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + lrow;
        const int lc = lchunk ^ ((row >> 1) & 7);
        int t = m0 + row;
        t = t < p.T ? t : p.T - 1;
        a_base[i] = t * p.C + lc * 4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        const int lc = lchunk ^ ((row >> 1) & 7);
        b_base[i] = (n0 + row) * p.C + lc * 4;
    }
    const size_t a_plane = size_t(p.T) * p.C, b_plane = size_t(p.K) * p.C;
    const int csteps = p.C >> 5;
    auto issue = [&](char* sA, int f, int c0) {
        char* sB = sA + A_BYTES;
        const float* Vf = p.V + f * a_plane + c0;
        const float* Uf = p.U + f * b_plane + c0;
#pragma unroll
        for (int i = 0; i < 2; ++i) ng_glds16(Vf + a_base[i], sA + (wave * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) ng_glds16(Uf + b_base[i], sB + (wave * 4 + i) * 1024);
    };

    // ---------------- compute: wave (wr, wc) = 32 tiles x 64 channels
    const int wr = wave >> 1, wc = wave & 1, half = lane >> 5;
    const int arow = wr * 32 + (lane & 31);
    const int a_off = arow * 128, a_key = (arow >> 1) & 7;
    int b_off[2], b_key[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = wc * 64 + nt * 32 + (lane & 31);
        b_off[nt] = row * 128;
        b_key[nt] = (row >> 1) & 7;
    }
    f32x16 M[2], Y[4][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) M[nt][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[j][nt][r] = 0.f;
    }
    auto compute = [&](const char* sA) {
        const char* sB = sA + A_BYTES;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + half;
            const f32x4 a = *reinterpret_cast<const f32x4*>(sA + a_off + ((chunk ^ a_key) << 4));
            f32x4 b[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(sB + b_off[nt] + ((chunk ^ b_key[nt]) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) M[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[nt][j], M[nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // fold the finished frequency f into the four outputs (coefficients 0, +-1; also -1/2, 1/4 for the 4x4 filter) and clear the product
    auto fold = [&](int f) {
        const int f1 = f / p.a, f2 = f - f1 * p.a;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float s = at_coef(a, f1, p.a) * at_coef(b, f2, p.a);
                if (s != 0.f) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) Y[a * 2 + b][nt] += s * M[nt];
                }
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) M[nt][r] = 0.f;
    };

    // ---------------- one K loop over (frequency, 32-channel slice); stage parity = step parity
    const int nk = (f_end - f_begin) * csteps;
    int f_i = f_begin, c_i = 0;       // coordinates of the step being ISSUED
    int f_c = f_begin, c_c = 0;       // coordinates of the step being COMPUTED
    auto next = [&](int& f, int& c) {
        c += 32;
        if (c >= p.C) { c = 0; ++f; }
    };
    issue(st0, f_begin, 0);
    for (int s = 0; s < nk; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) {
            next(f_i, c_i);
            issue((s & 1) ? st0 : st1, f_i, c_i);
        }
        compute((s & 1) ? st1 : st0);
        const bool last_of_f = c_c + 32 >= p.C;
        if (last_of_f) fold(f_c);
        next(f_c, c_c);
    }

    // ---------------- epilogue: four rounds, one output position (a, b) of the 2x2 tile each, through LDS (64 x 128 floats)
    float* buf = reinterpret_cast<float*>(lds);
    const int chunk = tid & 31, row0 = tid >> 5;             // 32 lanes x float4 per tile row, 8 rows per pass
    const int n = n0 + chunk * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && p.fsplit == 1) {               // split: the reduce pass adds the bias
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = n + j < p.K ? p.bias[n + j] : 0.f;
    }
    float* ybase = p.fsplit == 1 ? p.y : p.ws + sp * p.ws_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __syncthreads();                                    // K loop / previous round done with the buffer
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = wc * 64 + nt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BN + col] = Y[j][nt][r];
        }
        __syncthreads();
        const int oa = j >> 1, ob = j & 1;
        for (int row = row0; row < BM; row += 8) {
            const int t = m0 + row;
            if (t < p.T && n < p.K) {
                const int tx = t % p.TW;
                const int r2 = t / p.TW;
                const int ty = r2 % p.TH, b = r2 / p.TH;
                f32x4 v = *reinterpret_cast<const f32x4*>(buf + row * BN + chunk * 4);
                v += bv;
                if (2 * ty + oa >= p.H || 2 * tx + ob >= p.W) continue;        // odd extent: the last half tile has no pixel there
                float* dst = ybase + ((size_t(b) * p.H + 2 * ty + oa) * p.W + 2 * tx + ob) * p.K + n;
                if (n + 4 <= p.K) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < p.K) dst[q] = v[q];
                }
            }
        }
    }
}


// descriptor -> parameters of the GEMM stage (validation lives in winograd.hip::check_wino)
inline void build_wino_params(const nirgan_wino_desc* d, WinoG& g) {
    const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    g.V = d->V; g.U = d->U; g.bias = d->bias; g.y = d->y; g.zero = d->zero_page;
    g.T = int(T); g.C = d->C; g.K = d->K; g.TH = (d->H + 1) / 2; g.TW = (d->W + 1) / 2; g.H = d->H; g.W = d->W;
    g.mtiles = int((T + 63) / 64); g.ntiles = d->K / 128;
    g.a = d->r == 4 ? 5 : 4;
    g.fsplit = d->fsplit > 1 ? d->fsplit : 1;
    g.ws = d->split_ws; g.ws_stride = (long long)d->B * d->H * d->W * d->K;
}

}  // namespace ng

int ng_wino_gemm_params(const nirgan_wino_desc* d, ng::WinoG* g);   // winograd.hip: validation + parameters
