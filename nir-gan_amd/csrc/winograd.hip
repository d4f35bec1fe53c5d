// Winograd F(2x2, 3x3) forward of a stride-1 3x3 convolution on the fp32 matrix pipe (exact fp32 products, 2.25x fewer
// of them than the direct contraction).  Three launches:
//   wino_weight_kernel   U[f][k][c] = (G g G^T)[f]            per layer when the weights changed (16 x Cout x Cin)
//   wino_input_kernel    V[f][t][c] = (B^T d B)[f]            from the halo'd NHWC input (halo 1, written by the producer)
//   wino_gemm_kernel     Y[t][a,b][k] = sum_f A^T[a][f1] A^T[b][f2] (sum_c V[f][t][c] U[f][k][c])  (+ bias)
// with f = 4 f1 + f2, t = (image, tile row, tile column) over 2x2 output tiles.  The GEMM walks the 16 frequencies of its
// (64 tiles x 128 channels) block in ONE K loop (stage s: f = s / (C/32)); after each frequency the partial product M is
// added with its +-1 coefficients into the four output accumulators and cleared, so the transform-domain product never
// leaves the registers (a separate output transform would write and re-read 4x the activation bytes).
// Same staging as the direct tile: LDS-DMA of 8-row x 128-B pieces, chunk ^ ((row>>1)&7) swizzle, two stages.
#include "common.h"

namespace {

struct WinoW { const float* w; float* U; int K, C; };

// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void wino_weight_kernel(const WinoW p) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)p.K * p.C) return;
    const int k = int(i / p.C), c = int(i - (long long)k * p.C);
    const float* g = p.w + (size_t(k) * p.C + c) * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0;
        t[1][j] = 0.5f * (g0 + g1 + g2);
        t[2][j] = 0.5f * (g0 - g1 + g2);
        t[3][j] = g2;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
        const size_t plane = size_t(p.K) * p.C;
        float* U = p.U + size_t(k) * p.C + c;
        U[(a * 4 + 0) * plane] = u0;
        U[(a * 4 + 1) * plane] = u1;
        U[(a * 4 + 2) * plane] = u2;
        U[(a * 4 + 3) * plane] = u3;
    }
}

struct WinoIn { const float* x; float* V; int B, H, W, C, x_row, x_img, TH, TW; long long T; };

// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]; one thread = one tile x 4 channels
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoIn p) {
    const int q4 = p.C / 4;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const float* src = p.x + size_t(b) * p.x_img + size_t(2 * ty) * p.x_row + size_t(2 * tx) * p.C + q * 4;
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[a][c] = *reinterpret_cast<const f32x4*>(src + size_t(a) * p.x_row + size_t(c) * p.C);
    f32x4 m[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m[0][c] = d[0][c] - d[2][c];
        m[1][c] = d[1][c] + d[2][c];
        m[2][c] = d[2][c] - d[1][c];
        m[3][c] = d[1][c] - d[3][c];
    }
    const size_t plane = size_t(p.T) * p.C;
    float* V = p.V + size_t(t) * p.C + q * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        *reinterpret_cast<f32x4*>(V + (a * 4 + 0) * plane) = m[a][0] - m[a][2];
        *reinterpret_cast<f32x4*>(V + (a * 4 + 1) * plane) = m[a][1] + m[a][2];
        *reinterpret_cast<f32x4*>(V + (a * 4 + 2) * plane) = m[a][2] - m[a][1];
        *reinterpret_cast<f32x4*>(V + (a * 4 + 3) * plane) = m[a][1] - m[a][3];
    }
}

struct WinoG {
    const float* V; const float* U; const float* bias; float* y; const float* zero;
    int T, C, K, TH, TW, H, W;
    int mtiles, ntiles;
};

// A^T = [[1,1,1,0],[0,1,-1,-1]]: coefficient of frequency f1 in output row a
__device__ __forceinline__ int at_coef(int a, int f) { return a == 0 ? (f < 3 ? 1 : 0) : (f == 0 ? 0 : (f == 1 ? 1 : -1)); }

__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const WinoG p) {
    constexpr int BM = 64, BN = 128;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;      // 8 KB + 16 KB
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE > BM * BN * 4 ? 2 * STAGE : BM * BN * 4];
    char* st0 = lds;
    char* st1 = lds + STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = ng_xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int n0 = (id % p.ntiles) * BN, m0 = (id / p.ntiles) * BM;

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
    };
    // fold the finished frequency f into the four outputs (coefficients 0, +1, -1) and clear the product
    auto fold = [&](int f) {
        const int f1 = f >> 2, f2 = f & 3;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int cf = at_coef(a, f1) * at_coef(b, f2);
                if (cf != 0) {
                    const float s = float(cf);
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
    const int nk = 16 * csteps;
    int f_i = 0, c_i = 0;             // coordinates of the step being ISSUED
    int f_c = 0, c_c = 0;             // coordinates of the step being COMPUTED
    auto next = [&](int& f, int& c) {
        c += 32;
        if (c >= p.C) { c = 0; ++f; }
    };
    issue(st0, 0, 0);
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
    if (p.bias != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = n + j < p.K ? p.bias[n + j] : 0.f;
    }
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
                float* dst = p.y + ((size_t(b) * p.H + 2 * ty + oa) * p.W + 2 * tx + ob) * p.K + n;
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

}  // namespace

extern "C" int64_t nirgan_wino_ws_elems(int B, int H, int W, int C, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || K <= 0) return 0;
    return 16ll * B * (H / 2) * (W / 2) * C + 16ll * K * C;
}

extern "C" int nirgan_wino_weights(const float* w, int K, int C, float* U, void* stream) {
    NG_REQUIRE(w && U && K > 0 && C > 0, "wino_weights: bad arguments");
    WinoW p{w, U, K, C};
    const long long n = (long long)K * C;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino_weights");
}

extern "C" int nirgan_wino_conv3x3(const nirgan_wino_desc* d, void* stream) {
    NG_REQUIRE(d && d->x && d->U && d->V && d->y && d->zero_page, "wino_conv3x3: null pointer");
    NG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && !(d->H & 1) && !(d->W & 1), "wino_conv3x3: H and W must be even (H=%d W=%d)", d->H, d->W);
    NG_REQUIRE(d->C % 32 == 0 && d->C > 0 && d->K > 0 && d->K % 128 == 0, "wino_conv3x3: C %% 32 == 0 and K %% 128 == 0 (C=%d K=%d)", d->C, d->K);
    NG_REQUIRE(d->x_hp == d->H + 2 && d->x_wp == d->W + 2, "wino_conv3x3: the input must carry a halo of exactly 1 (%dx%d for %dx%d)", d->x_hp, d->x_wp, d->H, d->W);
    NG_REQUIRE(ng_aligned16(d->x) && ng_aligned16(d->U) && ng_aligned16(d->V) && ng_aligned16(d->y) && ng_aligned16(d->zero_page), "wino_conv3x3: pointers must be 16-byte aligned");
    const long long T = (long long)d->B * (d->H / 2) * (d->W / 2);
    NG_REQUIRE(16 * T * d->C < (1ll << 31) * 4 && T < (1ll << 31) / d->C, "wino_conv3x3: problem too large for 32-bit tile offsets");
    NG_REQUIRE(d->V_elems >= 16 * T * d->C, "wino_conv3x3: V workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    WinoIn in;
    in.x = d->x; in.V = d->V; in.B = d->B; in.H = d->H; in.W = d->W; in.C = d->C;
    in.x_row = d->x_wp * d->C; in.x_img = d->x_hp * in.x_row; in.TH = d->H / 2; in.TW = d->W / 2; in.T = T;
    const long long nthreads = T * (d->C / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3(unsigned((nthreads + 255) / 256)), dim3(256), 0, st, in);
    WinoG g;
    g.V = d->V; g.U = d->U; g.bias = d->bias; g.y = d->y; g.zero = d->zero_page;
    g.T = int(T); g.C = d->C; g.K = d->K; g.TH = d->H / 2; g.TW = d->W / 2; g.H = d->H; g.W = d->W;
    g.mtiles = int((T + 63) / 64); g.ntiles = d->K / 128;
    hipLaunchKernelGGL(wino_gemm_kernel, dim3(g.mtiles * g.ntiles), dim3(256), 0, st, g);
    return nirgan_check_launch("wino_conv3x3");
}
