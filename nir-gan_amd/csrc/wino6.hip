// Winograd F(4x4, 3x3) for the stride-1 3x3 convolutions of the ResnetBlocks (model/networks.py:405-427) in exact-fp32 mode:
// 36 products per 4x4 outputs instead of 144 (F(2x2,3x3): 64), i.e. 1.78x fewer matrix-pipe cycles than the F(2x2,3x3) path of
// winograd.hip and 4x fewer than the direct contraction.  16 output accumulators per tile do not fit the register file next to the
// product, so the transform-domain product goes through HBM once:
//   wino6_weight      U[f][k][c] = (G g G^T)[f]                      per layer when the weights changed (36 x Cout x Cin)
//   wino6_input       V[f][t][c] = (B^T d B)[f]                      6x6 patches (stride 4) of the halo'd NHWC input; HBM-bound
//   wino6_gemm        M[f][t][k] = sum_c V[f][t][c] U[f][k][c]       36 independent [T x C] x [C x K] GEMMs in ONE grid on the direct
//                                                                    128 x 128 tile (conv_tile: LDS-DMA staging, v_mfma_f32_32x32x2_f32)
//   wino6_output      y[4ty+i][4tx+j][k] = (A^T M A)[i][j] + bias    HBM-bound
// with f = 6 f1 + f2 and t = (image, tile row, tile column) over 4x4 output tiles.  The transform-domain tensors are 2.25 floats per
// input element (F(2x2,3x3): 4), so the two extra passes move fewer bytes than the F(2x2) input transform alone did.
// Data gradient: the same three stages on dY (zero halo 2) with the flipped, transposed filter over the padded input extent.
// Weight gradient in the transform domain: dU[f][k][c] = sum_t Yt[f][t][k] V[f][t][c] with Yt = A dY A^T (36 weight-gradient
// problems as ONE nirgan_wgrad_igemm launch with nplanes = 36, V kept from the forward), then dW = G^T dU G.
//
// Cook-Toom over the points 0, 1, -1, 2, -1/2, inf (fp32 error of the forward 3.8e-6 of the output's maximum against 5e-7 for a
// sequential direct sum and 1.1e-5 for the textbook points 0, +-1, +-2, inf; scripts/exp_wino6_points.py), rows scaled so that B^T
// (applied to the activations) is integer:
//   B^T = [[2,3,-4,-3,2,0],[0,2,5,1,-2,0],[0,2,1,-5,2,0],[0,-1,-2,1,2,0],[0,-2,1,2,-1,0],[0,2,3,-4,-3,2]]
//   G   = [[1/2,0,0],[1/6,1/6,1/6],[1/6,-1/6,1/6],[1/30,1/15,2/15],[16/15,-8/15,4/15],[0,0,1/2]]
//   A^T = [[1,1,1,1,1,0],[0,1,-1,2,-1/2,0],[0,1,1,4,1/4,0],[0,1,-1,8,-1/8,1]]
#include "igemm_tiles.h"
#include "igemm_tile256.h"
#include "igemm_x3.h"
#include "instnorm_dev.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------------ 1-D transforms
template <class T> __device__ __forceinline__ void w6_bt(const T* d, T* t) {            // t = B^T d
    t[0] = 2.f * (d[0] + d[4]) + 3.f * (d[1] - d[3]) - 4.f * d[2];
    t[1] = 2.f * (d[1] - d[4]) + 5.f * d[2] + d[3];
    t[2] = 2.f * (d[1] + d[4]) + d[2] - 5.f * d[3];
    t[3] = 2.f * (d[4] - d[2]) + (d[3] - d[1]);
    t[4] = 2.f * (d[3] - d[1]) + (d[2] - d[4]);
    t[5] = 2.f * (d[1] + d[5]) + 3.f * (d[2] - d[4]) - 4.f * d[3];
}
template <class T> __device__ __forceinline__ void w6_at(const T* m, T* y) {            // y = A^T m  (6 -> 4)
    const T s12 = m[1] + m[2], d12 = m[1] - m[2];
    y[0] = m[0] + s12 + m[3] + m[4];
    y[1] = d12 + 2.f * m[3] - 0.5f * m[4];
    y[2] = s12 + 4.f * m[3] + 0.25f * m[4];
    y[3] = d12 + 8.f * m[3] - 0.125f * m[4] + m[5];
}
template <class T> __device__ __forceinline__ void w6_a(const T* e, T* u) {             // u = A e  (4 -> 6)
    u[0] = e[0];
    u[1] = (e[0] + e[2]) + (e[1] + e[3]);
    u[2] = (e[0] + e[2]) - (e[1] + e[3]);
    u[3] = e[0] + 2.f * e[1] + 4.f * e[2] + 8.f * e[3];
    u[4] = e[0] - 0.5f * e[1] + 0.25f * e[2] - 0.125f * e[3];
    u[5] = e[3];
}
__device__ __forceinline__ void w6_g(const float* g, float* w) {                        // w = G g  (3 -> 6)
    w[0] = 0.5f * g[0];
    w[1] = (g[0] + g[1] + g[2]) * (1.f / 6.f);
    w[2] = (g[0] - g[1] + g[2]) * (1.f / 6.f);
    w[3] = g[0] * (1.f / 30.f) + g[1] * (1.f / 15.f) + g[2] * (2.f / 15.f);
    w[4] = g[0] * (16.f / 15.f) - g[1] * (8.f / 15.f) + g[2] * (4.f / 15.f);
    w[5] = 0.5f * g[2];
}
__device__ __forceinline__ void w6_gt(const float* u, float* g) {                       // g = G^T u  (6 -> 3)
    g[0] = 0.5f * u[0] + (u[1] + u[2]) * (1.f / 6.f) + u[3] * (1.f / 30.f) + u[4] * (16.f / 15.f);
    g[1] = (u[1] - u[2]) * (1.f / 6.f) + u[3] * (1.f / 15.f) - u[4] * (8.f / 15.f);
    g[2] = (u[1] + u[2]) * (1.f / 6.f) + u[3] * (2.f / 15.f) + u[4] * (4.f / 15.f) + 0.5f * u[5];
}


// F(4x4, 4x4) -- nn.Conv2d(C, K, 4, stride 1) of the PatchGAN (model/networks.py:573-579): 49 products per 4x4 outputs (F(2x2,4x4): 100,
// direct: 256).  Cook-Toom over 0, 1, -1, 2, -2, 1/2, inf (fp32 error of the forward 1.0e-5 of the output's maximum, weight gradient
// 5e-6: scripts/exp_wino6_points.py), rows scaled so that B^T is integer.
__host__ __device__ constexpr float w7_BT(int i, int j) {
    constexpr float m[7][7] = {{4, -8, -5, 10, 1, -2, 0}, {0, -4, 4, 9, -1, -2, 0}, {0, -4, 12, -7, -3, 2, 0}, {0, 2, -3, -4, 3, 2, 0},
                               {0, 2, -5, 0, 5, -2, 0}, {0, 4, 0, -5, 0, 1, 0}, {0, -4, 8, 5, -10, -1, 2}};
    return m[i][j];
}
__host__ __device__ constexpr float w7_G(int i, int j) {
    constexpr float m[7][4] = {{1.f / 4, 0, 0, 0}, {1.f / 6, 1.f / 6, 1.f / 6, 1.f / 6}, {1.f / 18, -1.f / 18, 1.f / 18, -1.f / 18},
                               {1.f / 72, 1.f / 36, 1.f / 18, 1.f / 9}, {1.f / 120, -1.f / 60, 1.f / 30, -1.f / 15},
                               {32.f / 45, 16.f / 45, 8.f / 45, 4.f / 45}, {0, 0, 0, 1.f / 2}};
    return m[i][j];
}
__host__ __device__ constexpr float w7_AT(int i, int j) {
    constexpr float m[4][7] = {{1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0.5f, 0}, {0, 1, 1, 4, 4, 0.25f, 0}, {0, 1, -1, 8, -8, 0.125f, 1}};
    return m[i][j];
}
// F(6x6, 3x3): 64 products per 6x6 outputs (1.78 per output; F(4x4,3x3): 2.25, direct: 9).  Cook-Toom over 0, 1, -1, 2, -2, 1/2, -1/2, inf,
// rows scaled so that B^T is integer; fp32 error of the forward 1.7e-5 of the output's maximum, weight gradient 4e-6
// (scripts/exp_wino6_points.py) -- the level of F(4x4,4x4) above.
__host__ __device__ constexpr float w8_BT(int i, int j) {
    constexpr float m[8][8] = {{4, 0, -21, 0, 21, 0, -4, 0}, {0, -4, -4, 17, 17, -4, -4, 0}, {0, 4, -4, -17, 17, 4, -4, 0}, {0, 2, 1, -10, -5, 8, 4, 0},
                               {0, -2, 1, 10, -5, -8, 4, 0}, {0, 4, 8, -5, -10, 1, 2, 0}, {0, -4, 8, 5, -10, -1, 2, 0}, {0, -4, 0, 21, 0, -21, 0, 4}};
    return m[i][j];
}
__host__ __device__ constexpr float w8_G(int i, int j) {
    constexpr float m[8][3] = {{1.f / 4, 0, 0}, {1.f / 18, 1.f / 18, 1.f / 18}, {1.f / 18, -1.f / 18, 1.f / 18}, {1.f / 360, 1.f / 180, 1.f / 90},
                               {1.f / 360, -1.f / 180, 1.f / 90}, {16.f / 45, 8.f / 45, 4.f / 45}, {16.f / 45, -8.f / 45, 4.f / 45}, {0, 0, 1.f / 4}};
    return m[i][j];
}
__host__ __device__ constexpr float w8_AT(int i, int j) {
    constexpr float m[6][8] = {{1, 1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 1.f / 2, -1.f / 2, 0}, {0, 1, 1, 4, 4, 1.f / 4, 1.f / 4, 0},
                               {0, 1, -1, 8, -8, 1.f / 8, -1.f / 8, 0}, {0, 1, 1, 16, 16, 1.f / 16, 1.f / 16, 0}, {0, 1, -1, 32, -32, 1.f / 32, -1.f / 32, 1}};
    return m[i][j];
}
// acc += c * x with the constant folded after unrolling (0: nothing, +-1: add / subtract)
template <typename T> __device__ __forceinline__ void w6_mac(T& acc, const float c, const T& x) {
    if (c == 0.f) return;
    if (c == 1.f) acc += x;
    else if (c == -1.f) acc -= x;
    else acc += c * x;
}

// V = variant: 3 = F(4x4,3x3), 4 = F(4x4,4x4), 6 = F(6x6,3x3).  R = filter size, MO x MO outputs per tile, N = MO + R - 1 points per
// dimension, N * N planes, patches of N x N at stride MO.
template <int V> struct W6 {
    static constexpr int R = V == 6 ? 3 : V, MO = V == 6 ? 6 : 4, N = MO + R - 1, NP = N * N;
    static __host__ __device__ constexpr float cBT(int i, int j) { return V == 4 ? w7_BT(i, j) : w8_BT(i, j); }
    static __host__ __device__ constexpr float cG(int i, int j) { return V == 4 ? w7_G(i, j) : w8_G(i, j); }
    static __host__ __device__ constexpr float cAT(int i, int j) { return V == 4 ? w7_AT(i, j) : w8_AT(i, j); }
    template <class T> static __device__ __forceinline__ void bt(const T* d, T* t) {                // t = B^T d   (N -> N)
        if constexpr (V == 3) w6_bt(d, t);
        else {
#pragma unroll
            for (int o = 0; o < N; ++o) {
                T s = d[0] * 0.f;
#pragma unroll
                for (int i = 0; i < N; ++i) w6_mac(s, cBT(o, i), d[i]);
                t[o] = s;
            }
        }
    }
    template <class T> static __device__ __forceinline__ void at(const T* m, T* y) {                // y = A^T m   (N -> MO)
        if constexpr (V == 3) w6_at(m, y);
        else {
#pragma unroll
            for (int o = 0; o < MO; ++o) {
                T s = m[0] * 0.f;
#pragma unroll
                for (int i = 0; i < N; ++i) w6_mac(s, cAT(o, i), m[i]);
                y[o] = s;
            }
        }
    }
    template <class T> static __device__ __forceinline__ void a(const T* e, T* u) {                 // u = A e     (MO -> N)
        if constexpr (V == 3) w6_a(e, u);
        else {
#pragma unroll
            for (int o = 0; o < N; ++o) {
                T s = e[0] * 0.f;
#pragma unroll
                for (int i = 0; i < MO; ++i) w6_mac(s, cAT(i, o), e[i]);
                u[o] = s;
            }
        }
    }
    static __device__ __forceinline__ void g(const float* gg, float* w) {                           // w = G g     (R -> N)
        if constexpr (V == 3) w6_g(gg, w);
        else {
#pragma unroll
            for (int o = 0; o < N; ++o) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < R; ++i) w6_mac(s, cG(o, i), gg[i]);
                w[o] = s;
            }
        }
    }
    static __device__ __forceinline__ void gt(const float* u, float* gg) {                          // g = G^T u   (N -> R)
        if constexpr (V == 3) w6_gt(u, gg);
        else {
#pragma unroll
            for (int o = 0; o < R; ++o) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < N; ++i) w6_mac(s, cG(i, o), u[i]);
                gg[o] = s;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ weights
struct W6W { const float* w; float* U; int K, C, flip; unsigned short* U3; };       // U3: the three bf16 planes of U (precision 3), or NULL

template <int V>
__device__ __forceinline__ void wino6_weight_one(const W6W& p, const long long i) {
    constexpr int N = W6<V>::N, R = W6<V>::R, RR = R * R;
    if (i >= (long long)p.K * p.C) return;
    const int k = int(i / p.C), c = int(i - (long long)k * p.C);
    // flip: the data-gradient filter g'[k][c][i][j] = W[c][k][R-1-i][R-1-j] (W stored [C][K][R][R]: rows are the forward OUTPUT channels)
    const float* g = p.flip ? p.w + (size_t(c) * p.K + k) * RR : p.w + (size_t(k) * p.C + c) * RR;
    float t[N][R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        float col[R], o[N];
#pragma unroll
        for (int a = 0; a < R; ++a) col[a] = p.flip ? g[RR - 1 - (a * R + j)] : g[a * R + j];
        W6<V>::g(col, o);
#pragma unroll
        for (int a = 0; a < N; ++a) t[a][j] = o[a];
    }
    const size_t plane = size_t(p.K) * p.C;
    float* U = p.U + size_t(k) * p.C + c;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        float o[N];
        W6<V>::g(t[a], o);
        if (p.U != nullptr) {
#pragma unroll
            for (int b = 0; b < N; ++b) U[(a * N + b) * plane] = o[b];
        }
        if (p.U3 != nullptr) {
            // the same value as three bf16 terms h + m + l (nirgan_split3's rule) for the three-term split plane GEMMs: planes N*N*K*C apart
            unsigned short* U3 = p.U3 + size_t(k) * p.C + c;
            const size_t term = size_t(N) * N * plane;
#pragma unroll
            for (int b = 0; b < N; ++b) {
                const unsigned h = ng::x3_pk(o[b], 0.f) & 0xffffu;
                const float r1 = o[b] - __builtin_bit_cast(float, h << 16);
                const unsigned m = ng::x3_pk(r1, 0.f) & 0xffffu;
                const unsigned l = ng::x3_pk(r1 - __builtin_bit_cast(float, m << 16), 0.f) & 0xffffu;
                U3[(a * N + b) * plane] = (unsigned short)h;
                U3[term + (a * N + b) * plane] = (unsigned short)m;
                U3[2 * term + (a * N + b) * plane] = (unsigned short)l;
            }
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void wino6_weight_kernel(const W6W p) { wino6_weight_one<V>(p, blockIdx.x * 256ll + threadIdx.x); }

// all weight transforms of a step in one launch: 8 x int64 per job {w, U, K, C, flip, first_block, variant, U3 (three bf16 planes of U, or 0)}
__global__ __launch_bounds__(256) void wino6_weights_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    for (int i = 1; i < njobs; ++i)
        if (int(blockIdx.x) >= int(jobs[i * 8 + 5])) j = i;
    const long long* J = jobs + j * 8;
    W6W p{reinterpret_cast<const float*>(J[0]), reinterpret_cast<float*>(J[1]), int(J[2]), int(J[3]), int(J[4]), reinterpret_cast<unsigned short*>(J[7])};
    const long long i = (long long)(int(blockIdx.x) - int(J[5])) * 256 + threadIdx.x;
    if (J[6] == 4) wino6_weight_one<4>(p, i);
    else if (J[6] == 6) wino6_weight_one<6>(p, i);
    else wino6_weight_one<3>(p, i);
}

// ------------------------------------------------------------------------------------------------ input transform
struct W6In {
    const float* x; float* V; int B, C, x_row, x_img, x_hp, x_wp, TH, TW; long long T;
    float* Yt; int yTH, yTW; long long yT;       // Yt != nullptr: x is a dY buffer with a zero halo of R-1 and the lower-right 4x4 block of patch (ty, tx)
                                                 // is output-gradient tile (ty, tx): Yt = A dY A^T is emitted from the same read of dY
    // normalising variant: x = act((y - mean) * rstd) of a dense [B][H][W][C] tensor under a REFLECT halo of 1, evaluated on the fly
    const float* y; const float* mean; const float* rstd; int H, W, act; float slope;
    // dY variant that evaluates the instance-norm backward's second pass on the fly (lane-spread kernel, MODE 2): x is NOT read;
    // dY(h, w) = rstd * (g_z - mean(g_z) - z * mean(g_z z)) from the first pass's sums (in_bwd_dy: bitwise what in_bwd_pass2_kernel stores)
    InBwd nb; int nbB;
};

template <int VW> struct W6Vec;
template <> struct W6Vec<4> { typedef f32x4 T; };
template <> struct W6Vec<2> { typedef f32x2 T; };

// One thread = one N x N patch x VW channels (4 for the 6x6 patches of F(4x4,3x3); 2 for the 7x7 / 8x8 patches of F(4x4,4x4) /
// F(6x6,3x3): 49 / 64 float4 intermediates would not fit the register file).  MODE 0: x from the halo'd buffer; 1: forward input
// normalised on the fly (3x3 filters).
template <int V> struct W6VW { static constexpr int value = V == 3 ? 4 : 2; };

template <int V, int MODE>
__global__ __launch_bounds__(256) void wino6_input_kernel(const W6In p) {
    constexpr bool NORM = MODE == 1;
    constexpr int N = W6<V>::N, R = W6<V>::R, MO = W6<V>::MO, VW = W6VW<V>::value;
    typedef typename W6Vec<VW>::T V4;
    static_assert(MODE == 0 || R == 3, "the fused variants exist for the 3x3 filter");
    const int q4 = p.C / VW;
    const long long i = ng_xcd_remap(blockIdx.x, gridDim.x) * 256ll + threadIdx.x;        // neighbouring patches on one XCD (see the lane-spread kernel)
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    V4 z4;
#pragma unroll
    for (int e = 0; e < VW; ++e) z4[e] = 0.f;
    V4 mean = z4, rstd = z4;
    const float* base;
    if constexpr (NORM) {
        mean = *reinterpret_cast<const V4*>(p.mean + size_t(b) * p.C + q * VW);
        rstd = *reinterpret_cast<const V4*>(p.rstd + size_t(b) * p.C + q * VW);
        base = p.y + size_t(b) * p.H * p.W * p.C + q * VW;
    } else {
        base = p.x + size_t(b) * p.x_img + q * VW;
    }
    // value of the (virtual) halo'd buffer at patch position (a, c); lines past the buffer (extent not a multiple of MO) read zero:
    // they feed only outputs that are never stored.  Normalising variant: reflected row / column offsets are formed once per patch and
    // lines past the buffer read a clamped address times 0 -- an element costs one add, one load and the normalisation, no branch
    // (75 -> 39 us per layer for F(6x6,3x3); the same scheme for the plain variant raises its register count past the occupancy
    // it needs: 74 -> 105 us for the dY pass, so that one keeps the bounds test per element).
    int roff[N], coff[N];
    float rok[N], cok[N];
    float neg = 1.f;                                             // x > 0 ? x : x * neg  covers none / ReLU / LeakyReLU
    if constexpr (NORM) {
        neg = p.act == NIRGAN_ACT_RELU ? 0.f : p.act == NIRGAN_ACT_LRELU ? p.slope : 1.f;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const int rb = MO * ty + a, cb = MO * tx + a;
            rok[a] = rb < p.x_hp ? 1.f : 0.f;
            cok[a] = cb < p.x_wp ? 1.f : 0.f;
            roff[a] = ng_reflect((rb < p.x_hp ? rb : 0) - 1, p.H) * p.W * p.C;
            coff[a] = ng_reflect((cb < p.x_wp ? cb : 0) - 1, p.W) * p.C;
        }
    }
    auto ld = [&](int a, int c) -> V4 {
        if constexpr (NORM) {
            V4 v = (*reinterpret_cast<const V4*>(base + (roff[a] + coff[c])) - mean) * rstd;     // in_apply_kernel's arithmetic
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * neg;
            return v * (rok[a] * cok[c]);
        } else {
            const int rb = MO * ty + a, cb = MO * tx + c;
            if (rb >= p.x_hp || cb >= p.x_wp) return z4;
            return *reinterpret_cast<const V4*>(base + size_t(rb) * p.x_row + size_t(cb) * p.C);
        }
    };
    V4 m[N][N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        V4 d[N], o[N];
#pragma unroll
        for (int a = 0; a < N; ++a) d[a] = ld(a, c);
        W6<V>::bt(d, o);
#pragma unroll
        for (int a = 0; a < N; ++a) m[a][c] = o[a];
    }
    const size_t plane = size_t(p.T) * p.C;
    float* Vp = p.V + size_t(t) * p.C + q * VW;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        V4 o[N];
        W6<V>::bt(m[a], o);
#pragma unroll
        for (int c = 0; c < N; ++c) *reinterpret_cast<V4*>(Vp + (a * N + c) * plane) = o[c];
    }
    if constexpr (!NORM) {
        if (p.Yt != nullptr && ty < p.yTH && tx < p.yTW) {
            // output-gradient tile (ty, tx) = patch rows / columns R-1 .. R+MO-2 (just read: L1 / L2 hits); rows past the extent read the zero halo
            V4 u[N][MO];
#pragma unroll
            for (int c = 0; c < MO; ++c) {
                V4 e[MO], o[N];
#pragma unroll
                for (int a = 0; a < MO; ++a) e[a] = ld(R - 1 + a, R - 1 + c);
                W6<V>::a(e, o);
#pragma unroll
                for (int a = 0; a < N; ++a) u[a][c] = o[a];
            }
            const size_t yplane = size_t(p.yT) * p.C;
            float* Y = p.Yt + ((size_t(b) * p.yTH + ty) * p.yTW + tx) * p.C + q * VW;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                V4 o[N];
                W6<V>::a(u[a], o);
#pragma unroll
                for (int c = 0; c < N; ++c) *reinterpret_cast<V4*>(Y + (a * N + c) * yplane) = o[c];
            }
        }
    }
}

// The same transforms for the 8 x 8 patches of F(6x6,3x3) with the patch spread over the LANES instead of one thread's registers
// (round 3).  The one-thread-per-patch kernel above keeps 64 x 2 channels in registers between its two passes: 166-174 VGPRs, two to
// three waves per SIMD, eight dependent load rounds each -- latency holds it at 4.5 TB/s.  Here a wave takes one patch x 32 channels:
// lane = (patch column c, channel quad qg); its 8 loads (the 8 rows of column c, 16 bytes each, whole 128-byte lines per pixel) are
// all in flight at once; B^T runs down the column in registers, the 8 x 8 transpose goes through a wave-private 8 KB of LDS (written
// as whole 1 KB rows), B^T runs along the row in the lane that now owns plane row a', and the 8 results leave as 16-byte stores (128
// contiguous bytes per plane and instruction).  ~60 VGPRs.  Same arithmetic in the same order as the kernel above: bitwise equal.
// MODE 0: halo'd buffer (Yt = A dY A^T from the same loads when p.Yt is set); 1: forward input normalised on the fly; 2: dY pass whose
// elements are the instance-norm backward's second pass evaluated on the fly (dY is never stored: the pass that would write it and this
// kernel's read of it disappear -- 540 -> 388 MB per residual-block layer; two loads per element instead of one, which the per-thread
// form of round 2 could not afford at 240 VGPRs).
template <int V, int MODE>
__global__ __launch_bounds__(256) void wino6_input_coop_kernel(const W6In p) {
    constexpr int N = W6<V>::N, MO = W6<V>::MO, R = W6<V>::R;       // F(6x6,3x3): 8, 6, 3; F(4x4,4x4): 7, 4, 4 (lane rows >= N idle)
    constexpr bool NORM = MODE == 1;
    __shared__ __attribute__((aligned(16))) f32x4 lds[4][N * N * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane >> 3, qg = lane & 7;
    const int groups = p.C / 32;
    // consecutive patches on ONE XCD (ng_xcd_remap): neighbouring patches share two of their eight rows and columns, and under the
    // round-robin dispatch they sat on different XCDs -- every L2 fetched the overlap for itself (PMC: 230 MB read per launch of the fused
    // dY pass against 134 MB of tensors)
    const long long unit = ng_xcd_remap(blockIdx.x, gridDim.x) * 4ll + wave, units = p.T * groups;
    const bool live = unit < units;
    const long long t = live ? unit / groups : 0;
    const int cg = live ? int(unit - t * groups) : 0;
    const int tx = int(t % p.TW);
    const long long r_ = t / p.TW;
    const int ty = int(r_ % p.TH), b = int(r_ / p.TH);
    const int ch = cg * 32 + qg * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[N];
    if constexpr (NORM) {
        const f32x4 mean = ld4(p.mean + size_t(b) * p.C + ch), rstd = ld4(p.rstd + size_t(b) * p.C + ch);
        const float neg = p.act == NIRGAN_ACT_RELU ? 0.f : p.act == NIRGAN_ACT_LRELU ? p.slope : 1.f;
        const float* base = p.y + size_t(b) * p.H * p.W * p.C + ch;
        const int cb = MO * tx + c;
        const float cok = cb < p.x_wp ? 1.f : 0.f;
        const int coff = ng_reflect((cb < p.x_wp ? cb : 0) - 1, p.W) * p.C;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const int rb = MO * ty + a;
            const float ok = (rb < p.x_hp ? 1.f : 0.f) * cok;
            const int roff = ng_reflect((rb < p.x_hp ? rb : 0) - 1, p.H) * p.W * p.C;
            f32x4 v = (ld4(base + (roff + coff)) - mean) * rstd;                  // in_apply_kernel's arithmetic
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * neg;
            d[a] = v * ok;
        }
    } else if constexpr (MODE == 2) {
        const InBwd& n = p.nb;
        const int q = ch >> 2;
        const float* mm = n.ws + size_t(p.nbB) * n.pchunks * 2 * n.C + size_t(b) * 2 * n.C;
        const f32x4 m1 = ld4(mm + ch), m2 = ld4(mm + n.C + ch);
        const f32x4 mean = ld4(n.mean + size_t(b) * n.C + ch), rstd = ld4(n.rstd + size_t(b) * n.C + ch);
        const float* yb = y_at(n.y, size_t(b) * n.HW * n.C, n.y16);
        const float* gsb = n.gsum_out ? n.gsum_out + size_t(b) * n.HW * n.C : nullptr;
        const float* gb = n.g ? y_at(n.g, size_t(b) * n.g_img, n.g16) : nullptr;
        const float* g2b = n.g2 ? n.g2 + size_t(b) * n.HW * n.C : nullptr;
        const int w = MO * tx + c - (R - 1);                    // the dY grid has a zero halo of R - 1
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const int h = MO * ty + a - (R - 1);
            d[a] = (c < N && h >= 0 && w >= 0 && h < n.H && w < n.W) ? in_bwd_dy(n, gb, g2b, gsb, yb, mean, rstd, m1, m2, h, w, q) : z4;
        }
    } else {
        const float* base = p.x + size_t(b) * p.x_img + ch;
        const int cb = MO * tx + c;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const int rb = MO * ty + a;
            d[a] = (c < N && rb < p.x_hp && cb < p.x_wp) ? ld4(base + size_t(rb) * p.x_row + size_t(cb) * p.C) : z4;
        }
    }
    f32x4* buf = lds[wave];
    f32x4 o[N], m[N];
    W6<V>::bt(d, o);
    const bool lane_ok = c < N;                                 // (7 x 7 patches leave the eighth lane row idle)
    if (lane_ok) {
#pragma unroll
        for (int a = 0; a < N; ++a) buf[(a * N + c) * 8 + qg] = o[a];
    }
    __syncthreads();
    const int ap = lane_ok ? c : 0;                             // this lane now owns plane row a' = its old column index
#pragma unroll
    for (int cc = 0; cc < N; ++cc) m[cc] = buf[(ap * N + cc) * 8 + qg];
    W6<V>::bt(m, o);
    if (live && lane_ok) {
        const size_t plane = size_t(p.T) * p.C;
        float* Vp = p.V + size_t(t) * p.C + ch;
#pragma unroll
        for (int cc = 0; cc < N; ++cc) st4(Vp + (ap * N + cc) * plane, o[cc]);
    }
    if constexpr (!NORM) {
        if (p.Yt != nullptr) {                                  // uniform over the block
            // output-gradient tile (ty, tx) = patch rows / columns 2 .. 7: the column pass of A on the rows already in registers
            f32x4 u[N];
            W6<V>::a(d + (R - 1), u);                           // lanes of columns < R - 1 carry values nobody reads
            __syncthreads();
            if (lane_ok) {
#pragma unroll
                for (int a = 0; a < N; ++a) buf[(a * N + c) * 8 + qg] = u[a];
            }
            __syncthreads();
            f32x4 e[MO];
#pragma unroll
            for (int cc = 0; cc < MO; ++cc) e[cc] = buf[(ap * N + (R - 1) + cc) * 8 + qg];
            W6<V>::a(e, u);
            if (live && lane_ok && ty < p.yTH && tx < p.yTW) {
                const size_t yplane = size_t(p.yT) * p.C;
                float* Y = p.Yt + ((size_t(b) * p.yTH + ty) * p.yTW + tx) * p.C + ch;
#pragma unroll
                for (int cc = 0; cc < N; ++cc) st4(Y + (ap * N + cc) * yplane, u[cc]);
            }
        }
    }
}

// Yt[f][t][k] = (A dY A^T)[f] alone (weight gradient without a Winograd data gradient next to it)
struct W6Dy { const float* dy; float* Yt; int B, H, W, K, d_row, d_img, d_org, TH, TW; long long T; };

template <int V>
__global__ __launch_bounds__(256) void wino6_dy_kernel(const W6Dy p) {
    constexpr int N = W6<V>::N, MO = W6<V>::MO, VW = W6VW<V>::value;
    typedef typename W6Vec<VW>::T V4;
    const int q4 = p.K / VW;
    const long long i = ng_xcd_remap(blockIdx.x, gridDim.x) * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const float* src = p.dy + size_t(b) * p.d_img + p.d_org + q * VW;
    V4 z4;
#pragma unroll
    for (int e = 0; e < VW; ++e) z4[e] = 0.f;
    V4 u[N][MO];
#pragma unroll
    for (int c = 0; c < MO; ++c) {
        V4 e[MO], o[N];
#pragma unroll
        for (int a = 0; a < MO; ++a) {
            const int h = MO * ty + a, w = MO * tx + c;
            e[a] = (h < p.H && w < p.W) ? *reinterpret_cast<const V4*>(src + size_t(h) * p.d_row + size_t(w) * p.K) : z4;
        }
        W6<V>::a(e, o);
#pragma unroll
        for (int a = 0; a < N; ++a) u[a][c] = o[a];
    }
    const size_t plane = size_t(p.T) * p.K;
    float* Y = p.Yt + size_t(t) * p.K + q * VW;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        V4 o[N];
        W6<V>::a(u[a], o);
#pragma unroll
        for (int c = 0; c < N; ++c) *reinterpret_cast<V4*>(Y + (a * N + c) * plane) = o[c];
    }
}

// ------------------------------------------------------------------------------------------------ plane GEMMs
struct W6Gemm { ng::ConvParams p; long long in_plane, w_plane, out_plane; int per_plane, total; };

// general form: the direct 128 x 128 x 32 tile (two resident workgroups per CU)
__global__ __launch_bounds__(256, 2) void wino6_gemm_kernel(const W6Gemm g) {
    __shared__ __attribute__((aligned(16))) char st0[(128 + 128) * 128];
    __shared__ __attribute__((aligned(16))) char st1[(128 + 128) * 128];
    // consecutive logical ids on one XCD: the blocks of a plane share U[f] (256 KB) and, pairwise, their V rows in that XCD's L2
    const int rid = ng_xcd_remap(blockIdx.x, g.total);
    const int plane = rid / g.per_plane, local = rid - plane * g.per_plane;
    ng::conv_tile<128, 0>(g.p, local, st0, st1, g.p.in + size_t(plane) * g.in_plane, g.p.w + size_t(plane) * g.w_plane,
                          g.p.out + size_t(plane) * g.out_plane);
}

// (round 4, A/B: NIRGAN_W6_TILE256) the plane GEMMs on the exact-fp32 form of the 256 x 256 eight-phase tile (igemm_tile256.h), persistent
// workgroups, one per CU, walking (plane, tile) pairs: measured against wino6_gemm32p_kernel in profiles/r04_plane_gemm_tile256.txt
__global__ __launch_bounds__(512, 2) void wino6_gemm256_kernel(const W6Gemm g, const int per_plane, const int total) {
    __shared__ __attribute__((aligned(16))) char lds[ng::T256_LDS];
    bool again = false;
    for (int rid = ng_xcd_remap(blockIdx.x, gridDim.x); rid < total; rid += gridDim.x) {
        if (again) ng::t256_bar();
        const int plane = rid / per_plane, local = rid - plane * per_plane;
        ng::conv_tile256<true>(g.p, local, lds, g.p.in + size_t(plane) * g.in_plane, g.p.w + size_t(plane) * g.w_plane, g.p.out + size_t(plane) * g.out_plane);
        again = true;
    }
}

// The data gradient's plane GEMMs and the 36 weight-gradient problems of the same layer in ONE grid (both read what the dY pass just
// wrote): the long weight-gradient blocks (18 K-steps) are dispatched first, the GEMM blocks (8 K-steps) pack behind them, so neither
// launch pays its own partly filled last round.  32-k stages for both halves (the direct tile and the weight-gradient tile share 64 KB).
__global__ __launch_bounds__(256, 2) void wino6_pair_kernel(const W6Gemm g, const ng::WgradParams wp, const int wgrad_blocks, const int wgrad_first) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int bid = blockIdx.x;
    const int gemm_first_id = wgrad_first ? wgrad_blocks : 0;
    const bool is_wgrad = wgrad_first ? bid < wgrad_blocks : bid >= g.total;
    if (is_wgrad) {
        ng::wgrad_tile<128, 0>(wp, wgrad_first ? bid : bid - g.total, lds, lds + 32768);
    } else {
        const int rid = ng_xcd_remap(bid - gemm_first_id, g.total);
        const int plane = rid / g.per_plane, local = rid - plane * g.per_plane;
        ng::conv_tile<128, 0>(g.p, local, lds, lds + 32768, g.p.in + size_t(plane) * g.in_plane, g.p.w + size_t(plane) * g.w_plane,
                              g.p.out + size_t(plane) * g.out_plane);
    }
}

// The plane GEMMs contract over C = 256 only: 8 K-steps of the tile above against a prologue + epilogue of ~20 k cycles per tile,
// (measured 103-110 TFLOP/s at C = 256 against 126 at C = 1024, scripts/bench_wino6_gemm.py).  This variant stages 16 k per step:
// 64-byte rows, 2 x 16 KB of LDS and 112 VGPRs, so up to FOUR workgroups are resident per CU.  Measured: within 2 % of the 32-k tile
// either way (186 vs 188 us at T = 4096, 211 vs 220 at T = 4624, 355 vs 341 at T = 8192) -- more resident workgroups do not buy back
// the per-tile cost; it is the default because the step as a whole runs 0.6 % faster with it (NIRGAN_WINO6_GEMM32=1 selects the other).
//   A [T][C] (row-major, rows = tiles), B [K][C], Out [T][K];  128 x 128 block tile, 4 waves as 2 x 2, wave = 64 x 64.
//   LDS-DMA piece = 16 rows x 64 B; the 16-byte chunk index is XOR-swizzled with (row >> 2) & 3 on the source side and at the
//   ds_read_b128 (8 consecutive rows hit 8 distinct 16-byte bank groups).  Lanes 0-31 read chunk 2g, lanes 32-63 chunk 2g+1 of their
//   row: MFMA j of group g contracts k = 8g + j and 8g + 4 + j (same permutation of k for A and B: the sum is unchanged).
struct W6G16 { const float* A; const float* Bw; float* Out; const float* zero; int T, C, K, mtiles, ntiles, per_plane, total;
               long long a_plane, b_plane, o_plane; };

__global__ __launch_bounds__(256, 3) void wino6_gemm16_kernel(const W6G16 p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 16384];
    constexpr int STAGE = 16384, A_BYTES = 8192;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rid = ng_xcd_remap(blockIdx.x, p.total);
    const int plane = rid / p.per_plane;
    const int id = rid - plane * p.per_plane;        // consecutive in time on one XCD: the two N tiles of an M tile share their V rows in L2
    const int n0 = (id % p.ntiles) * 128, m0 = (id / p.ntiles) * 128;
    const float* A = p.A + size_t(plane) * p.a_plane;
    const float* Bw = p.Bw + size_t(plane) * p.b_plane;
    float* Out = p.Out + size_t(plane) * p.o_plane;

    // ---------------- loader: wave w owns A pieces 2w, 2w+1 and B pieces 2w, 2w+1 (16 rows x 64 B each)
    const int prow = lane >> 2, lchunk = lane & 3;
    int a_base[2], b_base[2];
    bool b_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + prow;
        const int sc = lchunk ^ ((row >> 2) & 3);
        int m = m0 + row;
        m = m < p.T ? m : p.T - 1;
        a_base[i] = m * p.C + sc * 4;
        const int n = n0 + row;
        b_ok[i] = n < p.K;
        b_base[i] = (b_ok[i] ? n : 0) * p.C + sc * 4;
    }
    auto issue = [&](char* sA, int c0) {
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) ng_glds16(A + (a_base[i] + c0), sA + (wave * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) ng_glds16(b_ok[i] ? Bw + (b_base[i] + c0) : p.zero, sB + (wave * 2 + i) * 1024);
    };

    // ---------------- compute: wave (wr, wc) = rows wr*64 .. +63, columns wc*64 .. +63
    const int wr = wave >> 1, wc = wave & 1, half = lane >> 5;
    int a_off[2], a_key[2], b_off[2], b_key[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wr * 64 + t * 32 + (lane & 31), rb = wc * 64 + t * 32 + (lane & 31);
        a_off[t] = ra * 64; a_key[t] = (ra >> 2) & 3;
        b_off[t] = rb * 64; b_key[t] = (rb >> 2) & 3;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    auto compute = [&](const char* sA) {
        const char* sB = sA + A_BYTES;
        f32x4 a[2][2], b[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int chunk = 2 * g + half;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[g][t] = *reinterpret_cast<const f32x4*>(sA + a_off[t] + ((chunk ^ a_key[t]) << 4));
                b[g][t] = *reinterpret_cast<const f32x4*>(sB + b_off[t] + ((chunk ^ b_key[t]) << 4));
            }
        }
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][mt][j], b[g][nt][j], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nk = p.C >> 4;
    issue(lds, 0);
    for (int s = 0; s < nk; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) issue(lds + ((s + 1) & 1) * STAGE, (s + 1) << 4);
        compute(lds + (s & 1) * STAGE);
    }

    // ---------------- epilogue: the tile goes through LDS in two halves of 64 rows (32 KB) so that every store is a whole 16-byte
    // row segment.  C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    float* buf = reinterpret_cast<float*>(lds);
    const int chunk = tid & 31, row0 = tid >> 5;
    const int n = n0 + chunk * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wr == h) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int col = wc * 64 + nt * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 128 + col] = acc[mt][nt][r];
                }
        }
        __syncthreads();
#pragma unroll 4
        for (int row = row0; row < 64; row += 8) {
            const int m = m0 + h * 64 + row;
            if (m < p.T && n < p.K) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(buf + row * 128 + chunk * 4);
                *reinterpret_cast<f32x4*>(Out + size_t(m) * p.K + n) = v;           // K % 4 == 0: whole float4s
            }
        }
    }
}

// The same tile as a PERSISTENT workgroup with the epilogue folded into the next tile's K loop.  Measured on the kernel above: a tile
// of 16 K-steps spends ~3.5 us per wave outside its MFMA stream (first DMA wait + epilogue through LDS), and because equal tiles keep
// the resident workgroups of a CU in lockstep those phases coincide instead of filling each other (0.64 of peak at C = 256 against
// 0.82 at C = 1024).  Here a workgroup walks its tiles as ONE stream of K-steps: the DMA of the next tile's first step is issued
// during this tile's last step, and the finished tile's accumulators (a second set, 64 AGPRs) drain one 8 x 32 piece per K-step of
// the next tile -- 4 ds_write_b32 into a wave-private 1 KB transpose buffer, one ds_read_b128, one 16-byte-per-lane store -- under
// that step's 32 MFMAs.  No barrier beyond the one per K-step, no MFMA-free phase except the very first DMA wait.
// KS = k per stage: 16 (64-byte rows, 2 x 16 KB of stages) or 32 (128-byte rows, 2 x 32 KB: half the barriers per product)
template <int KS> constexpr int w6p_lds() { return 2 * 2 * 128 * KS * 4 + 4 * 2 * 1280; }
constexpr int W6P_LDS = w6p_lds<16>();

template <int KS>
__device__ __forceinline__ void w6_gemmp_body(const W6G16& p, const int first, const int stride, char* lds) {
    constexpr int ROWB = KS * 4, CH = KS / 4, RPP = 1024 / ROWB, PW = (128 / RPP) / 4, NG = KS / 8, PPS = KS / 16;
    constexpr int A_BYTES = 128 * ROWB, STAGE = 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.C / KS;                         // host guarantees nk * PPS >= 16: the drain of a tile takes 16 / PPS K-steps of the next

    // ---------------- loader: wave w owns A pieces PW w .. and B pieces PW w .. (a piece = RPP rows x ROWB bytes = 1 KB); the 16-byte
    // chunk of a row is XOR-swizzled with key(row) on the source side and at the fragment reads
    auto key = [](int row) { return KS == 16 ? (row >> 2) & 3 : (row >> 1) & 7; };
    const int prow = lane / CH, lchunk = lane % CH;
    // per-lane BYTE offsets (32 bits) from the plane's uniform base pointers: the LDS-DMA then takes its address as SGPR base + VGPR
    // offset and a K-step advances the scalar base only -- no vector arithmetic per piece
    struct Ld { const char* A; const char* Bw; unsigned a_off[PW], b_off[PW]; };
    auto setup = [&](int logical, Ld& l, int& plane, int& m0, int& n0) {
        const int rid = ng_xcd_remap(logical, p.total);
        plane = rid / p.per_plane;
        const int id = rid - plane * p.per_plane;
        n0 = (id % p.ntiles) * 128;
        m0 = (id / p.ntiles) * 128;
        l.A = reinterpret_cast<const char*>(p.A + size_t(plane) * p.a_plane);
        l.Bw = reinterpret_cast<const char*>(p.Bw + size_t(plane) * p.b_plane);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = (wave * PW + i) * RPP + prow;
            const int sc = lchunk ^ key(row);
            int m = m0 + row;
            m = m < p.T ? m : p.T - 1;
            l.a_off[i] = unsigned(m * p.C + sc * 4) * 4u;
            int n = n0 + row;
            n = n < p.K ? n : p.K - 1;                       // rows past K re-read the last one: their columns are never stored
            l.b_off[i] = unsigned(n * p.C + sc * 4) * 4u;
        }
    };
    auto issue = [&](const Ld& l, char* sA, int c0) {
        char* sB = sA + A_BYTES;
        const char* ab = ng_uniform_ptr(l.A + size_t(c0) * 4);
        const char* bb = ng_uniform_ptr(l.Bw + size_t(c0) * 4);
#pragma unroll
        for (int i = 0; i < PW; ++i) ng_glds16_so(ab, l.a_off[i], sA + (wave * PW + i) * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) ng_glds16_so(bb, l.b_off[i], sB + (wave * PW + i) * 1024);
    };

    // ---------------- compute: wave (wr, wc) = rows wr*64 .. +63, columns wc*64 .. +63
    const int wr = wave >> 1, wc = wave & 1, half = lane >> 5;
    int a_off[2], a_key[2], b_off[2], b_key[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wr * 64 + t * 32 + (lane & 31), rb = wc * 64 + t * 32 + (lane & 31);
        a_off[t] = ra * ROWB; a_key[t] = key(ra);
        b_off[t] = rb * ROWB; b_key[t] = key(rb);
    }
    // group g contracts k 8g .. 8g+7: lanes 0-31 read chunk 2g, lanes 32-63 chunk 2g+1 of their row; the fragment reads of group g+1 are
    // issued before the 16 MFMAs of group g
    // `between(g)` runs after the MFMAs of group g have been issued: whatever it does executes in their shadow (the drain of the
    // previous tile: measured 13 % of the launch when it sat in front of the step's MFMAs, where its LDS round trip stalled the wave)
    auto compute = [&](const char* sA, f32x16 (&acc)[2][2], auto&& between) {
        const char* sB = sA + A_BYTES;
        f32x4 a[2][2], b[2][2];
        auto load = [&](int g, int slot) {
            const int chunk = 2 * g + half;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[slot][t] = *reinterpret_cast<const f32x4*>(sA + a_off[t] + ((chunk ^ a_key[t]) << 4));
                b[slot][t] = *reinterpret_cast<const f32x4*>(sB + b_off[t] + ((chunk ^ b_key[t]) << 4));
            }
        };
        load(0, 0);
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load(g + 1, (g + 1) & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][mt][j], b[g & 1][nt][j], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            between(g);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---------------- drain: piece q = (mt, nt, j) of a finished tile: registers 4j .. 4j+3 of acc[mt][nt] = rows 8j + {0..3} + 4 half of
    // that 32 x 32 block.  Wave-private transpose buffer [8 rows][40 floats] (the 8-float pad keeps the two half-waves on disjoint banks),
    // double-buffered by piece parity so that the only ordering needed is the wave's own lgkmcnt.
    float* tb = reinterpret_cast<float*>(lds + 2 * STAGE) + wave * 640;
    const int t_wr = ((lane >> 5) * 4) * 40 + (lane & 31);           // + (r & 3) * 40
    const int t_rd = (lane >> 3) * 40 + (lane & 7) * 4;              // lane -> (row lane/8, 4 columns)
    f32x4 dv;                                                        // the piece on its way from the transpose buffer to memory
    auto drain_lds = [&](const f32x16 (&acc)[2][2], int q) {
        const int mt = q >> 3, nt = (q >> 2) & 1, j = q & 3;
        float* b = tb + (q & 1) * 320;
#pragma unroll
        for (int r = 0; r < 4; ++r) b[t_wr + r * 40] = acc[mt][nt][4 * j + r];
        dv = *reinterpret_cast<const f32x4*>(b + t_rd);
    };
    auto drain_store = [&](int q, int plane, int m0, int n0) {
        const int mt = q >> 3, nt = (q >> 2) & 1, j = q & 3;
        const int rr = lane >> 3;                                     // buffer row = (r & 3) + 4 half  ->  tile row 8j + rr
        const int m = m0 + wr * 64 + mt * 32 + 8 * j + rr;
        const int n = n0 + wc * 64 + nt * 32 + (lane & 7) * 4;
        if (m < p.T && n < p.K) *reinterpret_cast<f32x4*>(p.Out + size_t(plane) * p.o_plane + size_t(m) * p.K + n) = dv;
    };
    auto drain = [&](const f32x16 (&acc)[2][2], int q, int plane, int m0, int n0) {
        drain_lds(acc, q);
        drain_store(q, plane, m0, n0);
    };

    f32x16 accA[2][2], accB[2][2];
    auto zero = [&](f32x16 (&acc)[2][2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };

    // one tile: nk K-steps into `cur`, draining `prev` (the previous tile, if any) along the first 16 of them; the DMA of the next
    // tile's first step goes out during the last step.  `stage` = parity of the running step count.
    int step = 0;
    Ld lcur, lnext;
    int plane, m0, n0, pplane = 0, pm0 = 0, pn0 = 0, nplane = 0, nm0 = 0, nn0 = 0;
    auto one_step = [&](f32x16 (&cur)[2][2], int s, int next_logical, auto&& between) {
        if (s + 1 < nk) issue(lcur, lds + ((step + 1) & 1) * STAGE, (s + 1) * KS);
        else if (next_logical < p.total) {
            setup(next_logical, lnext, nplane, nm0, nn0);
            issue(lnext, lds + ((step + 1) & 1) * STAGE, 0);
        }
        compute(lds + (step & 1) * STAGE, cur, between);
        ++step;
    };
    auto run_tile = [&](f32x16 (&cur)[2][2], const f32x16 (&prev)[2][2], bool have_prev, int next_logical) {
        zero(cur);
        static_assert(PPS == 1 || PPS == 2, "one or two drain pieces per K-step");
#pragma unroll
        for (int s = 0; s < 16 / PPS; ++s) {                 // the piece index is a compile-time constant: register-indexed drain
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // the LDS half of a piece after the first MFMA group, its store after the next one (the round trip is long over by then)
            one_step(cur, s, next_logical, [&](int g) {
                if (!have_prev) return;
                if (PPS == 1) {
                    if (g == 0) drain_lds(prev, s);
                    if (g == NG - 1) drain_store(s, pplane, pm0, pn0);
                } else {
                    if (g == 0) drain_lds(prev, 2 * s);
                    if (g == 1) { drain_store(2 * s, pplane, pm0, pn0); drain_lds(prev, 2 * s + 1); }
                    if (g == NG - 1) drain_store(2 * s + 1, pplane, pm0, pn0);
                }
            });
        }
        for (int s = 16 / PPS; s < nk; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            one_step(cur, s, next_logical, [](int) {});
        }
        pplane = plane; pm0 = m0; pn0 = n0;
        lcur = lnext; plane = nplane; m0 = nm0; n0 = nn0;
    };
    // tiles first, first + stride, ...: a drawn-from-a-counter assignment was tried (tests green) and measured 12 % SLOWER -- tiles in
    // flight at the same time stop being neighbours, and the XCD-local L2 sharing of V rows and U planes is worth more than the balance
    if (first >= p.total) return;
    setup(first, lcur, plane, m0, n0);
    issue(lcur, lds, 0);
    int logical = first;
    bool have_prev = false;
    while (true) {
        run_tile(accA, accB, have_prev, logical + stride);
        logical += stride;
        have_prev = true;
        if (logical >= p.total) {
#pragma unroll
            for (int q = 0; q < 16; ++q) drain(accA, q, pplane, pm0, pn0);
            break;
        }
        run_tile(accB, accA, true, logical + stride);
        logical += stride;
        if (logical >= p.total) {
#pragma unroll
            for (int q = 0; q < 16; ++q) drain(accB, q, pplane, pm0, pn0);
            break;
        }
    }
}

__global__ __launch_bounds__(256, 2) void wino6_gemm16p_kernel(const W6G16 p) {
    __shared__ __attribute__((aligned(16))) char lds[w6p_lds<16>()];
    w6_gemmp_body<16>(p, blockIdx.x, gridDim.x, lds);
}

__global__ __launch_bounds__(256, 2) void wino6_gemm32p_kernel(const W6G16 p) {
    __shared__ __attribute__((aligned(16))) char lds[w6p_lds<32>()];
    w6_gemmp_body<32>(p, blockIdx.x, gridDim.x, lds);
}

// (KS = 32 -- 128-byte rows, half the barriers per product, 74 KB of LDS -- was instantiated and measured: 176 / 195 us against
// 178 / 194 us at T = 4096 / 4624: the barrier count is not what holds the tile at 0.7)

// One grid for a layer's backward products: 512 persistent workgroups, each first walks its share of the transform-domain weight-
// gradient units (both operands were just written by the dY pass and the forward), then its share of the data gradient's plane-GEMM
// tiles -- M is the last thing written before the output transform reads it.  With F(6x6,3x3) at bs 16 the shares are exact:
// 1 024 weight-gradient units = 2 per workgroup, 2 048 GEMM tiles = 4 per workgroup.
__global__ __launch_bounds__(256, 2) void wino6_pair16p_kernel(const W6G16 q, const int nblocks, const ng::WgradParams wp) {
    __shared__ __attribute__((aligned(16))) char lds[w6p_lds<32>() > 65536 ? w6p_lds<32>() : 65536];
    ng::wgrad_persist(wp, blockIdx.x, nblocks, lds);
    __syncthreads();
    w6_gemmp_body<32>(q, blockIdx.x, nblocks, lds);          // 32-k stages: see wino6_gemm32p_kernel
}

// (fallback) the weight-gradient tiles one per workgroup, dispatched first, and the persistent GEMM workgroups behind them
__global__ __launch_bounds__(256, 2) void wino6_pair16_kernel(const W6G16 q, const int gemm_blocks, const ng::WgradParams wp, const int wgrad_blocks) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int bid = blockIdx.x;
    if (bid < wgrad_blocks) ng::wgrad_tile<128, 0>(wp, bid, lds, lds + 32768);
    else w6_gemmp_body<16>(q, bid - wgrad_blocks, gemm_blocks, lds);
}

// ------------------------------------------------------------------------------------------------ output transform
struct W6Out { const float* M; const float* bias; float* y; int B, H, W, K, TH, TW; long long T; float* stats; };

template <int V>
__global__ __launch_bounds__(256) void wino6_output_kernel(const W6Out p) {
    constexpr int N = W6<V>::N, MO = W6<V>::MO, VW = W6VW<V>::value;
    typedef typename W6Vec<VW>::T V4;
    const int q4 = p.K / VW;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const size_t plane = size_t(p.T) * p.K;
    const float* M = p.M + size_t(t) * p.K + q * VW;
    V4 s[MO][N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        V4 m[N], o[MO];
#pragma unroll
        for (int a = 0; a < N; ++a) m[a] = *reinterpret_cast<const V4*>(M + (a * N + c) * plane);
        W6<V>::at(m, o);
#pragma unroll
        for (int a = 0; a < MO; ++a) s[a][c] = o[a];
    }
    V4 bv;
#pragma unroll
    for (int e = 0; e < VW; ++e) bv[e] = 0.f;
    if (p.bias != nullptr) bv = *reinterpret_cast<const V4*>(p.bias + q * VW);
    float* yb = p.y + size_t(b) * p.H * p.W * p.K + q * VW;
    // partial sums for the instance norm that follows, per tile and channel FOUR values: a shift k (the tile's first output, without
    // the bias), sum (o - k), sum (o - k)^2 over the tile's stored outputs, and their count (edge tiles store fewer): about a value of
    // the data itself the sums carry no cancellation; nirgan_instnorm_fwd re-bases the tiles onto one shift.  The statistics pass over y
    // is not needed then
    V4 s1 = bv * 0.f, s2 = bv * 0.f, k0 = bv * 0.f;
    int cnt = 0;
#pragma unroll
    for (int a = 0; a < MO; ++a) {
        V4 o[MO];
        W6<V>::at(s[a], o);
        if (a == 0) k0 = o[0];                      // (h, w) = (MO ty, MO tx) always exists
        const int h = MO * ty + a;
        if (h >= p.H) continue;
#pragma unroll
        for (int c = 0; c < MO; ++c) {
            const int w = MO * tx + c;
            if (w < p.W) {
                *reinterpret_cast<V4*>(yb + (size_t(h) * p.W + w) * p.K) = o[c] + bv;
                const V4 dlt = o[c] - k0;
                s1 += dlt;
                s2 += dlt * dlt;
                ++cnt;
            }
        }
    }
    if (p.stats != nullptr) {
        float* sp = p.stats + size_t(t) * 4 * p.K + q * VW;
        V4 nv;
#pragma unroll
        for (int e = 0; e < VW; ++e) nv[e] = float(cnt);
        *reinterpret_cast<V4*>(sp) = k0;
        *reinterpret_cast<V4*>(sp + p.K) = s1;
        *reinterpret_cast<V4*>(sp + 2 * p.K) = s2;
        *reinterpret_cast<V4*>(sp + 3 * p.K) = nv;
    }
}

// ------------------------------------------------------------------------------------------------ output transform + first pass of the
// instance-norm backward of the layer that consumes this data gradient (nirgan_wino6_desc.fuse_*).  The tile's MO x MO outputs over the
// padded extent stay in registers: the reflect halo (width 1) is folded onto the interior there (rows, then columns: the adjoint of
// ReflectionPad2d(1) is separable; the host guarantees that the far halo line and its partner share a tile), the skip gradient is
// added, g_a is stored dense, and the tile's sums of g_z = g_a * act'(z) and g_z * z leave as one chunk of partial sums.
// (the pointers are separate __restrict__ kernel arguments: inside a parameter struct the stores of g_a would fence the loads of the
// next pixel -- 90 us per layer instead of the ~55 the bytes cost)
struct W6Fuse { const float* __restrict__ y; const float* __restrict__ mean; const float* __restrict__ rstd; const float* __restrict__ g2;
                float* __restrict__ gz; float* __restrict__ part; int act; float slope; };

template <int V>
__global__ __launch_bounds__(256) void wino6_output_inbwd_kernel(const W6Out p, const float* __restrict__ fy, const float* __restrict__ fmean,
                                                                 const float* __restrict__ frstd, const float* __restrict__ fg2,
                                                                 float* __restrict__ fgz, float* __restrict__ fpart, const int fact, const float fslope) {
    const W6Fuse f{fy, fmean, frstd, fg2, fgz, fpart, fact, fslope};
    constexpr int N = W6<V>::N, MO = W6<V>::MO, VW = W6VW<V>::value;
    typedef typename W6Vec<VW>::T V4;
    const int q4 = p.K / VW;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const size_t plane = size_t(p.T) * p.K;
    const float* M = p.M + size_t(t) * p.K + q * VW;
    // A^T M A row by row of M: a row of 8 planes is reduced over the columns (8 -> 6), then spread over the 6 output rows with the
    // constants of A^T's column -- 36 accumulators instead of the 48 intermediates of the two-pass form (the fused mode needs the
    // whole tile in registers for the fold; this keeps the kernel at three waves per SIMD)
    static_assert(V != 3, "the row-streaming form reads A^T from the tables (F(4x4,4x4) / F(6x6,3x3))");
    V4 o[MO][MO];
#pragma unroll
    for (int a = 0; a < MO; ++a)
#pragma unroll
        for (int c = 0; c < MO; ++c) o[a][c] = V4{} * 0.f;
#pragma unroll
    for (int ap = 0; ap < N; ++ap) {
        V4 m[N], tr[MO];
#pragma unroll
        for (int c = 0; c < N; ++c) m[c] = *reinterpret_cast<const V4*>(M + (ap * N + c) * plane);
        W6<V>::at(m, tr);
#pragma unroll
        for (int a = 0; a < MO; ++a)
#pragma unroll
            for (int c = 0; c < MO; ++c) w6_mac(o[a][c], W6<V>::cAT(a, ap), tr[c]);
    }
    const int Hi = p.H - 2, Wi = p.W - 2;                     // interior extent; padded coordinate = interior + 1
    // fold: padded line 0 onto padded line 2, padded line Hi + 1 onto padded line Hi - 1 (rows, then columns)
#pragma unroll
    for (int a = 0; a < MO; ++a) {
        const int hp = MO * ty + a;
        if (a >= 2 && hp == 2) {
#pragma unroll
            for (int c = 0; c < MO; ++c) o[a][c] += o[a - 2][c];
        }
        if (a + 2 < MO && hp == Hi - 1) {
#pragma unroll
            for (int c = 0; c < MO; ++c) o[a][c] += o[a + 2][c];
        }
    }
#pragma unroll
    for (int c = 0; c < MO; ++c) {
        const int wp = MO * tx + c;
        if (c >= 2 && wp == 2) {
#pragma unroll
            for (int a = 0; a < MO; ++a) o[a][c] += o[a][c - 2];
        }
        if (c + 2 < MO && wp == Wi - 1) {
#pragma unroll
            for (int a = 0; a < MO; ++a) o[a][c] += o[a][c + 2];
        }
    }
    // Per-pixel phase on channel QUADS: lanes 2k and 2k+1 hold the channel pairs 4k..4k+1 and 4k+2..4k+3 of the same tile; they swap
    // halves of the tile (one DPP move per value) so that the even lane owns tile rows 0..2 and the odd lane rows 3..5 with four
    // channels each -- 18 pixels x 16 bytes per lane instead of 36 x 8: half the memory instructions of this phase.
    static_assert(VW == 2 && MO == 6, "the lane-pair exchange is written for channel pairs and 6x6 tiles");
    const bool odd = (threadIdx.x & 1) != 0;                    // q is odd exactly when the lane is (K / 2 and the block size are even)
    const int quad = (q >> 1) * 4;
    const f32x4 mean = *reinterpret_cast<const f32x4*>(f.mean + size_t(b) * p.K + quad);
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(f.rstd + size_t(b) * p.K + quad);
    const float neg = f.act == NIRGAN_ACT_RELU ? 0.f : f.act == NIRGAN_ACT_LRELU ? f.slope : 1.f;
    const size_t img = size_t(b) * Hi * Wi * p.K + quad;
    auto swap1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)); };   // lane i <- lane i ^ 1
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int h = MO * ty + (odd ? a + 3 : a) - 1;
#pragma unroll
        for (int c = 0; c < MO; ++c) {
            const V4 mine = odd ? o[a + 3][c] : o[a][c], send = odd ? o[a][c] : o[a + 3][c];
            V4 got;
            got[0] = swap1(send[0]);
            got[1] = swap1(send[1]);
            f32x4 ga = odd ? f32x4{got[0], got[1], mine[0], mine[1]} : f32x4{mine[0], mine[1], got[0], got[1]};
            const int w = MO * tx + c - 1;
            if (h < 0 || h >= Hi || w < 0 || w >= Wi) continue;
            const size_t off = img + (size_t(h) * Wi + w) * p.K;
            if (f.g2 != nullptr) ga += *reinterpret_cast<const f32x4*>(f.g2 + off);
            *reinterpret_cast<f32x4*>(f.gz + off) = ga;
            const f32x4 z = (*reinterpret_cast<const f32x4*>(f.y + off) - mean) * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) ga[e] = z[e] > 0.f ? ga[e] : ga[e] * neg;
            s1 += ga;
            s2 += ga * z;
        }
    }
    // the two lanes of a pair hold the sums of the two halves of the tile for the same four channels: add them, the even lane writes
    // the sum of g_z, the odd lane the sum of g_z * z
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s1[e] += swap1(s1[e]);
        s2[e] += swap1(s2[e]);
    }
    float* sp = f.part + size_t(t) * 2 * p.K + quad;
    *reinterpret_cast<f32x4*>(odd ? sp + p.K : sp) = odd ? s2 : s1;
}

// The same fused output transform with the tile spread over the lanes of a wave (round 3; the input-side counterpart is
// wino6_input_coop_kernel): a wave takes one tile x 32 channels.  Lane (a', qg) loads the 8 planes of plane row a' (16 bytes each, 128
// contiguous bytes per plane and instruction) and applies A^T along the row; the 8 x 6 intermediate is transposed through a wave-private
// 6 KB of LDS; lane (c, qg), c < 6, applies A^T down its column and then owns the 6 pixels of tile column c x 4 channels: the reflect
// fold runs along the column in registers and across columns by two lane shuffles (border tiles only), the per-pixel phase moves 16
// bytes per lane on 6 adjacent pixels per instruction.  Same transform arithmetic in the same order as the kernel above (g_a bitwise
// equal); the tile's two sums are added in another order (fixed: bitwise reproducible).  ~100 VGPRs instead of 158.
__global__ __launch_bounds__(256) void wino6_output_inbwd_coop_kernel(const W6Out p, const float* __restrict__ fy, const float* __restrict__ fmean,
                                                                      const float* __restrict__ frstd, const float* __restrict__ fg2,
                                                                      float* __restrict__ fgz, float* __restrict__ fpart, const int fact, const float fslope) {
    constexpr int V = 6, N = 8, MO = 6;
    __shared__ __attribute__((aligned(16))) f32x4 lds[4][MO * N * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hi = lane >> 3, qg = lane & 7;                    // hi = plane row a' in phase 1, tile column c in phase 2
    const int groups = p.K / 32;
    const long long unit = blockIdx.x * 4ll + wave, units = p.T * groups;
    const bool live = unit < units;
    const long long t = live ? unit / groups : 0;
    const int cg = live ? int(unit - t * groups) : 0;
    const int tx = int(t % p.TW);
    const long long r_ = t / p.TW;
    const int ty = int(r_ % p.TH), b = int(r_ / p.TH);
    const int ch = cg * 32 + qg * 4;
    const size_t plane = size_t(p.T) * p.K;
    const float* M = p.M + size_t(t) * p.K + ch;
    f32x4 m[N], tr[MO];
#pragma unroll
    for (int c = 0; c < N; ++c) m[c] = ld4(M + (hi * N + c) * plane);
    W6<V>::at(m, tr);                                           // A^T along plane row a' = hi: 8 -> 6 columns
    f32x4* buf = lds[wave];
#pragma unroll
    for (int c = 0; c < MO; ++c) buf[(c * N + hi) * 8 + qg] = tr[c];
    __syncthreads();
    const int c = hi;                                           // this lane now owns tile column c (lanes with c >= 6 idle)
    const bool col_ok = c < MO;
    f32x4 o[MO];
    {
        f32x4 v[N];
#pragma unroll
        for (int a = 0; a < N; ++a) v[a] = buf[((col_ok ? c : 0) * N + a) * 8 + qg];
        W6<V>::at(v, o);                                        // A^T down the column: o[a] = sum over a' of AT(a, a') tr_a'[c], a' ascending
    }
    const int Hi = p.H - 2, Wi = p.W - 2;                       // interior extent; padded coordinate = interior + 1
    // fold rows (padded line 0 onto 2, Hi + 1 onto Hi - 1): inside the lane
#pragma unroll
    for (int a = 0; a < MO; ++a) {
        const int hp = MO * ty + a;
        if (a >= 2 && hp == 2) o[a] += o[a - 2];
        if (a + 2 < MO && hp == Hi - 1) o[a] += o[a + 2];
    }
    // fold columns: the partner column lives two lane rows (16 lanes) away; only the first / last tile column of the image needs it
    const int wp = MO * tx + c;
    const bool need_l = col_ok && c >= 2 && wp == 2, need_r = col_ok && c + 2 < MO && wp == Wi - 1;
    const bool edge = tx == 0 || (MO * tx <= Wi - 1 && Wi - 1 < MO * tx + MO);      // wave-uniform
    if (edge) {
#pragma unroll
        for (int a = 0; a < MO; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lft = __shfl(o[a][e], (lane + 48) & 63, 64), rgt = __shfl(o[a][e], (lane + 16) & 63, 64);      // lanes - 16 / + 16
                o[a][e] += (need_l ? lft : 0.f) + (need_r ? rgt : 0.f);
            }
    }
    const f32x4 mean = ld4(fmean + size_t(b) * p.K + ch), rstd = ld4(frstd + size_t(b) * p.K + ch);
    const float neg = fact == NIRGAN_ACT_RELU ? 0.f : fact == NIRGAN_ACT_LRELU ? fslope : 1.f;
    const size_t img = size_t(b) * Hi * Wi * p.K + ch;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const int w = MO * tx + c - 1;
    if (live && col_ok && w >= 0 && w < Wi) {
#pragma unroll
        for (int a = 0; a < MO; ++a) {
            const int h = MO * ty + a - 1;
            if (h < 0 || h >= Hi) continue;
            const size_t off = img + (size_t(h) * Wi + w) * p.K;
            f32x4 ga = o[a];
            if (fg2 != nullptr) ga += ld4(fg2 + off);
            st4(fgz + off, ga);
            const f32x4 z = (ld4(fy + off) - mean) * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) ga[e] = z[e] > 0.f ? ga[e] : ga[e] * neg;
            s1 += ga;
            s2 += ga * z;
        }
    }
    // the tile's sums: over the 8 lane rows (columns 6, 7 hold zeros), fixed butterfly
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int d = 8; d < 64; d <<= 1) {
            s1[e] += __shfl_xor(s1[e], d, 64);
            s2[e] += __shfl_xor(s2[e], d, 64);
        }
    }
    if (live && hi < 2) st4(fpart + size_t(t) * 2 * p.K + (hi == 1 ? p.K : 0) + ch, hi == 1 ? s2 : s1);
}

// ------------------------------------------------------------------------------------------------ weight-gradient finish
struct W6Fin { const float* slabs; int nsplit, K, C; float* grad; int accumulate; };
// up to 16 layers of one geometry in one grid (blockIdx.y = layer): the finish of a single layer is 17 us of launch latency for 33 MB
struct W6FinBatch { const float* slabs[16]; float* grad[16]; int nsplit, K, C, accumulate; };

// dW[k][c] = G^T (sum over splits of dU[.][k][c]) G in the reference layout [K][C][R][R]; splits in order (deterministic).
// A block takes 64 (k, c) pairs: thread (e, fg) sums the planes fg, fg + 4, ... of pair e over the splits (coalesced 256-B rows),
// the N * N sums meet in LDS and threads 0-63 apply the R x N / N x R transforms.
template <int V> __device__ __forceinline__ void wino6_wgrad_finish_body(const W6Fin& p);

template <int V>
__global__ __launch_bounds__(256) void wino6_wgrad_finish_kernel(const W6Fin p) { wino6_wgrad_finish_body<V>(p); }

template <int V>
__global__ __launch_bounds__(256) void wino6_wgrad_finish_batch_kernel(const W6FinBatch b) {
    const W6Fin p{b.slabs[blockIdx.y], b.nsplit, b.K, b.C, b.grad[blockIdx.y], b.accumulate};
    wino6_wgrad_finish_body<V>(p);
}

template <int V>
__device__ __forceinline__ void wino6_wgrad_finish_body(const W6Fin& p) {
    constexpr int N = W6<V>::N, R = W6<V>::R, NP = N * N, PER = (NP + 3) / 4;
    __shared__ float u_s[NP][64];
    const int e = threadIdx.x & 63, fg = threadIdx.x >> 6;
    const long long i = blockIdx.x * 64ll + e;
    const size_t kc = size_t(p.K) * p.C;
    const bool ok = i < (long long)kc;
    float sum[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) sum[j] = 0.f;
    if (ok) {
        const float* src = p.slabs + i;                                            // slabs are [plane][split][K][C]
        for (int sp = 0; sp < p.nsplit; ++sp) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int f = fg + 4 * j;
                if (f < NP) sum[j] += src[(size_t(f) * p.nsplit + sp) * kc];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (fg + 4 * j < NP) u_s[fg + 4 * j][e] = sum[j];
    __syncthreads();
    if (fg != 0 || !ok) return;
    float t[R][N];
#pragma unroll
    for (int b = 0; b < N; ++b) {
        float col[N], o[R];
#pragma unroll
        for (int a = 0; a < N; ++a) col[a] = u_s[a * N + b][e];
        W6<V>::gt(col, o);
#pragma unroll
        for (int a = 0; a < R; ++a) t[a][b] = o[a];
    }
    float* g = p.grad + size_t(i) * (R * R);
#pragma unroll
    for (int a = 0; a < R; ++a) {
        float o[R];
        W6<V>::gt(t[a], o);
#pragma unroll
        for (int b = 0; b < R; ++b) {
            if (p.accumulate) g[a * R + b] += o[b]; else g[a * R + b] = o[b];
        }
    }
}

// the descriptors' `r` field selects the variant: 0 / 3 = F(4x4,3x3), 4 = F(4x4,4x4), 6 = F(6x6,3x3)
static inline int w6_r(int r) { return r == 0 ? 3 : r; }
static inline bool w6_known(int v) { return v == 3 || v == 4 || v == 6; }
static inline int w6_filter(int v) { return v == 6 ? 3 : v; }            // filter size
static inline int w6_mo(int v) { return v == 6 ? 6 : 4; }                // outputs per tile and dimension
static inline int w6_np(int v) { const int n = w6_mo(v) + w6_filter(v) - 1; return n * n; }      // planes: 36 / 49 / 64
static inline long long w6_tiles(int B, int H, int W, int v) { const int m = w6_mo(v); return (long long)B * ((H + m - 1) / m) * ((W + m - 1) / m); }

}  // namespace

extern "C" int64_t nirgan_wino6_tiles_r(int B, int H, int W, int r) {
    if (B <= 0 || H <= 0 || W <= 0 || !w6_known(w6_r(r))) return 0;
    return w6_tiles(B, H, W, w6_r(r));
}

extern "C" int64_t nirgan_wino6_tiles(int B, int H, int W) { return nirgan_wino6_tiles_r(B, H, W, 3); }

static int w6_weights_impl(const float* w, int K, int C, int r, int transpose_flip, float* U, void* U3, void* stream);
extern "C" int nirgan_wino6_weights_r(const float* w, int K, int C, int r, int transpose_flip, float* U, void* stream) {
    return w6_weights_impl(w, K, C, r, transpose_flip, U, nullptr, stream);
}
extern "C" int nirgan_wino6_weights_x3(const float* w, int K, int C, int r, int transpose_flip, float* U, void* U3_bf16, void* stream) {
    NG_REQUIRE(U3_bf16 != nullptr, "wino6_weights_x3: null planes");
    return w6_weights_impl(w, K, C, r, transpose_flip, U, U3_bf16, stream);
}
static int w6_weights_impl(const float* w, int K, int C, int r, int transpose_flip, float* U, void* U3, void* stream) {
    r = w6_r(r);
    NG_REQUIRE(w && (U || U3) && K > 0 && C > 0, "wino6_weights: bad arguments");        // (U may be omitted when only the planes are wanted)
    NG_REQUIRE(w6_known(r), "wino6_weights: variant %d (3, 4 or 6)", r);
    W6W p{w, U, K, C, transpose_flip ? 1 : 0, static_cast<unsigned short*>(U3)};
    const long long n = (long long)K * C;
    const dim3 grid(unsigned((n + 255) / 256));
    if (r == 3) hipLaunchKernelGGL(wino6_weight_kernel<3>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else if (r == 4) hipLaunchKernelGGL(wino6_weight_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino6_weight_kernel<6>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino6_weights");
}

extern "C" int nirgan_wino6_weights(const float* w, int K, int C, int transpose_flip, float* U, void* stream) {
    return nirgan_wino6_weights_r(w, K, C, 3, transpose_flip, U, stream);
}

extern "C" int nirgan_wino6_weights_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream) {
    NG_REQUIRE(jobs_device && njobs >= 1 && njobs <= 256 && total_blocks >= 1, "wino6_weights_batch: bad arguments");
    hipLaunchKernelGGL(wino6_weights_batch_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(jobs_device), njobs);
    return nirgan_check_launch("wino6_weights_batch");
}

static int w6_input_impl(const nirgan_wino6_desc* d, const nirgan_wino_dy_desc* y, const float* ny, const float* mean, const float* rstd,
                         int act, float slope, void* stream, const nirgan_in_bwd_desc* nb = nullptr) {
    NG_REQUIRE(d && d->V && (d->x || ny || nb), "wino6_input: null pointer");
    const int v = w6_r(d->r), np = w6_np(v), r = w6_filter(v), mo = w6_mo(v);
    NG_REQUIRE(w6_known(v), "wino6_input: variant %d (3, 4 or 6)", v);
    NG_REQUIRE(r == 3 || (!ny && !nb), "wino6_input: the fused variants exist for the 3x3 filter");
    NG_REQUIRE(!nb || (v == 6 && d->C % 32 == 0), "wino6_input_dy_norm: F(6x6,3x3) with C %% 32 == 0 only");
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->C > 0 && d->C % 4 == 0, "wino6_input: bad shape B=%d H=%d W=%d C=%d", d->B, d->H, d->W, d->C);
    NG_REQUIRE(ny || (d->x_hp == d->H + r - 1 && d->x_wp == d->W + r - 1), "wino6_input: the input must be (H+%d) x (W+%d) (%dx%d for %dx%d)", r - 1, r - 1, d->x_hp, d->x_wp, d->H, d->W);
    NG_REQUIRE(ng_aligned16(d->x) && ng_aligned16(d->V) && ng_aligned16(ny) && ng_aligned16(mean) && ng_aligned16(rstd), "wino6_input: pointers must be 16-byte aligned");
    const long long T = w6_tiles(d->B, d->H, d->W, v);
    NG_REQUIRE(T < (1ll << 31) / d->C, "wino6_input: problem too large for 32-bit tile offsets");
    NG_REQUIRE(d->V_elems >= np * T * d->C, "wino6_input: V workspace too small");
    W6In in;
    in.x = d->x; in.V = d->V; in.B = d->B; in.C = d->C;
    in.x_hp = d->H + r - 1; in.x_wp = d->W + r - 1; in.x_row = in.x_wp * d->C; in.x_img = in.x_hp * in.x_row;
    in.TH = (d->H + mo - 1) / mo; in.TW = (d->W + mo - 1) / mo; in.T = T;
    in.Yt = nullptr; in.yTH = in.yTW = 0; in.yT = 0;
    in.y = ny; in.mean = mean; in.rstd = rstd; in.H = d->H; in.W = d->W; in.act = act; in.slope = slope;
    if (y != nullptr) {
        // the same dY buffer seen twice: zero halo r-1, the data gradient covers (H_dy + r - 1) x (W_dy + r - 1) outputs
        NG_REQUIRE(!ny && (nb || y->dy == d->x) && y->Yt && w6_r(y->r) == v && y->dy_pad == r - 1 && y->B == d->B && y->K == d->C && y->dy_hp == d->x_hp
                   && y->dy_wp == d->x_wp && d->H == y->H + r - 1 && d->W == y->W + r - 1,
                   "wino6_input_dy: the two descriptors do not describe the same output-gradient buffer");
        NG_REQUIRE(ng_aligned16(y->Yt), "wino6_input_dy: pointers must be 16-byte aligned");
        in.yTH = (y->H + mo - 1) / mo; in.yTW = (y->W + mo - 1) / mo; in.yT = (long long)y->B * in.yTH * in.yTW;
        NG_REQUIRE(y->Yt_elems >= np * in.yT * y->K, "wino6_input_dy: Yt workspace too small");
        in.Yt = y->Yt;
    }
    const long long nthreads = T * (d->C / (v == 3 ? 4 : 2));
    const dim3 grid(unsigned((nthreads + 255) / 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // F(6x6,3x3): the 8 x 8 patches spread over the lanes, one wave per (patch, 32 channels).  Measured inside the step (bs 16):
    // dY pass 73.8 -> 61.5 us, plain input 43.6 -> 41.3 us; the normalising variant 41.7 -> 44.5 us, so that one keeps the
    // patch-per-thread kernel unless asked (algo = NIRGAN_W6_PATCH_PER_LANES).  Alone, back to back, both forms move 5.3 TB/s.
    in.nbB = 0;
    if (nb != nullptr) {
        NG_REQUIRE(y != nullptr && nb->norm && nb->y && nb->mean && nb->rstd && nb->ws && (nb->g || nb->g2 || nb->gsum_out), "wino6_input_dy_norm: the instance-norm descriptor needs y, mean, rstd, ws and a gradient");
        NG_REQUIRE(nb->B == d->B && nb->C == d->C && nb->H == y->H && nb->W == y->W, "wino6_input_dy_norm: the instance-norm descriptor describes another tensor");
        NG_REQUIRE(!nb->g || (nb->g_hp == nb->H + 2 * nb->g_pad && nb->g_wp == nb->W + 2 * nb->g_pad), "wino6_input_dy_norm: g geometry mismatch");
        in.nb = in_bwd_params(nb);
        in.nbB = nb->B;
        NG_REQUIRE(nb->ws_elems >= int64_t(nb->B) * in.nb.pchunks * 2 * nb->C + int64_t(nb->B) * 2 * nb->C, "wino6_input_dy_norm: ws too small");
        const dim3 cgrid(unsigned((T * (d->C / 32) + 3) / 4));
        hipLaunchKernelGGL((wino6_input_coop_kernel<6, 2>), cgrid, dim3(256), 0, st, in);
        return nirgan_check_launch("wino6_input_dy_norm");
    }
    const bool coop = (v == 6 || v == 4) && d->C % 32 == 0 && d->algo != NIRGAN_W6_PATCH_PER_THREAD && (!ny || d->algo == NIRGAN_W6_PATCH_PER_LANES);
    if (coop) {
        const dim3 cgrid(unsigned((T * (d->C / 32) + 3) / 4));
        if (ny) hipLaunchKernelGGL((wino6_input_coop_kernel<6, 1>), cgrid, dim3(256), 0, st, in);
        else if (v == 6) hipLaunchKernelGGL((wino6_input_coop_kernel<6, 0>), cgrid, dim3(256), 0, st, in);
        else hipLaunchKernelGGL((wino6_input_coop_kernel<4, 0>), cgrid, dim3(256), 0, st, in);
        return nirgan_check_launch("wino6_input");
    }
    if (ny && v == 3) hipLaunchKernelGGL((wino6_input_kernel<3, 1>), grid, dim3(256), 0, st, in);
    else if (ny) hipLaunchKernelGGL((wino6_input_kernel<6, 1>), grid, dim3(256), 0, st, in);
    else if (v == 3) hipLaunchKernelGGL((wino6_input_kernel<3, 0>), grid, dim3(256), 0, st, in);
    else if (v == 4) hipLaunchKernelGGL((wino6_input_kernel<4, 0>), grid, dim3(256), 0, st, in);
    else hipLaunchKernelGGL((wino6_input_kernel<6, 0>), grid, dim3(256), 0, st, in);
    return nirgan_check_launch("wino6_input");
}

extern "C" int nirgan_wino6_input(const nirgan_wino6_desc* d, void* stream) { return w6_input_impl(d, nullptr, nullptr, nullptr, nullptr, 0, 0.f, stream); }

extern "C" int nirgan_wino6_input_dy(const nirgan_wino6_desc* d, const nirgan_wino_dy_desc* y, void* stream) {
    NG_REQUIRE(y != nullptr, "wino6_input_dy: null pointer");
    return w6_input_impl(d, y, nullptr, nullptr, nullptr, 0, 0.f, stream);
}

extern "C" int nirgan_wino6_input_norm(const nirgan_wino6_desc* d, const float* y, const float* mean, const float* rstd, int act, float slope, void* stream) {
    NG_REQUIRE(y && mean && rstd, "wino6_input_norm: null pointer");
    NG_REQUIRE(act == NIRGAN_ACT_NONE || act == NIRGAN_ACT_RELU || act == NIRGAN_ACT_LRELU, "wino6_input_norm: activation %d", act);
    return w6_input_impl(d, nullptr, y, mean, rstd, act, slope, stream);
}

extern "C" int nirgan_wino6_input_dy_norm(const nirgan_wino6_desc* d, const nirgan_wino_dy_desc* y, const nirgan_in_bwd_desc* n, void* stream) {
    NG_REQUIRE(y != nullptr && n != nullptr, "wino6_input_dy_norm: null pointer");
    return w6_input_impl(d, y, nullptr, nullptr, nullptr, 0, 0.f, stream, n);
}

extern "C" int nirgan_wino6_dy(const nirgan_wino_dy_desc* d, void* stream) {
    NG_REQUIRE(d && d->dy && d->Yt, "wino6_dy: null pointer");
    const int v = w6_r(d->r), mo = w6_mo(v);
    NG_REQUIRE(w6_known(v), "wino6_dy: variant %d (3, 4 or 6)", v);
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->K > 0 && d->K % 4 == 0 && d->dy_pad >= 0, "wino6_dy: bad shape");
    NG_REQUIRE(d->dy_hp == d->H + 2 * d->dy_pad && d->dy_wp == d->W + 2 * d->dy_pad, "wino6_dy: dy geometry mismatch");
    NG_REQUIRE(ng_aligned16(d->dy) && ng_aligned16(d->Yt), "wino6_dy: pointers must be 16-byte aligned");
    W6Dy p;
    p.dy = d->dy; p.Yt = d->Yt; p.B = d->B; p.H = d->H; p.W = d->W; p.K = d->K;
    p.d_row = d->dy_wp * d->K; p.d_img = d->dy_hp * p.d_row; p.d_org = d->dy_pad * p.d_row + d->dy_pad * d->K;
    p.TH = (d->H + mo - 1) / mo; p.TW = (d->W + mo - 1) / mo; p.T = (long long)d->B * p.TH * p.TW;
    NG_REQUIRE(d->Yt_elems >= w6_np(v) * p.T * d->K, "wino6_dy: workspace too small");
    const long long n = p.T * (d->K / (v == 3 ? 4 : 2));
    const dim3 grid(unsigned((n + 255) / 256));
    if (v == 3) hipLaunchKernelGGL(wino6_dy_kernel<3>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else if (v == 4) hipLaunchKernelGGL(wino6_dy_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino6_dy_kernel<6>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino6_dy");
}

// validation + the 32-k form's parameters (one plane as a 1x1 'convolution' over a [1][T] image of C-channel pixels: the direct tile's descriptor)
static int w6_gemm_params(const nirgan_wino6_desc* d, W6Gemm& g, long long& T) {
    NG_REQUIRE(d && (d->U || d->U3) && d->V && d->M && d->zero_page, "wino6_gemm: null pointer");
    const int v = w6_r(d->r), np = w6_np(v);
    NG_REQUIRE(w6_known(v), "wino6_gemm: variant %d (3, 4 or 6)", v);
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->C > 0 && d->C % 4 == 0 && d->K > 64 && d->K % 4 == 0, "wino6_gemm: C %% 4 == 0, K > 64, K %% 4 == 0 (C=%d K=%d)", d->C, d->K);
    T = w6_tiles(d->B, d->H, d->W, v);
    NG_REQUIRE(T * d->C < (1ll << 31) && T * d->K < (1ll << 31), "wino6_gemm: problem too large for 32-bit offsets");
    NG_REQUIRE(d->V_elems >= np * T * d->C && d->M_elems >= np * T * d->K, "wino6_gemm: V / M workspace too small");
    nirgan_conv_desc c = {};
    c.in = d->V; c.in_elems = T * d->C; c.in_hp = 1; c.in_wp = int(T); c.in_cs = d->C; c.run = d->C; c.in_stride = 1;
    c.ntaps = 1;
    // (U may be omitted when its three bf16 planes are given and the split tile takes the launch -- checked by the launchers; the
    // planes' address stands in for the descriptor check)
    c.w = d->U ? d->U : static_cast<const float*>(d->U3); c.w_elems = (long long)d->K * d->C; c.bias = nullptr;
    c.out = d->M; c.out_elems = T * d->K; c.out_hp = 1; c.out_wp = int(T); c.out_cs = d->K; c.out_stride = 1;
    c.B = 1; c.OH = 1; c.OW = int(T); c.N = d->K; c.zero_page = d->zero_page;
    const int rc = ng::build_conv_params(&c, g.p);
    if (rc != NIRGAN_OK) return rc;
    g.in_plane = T * d->C; g.w_plane = (long long)d->K * d->C; g.out_plane = T * d->K;
    g.per_plane = g.p.mtiles * g.p.ntiles; g.total = np * g.per_plane;
    return NIRGAN_OK;
}

// The persistent tile (wino6_gemm16p_kernel) where it measures faster: C = 256 (16 K-steps per tile: 186 -> 165 us at T = 4096, 211 ->
// 196 at T = 4624, 251 -> 236 for the PatchGAN's 49 x [2048 x 256] x [512]).  At C = 512 a tile is twice as long and the uneven last
// round of fixed assignments costs more than the folded epilogue saves (579 vs 536 us for the PatchGAN layer's pair launch).
// (C = 512, the PatchGAN's F(4x4,4x4) data gradient: the plain GEMM gains 12 % as persistent workgroups, the pair launch loses 9 %)
// (the persistent bodies address a plane with 32-bit BYTE offsets, unsigned(row * C + chunk) * 4: T * C and K * C must stay below 2^30;
// larger problems take the one-tile-per-workgroup kernels)
static bool w6_persistent_ok(const nirgan_wino6_desc* d, bool pair = true) {
    const long long T = w6_tiles(d->B, d->H, d->W, w6_r(d->r));
    if (T * d->C >= (1ll << 30) || (long long)d->K * d->C >= (1ll << 30) || T * d->K >= (1ll << 30)) return false;
    return d->C == 256 || (d->C == 512 && !pair);
}

static W6G16 w6_g16_params(const nirgan_wino6_desc* d, long long T) {
    W6G16 q;
    q.A = d->V; q.Bw = d->U; q.Out = d->M; q.zero = d->zero_page; q.T = int(T); q.C = d->C; q.K = d->K;
    q.mtiles = int((T + 127) / 128); q.ntiles = (d->K + 127) / 128; q.per_plane = q.mtiles * q.ntiles; q.total = w6_np(w6_r(d->r)) * q.per_plane;
    q.a_plane = T * d->C; q.b_plane = (long long)d->K * d->C; q.o_plane = T * d->K;
    return q;
}

// which kernel a descriptor's plane GEMMs run on (one place: the launchers and the name queries read it)
enum W6Choice { W6_PERSIST32, W6_PERSIST16K, W6_ONE_TILE16, W6_DIRECT };
static W6Choice w6_gemm_choice(const nirgan_wino6_desc* d, bool pair) {
    if (d->C % 16 != 0 || d->algo == NIRGAN_W6_DIRECT_TILE) return W6_DIRECT;
    if (pair) {                                        // the pair launch has the persistent 32-k form and the direct-tile form
        return (w6_persistent_ok(d, true) && d->algo != NIRGAN_W6_ONE_TILE) ? W6_PERSIST32 : W6_DIRECT;
    }
    if (w6_persistent_ok(d, false) && d->algo != NIRGAN_W6_ONE_TILE) return d->algo == NIRGAN_W6_PERSIST16 ? W6_PERSIST16K : W6_PERSIST32;
    return W6_ONE_TILE16;
}

// precision 3 for the plane GEMMs: the three bf16 planes of U are there and the split tile's shapes apply
static bool w6_x3(const nirgan_wino6_desc* d, const W6Gemm& g) {
    return d->U3 != nullptr && d->C % 32 == 0 && d->K % 64 == 0 && g.p.off32 && (long long)w6_np(w6_r(d->r)) * d->K * d->C * 2 < (1ll << 40);
}
static bool w6_x3_desc(const nirgan_wino6_desc* d) {
    W6Gemm g;
    long long T;
    return d && d->U3 != nullptr && w6_gemm_params(d, g, T) == NIRGAN_OK && w6_x3(d, g);
}

extern "C" const char* nirgan_wino6_gemm_kernel_name(const nirgan_wino6_desc* d) {
    if (!d) return "";
    if (w6_x3_desc(d)) return d->K % 128 == 0 ? (d->algo == NIRGAN_W6_X3_R4 ? "conv_x3r_kernel<128> (planes)" : "conv_x3_kernel<128> (planes)") : "conv_x3_kernel<64> (planes)";
    if (d->algo == NIRGAN_W6_TILE256 && d->C % 32 == 0 && d->K % 256 == 0) return "wino6_gemm256_kernel";
    switch (w6_gemm_choice(d, false)) {
        case W6_PERSIST32: return "wino6_gemm32p_kernel";
        case W6_PERSIST16K: return "wino6_gemm16p_kernel";
        case W6_ONE_TILE16: return "wino6_gemm16_kernel";
        default: return "wino6_gemm_kernel";
    }
}

// the weight-gradient half of a pair launch walks its units persistently when it has the plane-matrix form
static bool w6_pair_wgrad_persistent(const ng::WgradParams& wp, const nirgan_wgrad_desc* w) {
    return ng::wgrad_persist_ok(wp) && ng::wgrad_matrix_form(wp) && w->algo != NIRGAN_WGRAD_ONE_UNIT;
}

extern "C" const char* nirgan_wino6_pair_kernel_name(const nirgan_wino6_desc* d, const nirgan_wgrad_desc* w) {
    if (!d || !w) return "";
    ng::WgradParams wp;
    if (ng::build_wgrad_params(w, wp) != NIRGAN_OK) return "";
    if (w->N <= 64 || wp.prec != 0 || wp.pq_bf16 || w6_x3_desc(d)) return "";                      // two ordinary launches
    if (w6_gemm_choice(d, true) == W6_PERSIST32) return w6_pair_wgrad_persistent(wp, w) ? "wino6_pair16p_kernel" : "wino6_pair16_kernel";
    return "wino6_pair_kernel";
}

extern "C" int nirgan_wino6_gemm_wgrad_pair(const nirgan_wino6_desc* d, const nirgan_wgrad_desc* w, void* stream) {
    W6Gemm g;
    long long T;
    int rc = w6_gemm_params(d, g, T);
    if (rc != NIRGAN_OK) return rc;
    ng::WgradParams wp;
    rc = ng::build_wgrad_params(w, wp);
    if (rc != NIRGAN_OK) return rc;
    if (w->N <= 64 || wp.prec != 0 || wp.pq_bf16 || w6_x3(d, g)) {           // not the wide fp32 tile, or the three-term split tiles: two ordinary launches
        rc = nirgan_wino6_gemm(d, stream);
        return rc != NIRGAN_OK ? rc : nirgan_wgrad_igemm(w, stream);
    }
    NG_REQUIRE(d->U != nullptr, "wino6_gemm_wgrad_pair: U is required unless the three-term split tile takes the launch");
    const int wgrad_blocks = wp.ntiles_n * wp.ntiles_k * wp.nsplit * wp.nplanes;
    if (w6_gemm_choice(d, true) == W6_PERSIST32) {
        NG_REQUIRE(ng_aligned16(d->U) && ng_aligned16(d->V) && ng_aligned16(d->M) && ng_aligned16(d->zero_page), "wino6_gemm_wgrad_pair: pointers must be 16-byte aligned");
        const W6G16 q = w6_g16_params(d, T);
        const int gemm_blocks = q.total < 512 ? q.total : 512;
        // the persistent weight-gradient walk needs the plane-matrix form (one tap, unit stride, one image row) and rows in every split
        if (w6_pair_wgrad_persistent(wp, w)) {
            hipLaunchKernelGGL(wino6_pair16p_kernel, dim3(gemm_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), q, gemm_blocks, wp);
            return nirgan_check_launch("wino6_gemm_wgrad_pair");
        }
        hipLaunchKernelGGL(wino6_pair16_kernel, dim3(gemm_blocks + wgrad_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), q, gemm_blocks, wp, wgrad_blocks);
        return nirgan_check_launch("wino6_gemm_wgrad_pair");
    }
    hipLaunchKernelGGL(wino6_pair_kernel, dim3(g.total + wgrad_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g, wp, wgrad_blocks, 1);
    return nirgan_check_launch("wino6_gemm_wgrad_pair");
}

extern "C" int nirgan_wino6_gemm(const nirgan_wino6_desc* d, void* stream) {
    W6Gemm g;
    long long T;
    const int rc0 = w6_gemm_params(d, g, T);
    if (rc0 != NIRGAN_OK) return rc0;
    const W6Choice ch = w6_gemm_choice(d, false);
    if (w6_x3(d, g)) {
        // precision 3: the plane GEMMs on the bf16 pipe, V split in the kernel, U from its three bf16 planes (nirgan_wino6_weights_x3)
        g.p.prec = 3;
        g.p.w3 = static_cast<const unsigned short*>(d->U3);
        g.p.w3_plane = (long long)w6_np(w6_r(d->r)) * d->K * d->C;
        g.p.algo = d->algo == NIRGAN_W6_X3_R4 ? NIRGAN_CONV_X3_R4 : 0;       // (A/B: the four-wave register-fed tile)
#ifdef NG_X3_DIAG
        g.p.algo = d->algo & 0xf00;          // diagnostic build only: the epilogue switches of igemm_x3.h
#endif
        return ng::ng_launch_conv_x3(&g.p, 1, d->K % 128 == 0 ? 128 : 64, w6_np(w6_r(d->r)), g.in_plane, g.w_plane, g.out_plane,
                                     static_cast<hipStream_t>(stream), "wino6_gemm (three-term split tile)");
    }
    NG_REQUIRE(d->U != nullptr, "wino6_gemm: U is required unless the three-term split tile takes the launch (U3, C %% 32 == 0, K %% 64 == 0)");
    if (d->algo == NIRGAN_W6_TILE256 && d->C % 32 == 0 && d->K % 256 == 0 && g.p.off32) {
        const int per_plane = ((g.p.M + 255) >> 8) * (g.p.N >> 8), total = w6_np(w6_r(d->r)) * per_plane;
        const int cus = ng::ng_cu_count_conv();
        hipLaunchKernelGGL(wino6_gemm256_kernel, dim3(total < cus ? total : cus), dim3(512), 0, static_cast<hipStream_t>(stream), g, per_plane, total);
        return nirgan_check_launch("wino6_gemm (256-wide tile)");
    }
    if (ch == W6_DIRECT) {
        hipLaunchKernelGGL(wino6_gemm_kernel, dim3(g.total), dim3(256), 0, static_cast<hipStream_t>(stream), g);
        return nirgan_check_launch("wino6_gemm");
    }
    NG_REQUIRE(ng_aligned16(d->U) && ng_aligned16(d->V) && ng_aligned16(d->M) && ng_aligned16(d->zero_page), "wino6_gemm: pointers must be 16-byte aligned");
    const W6G16 q = w6_g16_params(d, T);
    const int grid = q.total < 512 ? q.total : 512;           // persistent workgroups, epilogue folded into the next tile's K loop: 2 per CU
    // 32-k stages (half the barriers per product; 76 KB of LDS, still two workgroups per CU): 145.8 -> 141.7 us once the loader
    // carries no vector arithmetic (before that: no difference)
    if (ch == W6_PERSIST32) hipLaunchKernelGGL(wino6_gemm32p_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), q);
    else if (ch == W6_PERSIST16K) hipLaunchKernelGGL(wino6_gemm16p_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), q);
    else hipLaunchKernelGGL(wino6_gemm16_kernel, dim3(q.total), dim3(256), 0, static_cast<hipStream_t>(stream), q);       // 16-k stages, up to four resident workgroups per CU
    return nirgan_check_launch("wino6_gemm");
}

extern "C" int nirgan_wino6_output(const nirgan_wino6_desc* d, void* stream) {
    NG_REQUIRE(d && d->M && (d->y || d->fuse_gz), "wino6_output: null pointer");
    const int v = w6_r(d->r), mo = w6_mo(v);
    NG_REQUIRE(w6_known(v), "wino6_output: variant %d (3, 4 or 6)", v);
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->K > 0 && d->K % 4 == 0, "wino6_output: bad shape");
    NG_REQUIRE(ng_aligned16(d->M) && ng_aligned16(d->y) && ng_aligned16(d->bias), "wino6_output: pointers must be 16-byte aligned");
    const long long T = w6_tiles(d->B, d->H, d->W, v);
    NG_REQUIRE(d->M_elems >= w6_np(v) * T * d->K, "wino6_output: M workspace too small");
    NG_REQUIRE(!d->stats_ws || (d->stats_ws_elems >= T * 4 * d->K && ng_aligned16(d->stats_ws)), "wino6_output: stats_ws too small (T * 4 * K floats) or unaligned");
    if (d->fuse_gz != nullptr) {
        // the data gradient's output transform + the first pass of the consumer's instance-norm backward
        NG_REQUIRE(v == 6 && !d->bias && !d->stats_ws, "wino6_output: the fused instance-norm backward pass is for F(6x6,3x3) data gradients (no bias, no forward statistics)");
        NG_REQUIRE(d->fuse_y && d->fuse_mean && d->fuse_rstd && d->fuse_part, "wino6_output: fuse_y / fuse_mean / fuse_rstd / fuse_part required with fuse_gz");
        NG_REQUIRE(ng_aligned16(d->fuse_y) && ng_aligned16(d->fuse_mean) && ng_aligned16(d->fuse_rstd) && ng_aligned16(d->fuse_g2) && ng_aligned16(d->fuse_gz) && ng_aligned16(d->fuse_part),
                   "wino6_output: fuse pointers must be 16-byte aligned");
        NG_REQUIRE(d->H >= 6 && d->W >= 6 && (d->H - 3) / mo == (d->H - 1) / mo && (d->W - 3) / mo == (d->W - 1) / mo,
                   "wino6_output: the far halo line and its fold partner must share a tile (H=%d W=%d tile %d)", d->H, d->W, mo);
        NG_REQUIRE(d->fuse_part_elems >= T * 2 * d->K, "wino6_output: fuse_part too small");
        NG_REQUIRE(d->fuse_act == NIRGAN_ACT_NONE || d->fuse_act == NIRGAN_ACT_RELU || d->fuse_act == NIRGAN_ACT_LRELU, "wino6_output: fuse_act %d", d->fuse_act);
        W6Out pf{d->M, nullptr, nullptr, d->B, d->H, d->W, d->K, (d->H + mo - 1) / mo, (d->W + mo - 1) / mo, T, nullptr};
        const long long nf = T * (d->K / (v == 3 ? 4 : 2));
        const dim3 gridf(unsigned((nf + 255) / 256));
        hipStream_t sf = static_cast<hipStream_t>(stream);
        if (d->K % 32 == 0 && d->algo != NIRGAN_W6_PATCH_PER_THREAD) {      // the tile spread over a wave's lanes
            const dim3 cgrid(unsigned((T * (d->K / 32) + 3) / 4));
            hipLaunchKernelGGL(wino6_output_inbwd_coop_kernel, cgrid, dim3(256), 0, sf, pf, d->fuse_y, d->fuse_mean, d->fuse_rstd, d->fuse_g2, d->fuse_gz, d->fuse_part, d->fuse_act, d->fuse_slope);
            return nirgan_check_launch("wino6_output");
        }
        hipLaunchKernelGGL(wino6_output_inbwd_kernel<6>, gridf, dim3(256), 0, sf, pf, d->fuse_y, d->fuse_mean, d->fuse_rstd, d->fuse_g2, d->fuse_gz, d->fuse_part, d->fuse_act, d->fuse_slope);
        return nirgan_check_launch("wino6_output");
    }
    W6Out p{d->M, d->bias, d->y, d->B, d->H, d->W, d->K, (d->H + mo - 1) / mo, (d->W + mo - 1) / mo, T, d->stats_ws};
    // (the plain output transform keeps the tile-per-thread kernel: its lane-spread form measured 39.8 against 35.3 us per launch)
    const long long n = T * (d->K / (v == 3 ? 4 : 2));
    const dim3 grid(unsigned((n + 255) / 256));
    if (v == 3) hipLaunchKernelGGL(wino6_output_kernel<3>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else if (v == 4) hipLaunchKernelGGL(wino6_output_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino6_output_kernel<6>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino6_output");
}

extern "C" int nirgan_wino6_conv3x3(const nirgan_wino6_desc* d, void* stream) {
    int rc = nirgan_wino6_input(d, stream);
    if (rc == NIRGAN_OK) rc = nirgan_wino6_gemm(d, stream);
    return rc != NIRGAN_OK ? rc : nirgan_wino6_output(d, stream);
}

extern "C" int nirgan_wino6_wgrad_finish_r(const float* slabs, int nsplit, int K, int C, int r, float* grad, int accumulate, void* stream) {
    r = w6_r(r);
    NG_REQUIRE(slabs && grad && nsplit >= 1 && K > 0 && C > 0, "wino6_wgrad_finish: bad arguments");
    NG_REQUIRE(w6_known(r), "wino6_wgrad_finish: variant %d (3, 4 or 6)", r);
    W6Fin p{slabs, nsplit, K, C, grad, accumulate ? 1 : 0};
    const long long n = (long long)K * C;
    const dim3 grid(unsigned((n + 63) / 64));
    if (r == 3) hipLaunchKernelGGL(wino6_wgrad_finish_kernel<3>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else if (r == 4) hipLaunchKernelGGL(wino6_wgrad_finish_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino6_wgrad_finish_kernel<6>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino6_wgrad_finish");
}

extern "C" int nirgan_wino6_wgrad_finish_batch(const float* const* slabs, float* const* grads, int n, int nsplit, int K, int C, int r, int accumulate, void* stream) {
    r = w6_r(r);
    NG_REQUIRE(slabs && grads && n >= 1 && n <= 16 && nsplit >= 1 && K > 0 && C > 0, "wino6_wgrad_finish_batch: bad arguments (1..16 layers)");
    NG_REQUIRE(w6_known(r), "wino6_wgrad_finish_batch: variant %d (3, 4 or 6)", r);
    W6FinBatch b;
    for (int i = 0; i < 16; ++i) {
        NG_REQUIRE(i >= n || (slabs[i] && grads[i]), "wino6_wgrad_finish_batch: null pointer in layer %d", i);
        b.slabs[i] = slabs[i < n ? i : 0];
        b.grad[i] = grads[i < n ? i : 0];
    }
    b.nsplit = nsplit; b.K = K; b.C = C; b.accumulate = accumulate ? 1 : 0;
    const long long kc = (long long)K * C;
    const dim3 grid(unsigned((kc + 63) / 64), unsigned(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (r == 3) hipLaunchKernelGGL(wino6_wgrad_finish_batch_kernel<3>, grid, dim3(256), 0, st, b);
    else if (r == 4) hipLaunchKernelGGL(wino6_wgrad_finish_batch_kernel<4>, grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL(wino6_wgrad_finish_batch_kernel<6>, grid, dim3(256), 0, st, b);
    return nirgan_check_launch("wino6_wgrad_finish_batch");
}

extern "C" int nirgan_wino6_wgrad_finish(const float* slabs, int nsplit, int K, int C, float* grad, int accumulate, void* stream) {
    return nirgan_wino6_wgrad_finish_r(slabs, nsplit, K, C, 3, grad, accumulate, stream);
}
