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

struct WinoW { const float* w; float* U; int K, C, flip; };

// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__device__ __forceinline__ void wino_weight_one(const WinoW& p, const long long i) {
    if (i >= (long long)p.K * p.C) return;
    const int k = int(i / p.C), c = int(i - (long long)k * p.C);
    // flip: the data-gradient filter g'[k][c][i][j] = W[c][k][2-i][2-j] (W stored [C][K][3][3]: rows are the forward OUTPUT channels)
    const float* g = p.flip ? p.w + (size_t(c) * p.K + k) * 9 : p.w + (size_t(k) * p.C + c) * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = p.flip ? g[8 - j] : g[j], g1 = p.flip ? g[5 - j] : g[3 + j], g2 = p.flip ? g[2 - j] : g[6 + j];
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

__global__ __launch_bounds__(256) void wino_weight_kernel(const WinoW p) { wino_weight_one(p, blockIdx.x * 256ll + threadIdx.x); }

struct WinoIn { const float* x; float* V; int B, H, W, C, x_row, x_img, TH, TW; long long T;
                float* Yt; int yTH, yTW; long long yT; };   // Yt != nullptr: x is a dY buffer (zero halo r-1) and the tile's lower-right 2x2 block is ALSO
                                                             // emitted as Yt = A dY A^T of output-gradient tile (ty, tx) (one read of dY for both transforms)

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
    // odd H / W: the last tile row / column reaches one line past the halo; those lines feed only outputs that are never stored
    const int amax = p.H + 2 - 2 * ty, cmax = p.W + 2 - 2 * tx;      // valid lines / columns of this tile (4, or 3 at an odd edge)
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            d[a][c] = (a < amax && c < cmax) ? *reinterpret_cast<const f32x4*>(src + size_t(a) * p.x_row + size_t(c) * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (p.Yt != nullptr && ty < p.yTH && tx < p.yTW) {
        // dY rows 2ty, 2ty+1 / columns 2tx, 2tx+1 are d[2..3][2..3] (rows past an odd extent read the zero halo)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 r0[2] = {d[2][2], d[2][3]}, r1[2] = {d[2][2] + d[3][2], d[2][3] + d[3][3]};
        const f32x4 r2[2] = {d[2][2] - d[3][2], d[2][3] - d[3][3]}, r3[2] = {z - d[3][2], z - d[3][3]};
        const size_t yplane = size_t(p.yT) * p.C;
        float* Y = p.Yt + ((size_t(b) * p.yTH + ty) * p.yTW + tx) * p.C + q * 4;
        auto row = [&](int a, const f32x4* rr) {
            *reinterpret_cast<f32x4*>(Y + (a * 4 + 0) * yplane) = rr[0];
            *reinterpret_cast<f32x4*>(Y + (a * 4 + 1) * yplane) = rr[0] + rr[1];
            *reinterpret_cast<f32x4*>(Y + (a * 4 + 2) * yplane) = rr[0] - rr[1];
            *reinterpret_cast<f32x4*>(Y + (a * 4 + 3) * yplane) = z - rr[1];
        };
        row(0, r0); row(1, r1); row(2, r2); row(3, r3);
    }
}

struct WinoInN { const float* y; const float* mean; const float* rstd; float* V; int B, H, W, C, TH, TW; long long T; int act; float slope; };

// the same transform straight from a convolution's raw output y (dense [B][H][W][C]): x = act((y - mean) * rstd) with a REFLECT halo
// of 1, evaluated on the fly -- the instance-norm apply pass and the halo'd activation buffer of a layer whose only consumer is this
// transform (first convolution of a ResnetBlock -> second) are never written.  Same arithmetic as in_apply_kernel (bitwise equal V).
__global__ __launch_bounds__(256) void wino_input_norm_kernel(const WinoInN p) {
    const int q4 = p.C / 4;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(p.mean + size_t(b) * p.C + q * 4);
    const f32x4 rstd = *reinterpret_cast<const f32x4*>(p.rstd + size_t(b) * p.C + q * 4);
    const float* yb = p.y + size_t(b) * p.H * p.W * p.C + q * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int rb = 2 * ty + a;                                   // row of the virtual halo'd buffer: 0 .. H+1 (beyond: zero, odd extents)
        const int yr = ng_reflect(rb - 1, p.H);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cb = 2 * tx + c;
            const int yc = ng_reflect(cb - 1, p.W);
            if (rb < p.H + 2 && cb < p.W + 2) {
                f32x4 v = (*reinterpret_cast<const f32x4*>(yb + (size_t(yr) * p.W + yc) * p.C) - mean) * rstd;
                if (p.act == NIRGAN_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                } else if (p.act == NIRGAN_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                }
                d[a][c] = v;
            } else {
                d[a][c] = z;
            }
        }
    }
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

struct WinoDy { const float* dy; float* Yt; int B, H, W, K, d_row, d_img, d_org, TH, TW; long long T; };

// Yt[f][t][k] = (A dY A^T)[f] for the 2x2 output-gradient tile t; A = [[1,0],[1,1],[1,-1],[0,-1]].  One thread = one tile x 4 channels.
__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoDy p) {
    const int q4 = p.K / 4;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const float* src = p.dy + size_t(b) * p.d_img + p.d_org + size_t(2 * ty) * p.d_row + size_t(2 * tx) * p.K + q * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const bool h1 = 2 * ty + 1 < p.H, w1 = 2 * tx + 1 < p.W;                      // odd extents: the last half tile
    const f32x4 d00 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 d01 = w1 ? *reinterpret_cast<const f32x4*>(src + p.K) : z;
    const f32x4 d10 = h1 ? *reinterpret_cast<const f32x4*>(src + p.d_row) : z;
    const f32x4 d11 = (h1 && w1) ? *reinterpret_cast<const f32x4*>(src + p.d_row + p.K) : z;
    const f32x4 r0[2] = {d00, d01}, r1[2] = {d00 + d10, d01 + d11}, r2[2] = {d00 - d10, d01 - d11}, r3[2] = {z - d10, z - d11};
    const size_t plane = size_t(p.T) * p.K;
    float* Y = p.Yt + size_t(t) * p.K + q * 4;
    auto row = [&](int a, const f32x4* rr) {
        *reinterpret_cast<f32x4*>(Y + (a * 4 + 0) * plane) = rr[0];
        *reinterpret_cast<f32x4*>(Y + (a * 4 + 1) * plane) = rr[0] + rr[1];
        *reinterpret_cast<f32x4*>(Y + (a * 4 + 2) * plane) = rr[0] - rr[1];
        *reinterpret_cast<f32x4*>(Y + (a * 4 + 3) * plane) = z - rr[1];
    };
    row(0, r0); row(1, r1); row(2, r2); row(3, r3);
}

struct WinoFin { const float* slabs; int nsplit, K, C; float* grad; int accumulate; };

// dW[k][c] = G^T (sum over splits of dU[.][k][c]) G, written in the reference layout [K][C][3][3]; splits in order (deterministic).
// A block takes 64 (k, c) pairs: thread (e, fg) sums frequencies 4 fg .. 4 fg + 3 of pair e over the splits (coalesced 256-B rows,
// four independent chains), the 16 sums meet in LDS and threads 0-63 apply the 3x4 / 4x3 transforms.
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const WinoFin p) {
    __shared__ float u_s[16][64];
    const int e = threadIdx.x & 63, fg = threadIdx.x >> 6;
    const long long i = blockIdx.x * 64ll + e;
    const size_t kc = size_t(p.K) * p.C;
    const bool ok = i < (long long)kc;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const float* src = p.slabs + size_t(fg * 4) * p.nsplit * kc + i;
        for (int sp = 0; sp < p.nsplit; ++sp) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[j] += src[(size_t(j) * p.nsplit + sp) * kc];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) u_s[fg * 4 + j][e] = sum[j];
    __syncthreads();
    if (fg != 0 || !ok) return;
    float u[4][4];
#pragma unroll
    for (int f = 0; f < 16; ++f) u[f >> 2][f & 3] = u_s[f][e];
    // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
    float t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
        t[1][b] = 0.5f * (u[1][b] - u[2][b]);
        t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
    float* g = p.grad + size_t(i) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float g0 = t[a][0] + 0.5f * (t[a][1] + t[a][2]), g1 = 0.5f * (t[a][1] - t[a][2]), g2 = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
        if (p.accumulate) { g[a * 3] += g0; g[a * 3 + 1] += g1; g[a * 3 + 2] += g2; }
        else { g[a * 3] = g0; g[a * 3 + 1] = g1; g[a * 3 + 2] = g2; }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// F(2x2, 4x4): nn.Conv2d(C, K, 4, stride 1) of the PatchGAN (model/networks.py:573-579), 25 products per 2x2 outputs instead of 64.
// Cook-Toom over the points 0, 1, -1, -1/2, inf; row 0 of G / B^T rescaled so that B^T (applied to the activations) holds dyadic
// constants only.  y = A^T [(G g G^T) . (B^T d B)] A over 5x5 input tiles at stride 2.
__host__ __device__ constexpr float w4_G(int i, int j) {
    constexpr float m[5][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f / 3, 1.f / 3, 1.f / 3, 1.f / 3}, {-1.f, 1.f, -1.f, 1.f},
                               {8.f / 3, -4.f / 3, 2.f / 3, -1.f / 3}, {0.f, 0.f, 0.f, 1.f}};
    return m[i][j];
}
__host__ __device__ constexpr float w4_BT(int i, int j) {
    constexpr float m[5][5] = {{1.f, 2.f, -1.f, -2.f, 0.f}, {0.f, .5f, 1.5f, 1.f, 0.f}, {0.f, -.5f, -.5f, 1.f, 0.f},
                               {0.f, -1.f, 0.f, 1.f, 0.f}, {0.f, -.5f, -1.f, .5f, 1.f}};
    return m[i][j];
}
__host__ __device__ constexpr float w4_AT(int i, int j) {
    constexpr float m[2][5] = {{1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -.5f, 1.f}};
    return m[i][j];
}
// acc += c * x with the constant folded after unrolling (0: nothing, +-1: add / subtract)
template <typename T>
__device__ __forceinline__ void wmac(T& acc, const float c, const T& x) {
    if (c == 0.f) return;
    if (c == 1.f) acc += x;
    else if (c == -1.f) acc -= x;
    else acc += c * x;
}

__device__ __forceinline__ void wino4_weight_one(const WinoW& p, const long long i) {
    if (i >= (long long)p.K * p.C) return;
    const int k = int(i / p.C), c = int(i - (long long)k * p.C);
    // flip: the data-gradient filter g'[k][c][i][j] = W[c][k][3-i][3-j]
    const float* g = p.flip ? p.w + (size_t(c) * p.K + k) * 16 : p.w + (size_t(k) * p.C + c) * 16;
    float gv[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) gv[a][b] = p.flip ? g[15 - (a * 4 + b)] : g[a * 4 + b];
    float t[5][4];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) wmac(s, w4_G(a, r), gv[r][b]);
            t[a][b] = s;
        }
    const size_t plane = size_t(p.K) * p.C;
    float* U = p.U + size_t(k) * p.C + c;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) wmac(s, w4_G(b, r), t[a][r]);
            U[(a * 5 + b) * plane] = s;
        }
}

__global__ __launch_bounds__(256) void wino4_weight_kernel(const WinoW p) { wino4_weight_one(p, blockIdx.x * 256ll + threadIdx.x); }

// every weight transform of a plan in one launch: jobs in device memory (8 x int64 each): w, U, K, C, r, transpose_flip, first_block, 0
__global__ __launch_bounds__(256) void wino_weights_batch_kernel(const long long* __restrict__ jobs, int njobs) {
    int j = 0;
    for (int i = 1; i < njobs; ++i)
        if (int(blockIdx.x) >= int(jobs[i * 8 + 6])) j = i;
    const long long* J = jobs + j * 8;
    const WinoW p{reinterpret_cast<const float*>(J[0]), reinterpret_cast<float*>(J[1]), int(J[2]), int(J[3]), int(J[5])};
    const long long i = (long long)(int(blockIdx.x) - int(J[6])) * 256 + threadIdx.x;
    if (J[4] == 4) wino4_weight_one(p, i); else wino_weight_one(p, i);
}

// one thread = one 5x5 tile x 4 channels; the buffer holds (H+3) x (W+3) pixels for H x W outputs
__global__ __launch_bounds__(256) void wino4_input_kernel(const WinoIn p) {
    const int q4 = p.C / 4;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const float* src = p.x + size_t(b) * p.x_img + size_t(2 * ty) * p.x_row + size_t(2 * tx) * p.C + q * 4;
    const int amax = p.H + 3 - 2 * ty, cmax = p.W + 3 - 2 * tx;      // valid lines / columns of this tile (5, or 4 at an odd edge)
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 m[5][5];
    f32x4 dd[2][2];                            // the patch's lower-right 2x2 block = output-gradient tile (ty, tx) when x is a dY buffer
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        f32x4 d[5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
            d[a] = (a < amax && c < cmax) ? *reinterpret_cast<const f32x4*>(src + size_t(a) * p.x_row + size_t(c) * p.C) : z;
        if (c >= 3) { dd[0][c - 3] = d[3]; dd[1][c - 3] = d[4]; }
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            f32x4 s = z;
#pragma unroll
            for (int a = 0; a < 5; ++a) wmac(s, w4_BT(f, a), d[a]);
            m[f][c] = s;
        }
    }
    const size_t plane = size_t(p.T) * p.C;
    float* V = p.V + size_t(t) * p.C + q * 4;
#pragma unroll
    for (int f1 = 0; f1 < 5; ++f1)
#pragma unroll
        for (int f2 = 0; f2 < 5; ++f2) {
            f32x4 s = z;
#pragma unroll
            for (int c = 0; c < 5; ++c) wmac(s, w4_BT(f2, c), m[f1][c]);
            *reinterpret_cast<f32x4*>(V + (f1 * 5 + f2) * plane) = s;
        }
    if (p.Yt != nullptr && ty < p.yTH && tx < p.yTW) {
        f32x4 rr[5][2];
#pragma unroll
        for (int f = 0; f < 5; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 s = z;
#pragma unroll
                for (int a = 0; a < 2; ++a) wmac(s, w4_AT(a, f), dd[a][c]);
                rr[f][c] = s;
            }
        const size_t yplane = size_t(p.yT) * p.C;
        float* Y = p.Yt + ((size_t(b) * p.yTH + ty) * p.yTW + tx) * p.C + q * 4;
#pragma unroll
        for (int f1 = 0; f1 < 5; ++f1)
#pragma unroll
            for (int f2 = 0; f2 < 5; ++f2) {
                f32x4 s = z;
#pragma unroll
                for (int c = 0; c < 2; ++c) wmac(s, w4_AT(c, f2), rr[f1][c]);
                *reinterpret_cast<f32x4*>(Y + (f1 * 5 + f2) * yplane) = s;
            }
    }
}

// Yt[f][t][k] = (A dY A^T)[f], A = (A^T)^T is 5 x 2
__global__ __launch_bounds__(256) void wino4_dy_kernel(const WinoDy p) {
    const int q4 = p.K / 4;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= p.T * q4) return;
    const long long t = i / q4;
    const int q = int(i - t * q4);
    const int tx = int(t % p.TW);
    const long long r = t / p.TW;
    const int ty = int(r % p.TH), b = int(r / p.TH);
    const float* src = p.dy + size_t(b) * p.d_img + p.d_org + size_t(2 * ty) * p.d_row + size_t(2 * tx) * p.K + q * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const bool h1 = 2 * ty + 1 < p.H, w1 = 2 * tx + 1 < p.W;
    f32x4 d[2][2];
    d[0][0] = *reinterpret_cast<const f32x4*>(src);
    d[0][1] = w1 ? *reinterpret_cast<const f32x4*>(src + p.K) : z;
    d[1][0] = h1 ? *reinterpret_cast<const f32x4*>(src + p.d_row) : z;
    d[1][1] = (h1 && w1) ? *reinterpret_cast<const f32x4*>(src + p.d_row + p.K) : z;
    f32x4 rr[5][2];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 s = z;
#pragma unroll
            for (int a = 0; a < 2; ++a) wmac(s, w4_AT(a, f), d[a][c]);
            rr[f][c] = s;
        }
    const size_t plane = size_t(p.T) * p.K;
    float* Y = p.Yt + size_t(t) * p.K + q * 4;
#pragma unroll
    for (int f1 = 0; f1 < 5; ++f1)
#pragma unroll
        for (int f2 = 0; f2 < 5; ++f2) {
            f32x4 s = z;
#pragma unroll
            for (int c = 0; c < 2; ++c) wmac(s, w4_AT(c, f2), rr[f1][c]);
            *reinterpret_cast<f32x4*>(Y + (f1 * 5 + f2) * plane) = s;
        }
}

// dW[k][c] = G^T (sum over splits of dU[.][k][c]) G in the reference layout [K][C][4][4]; 64 (k, c) pairs per block, thread (e, fg) sums
// the frequencies fg, fg + 4, ... of pair e over the splits in order
__global__ __launch_bounds__(256) void wino4_wgrad_finish_kernel(const WinoFin p) {
    __shared__ float u_s[25][64];
    const int e = threadIdx.x & 63, fg = threadIdx.x >> 6;
    const long long i = blockIdx.x * 64ll + e;
    const size_t kc = size_t(p.K) * p.C;
    const bool ok = i < (long long)kc;
    for (int f = fg; f < 25; f += 4) {
        float s0 = 0.f, s1 = 0.f;
        if (ok) {
            const float* src = p.slabs + size_t(f) * p.nsplit * kc + i;
            int sp = 0;
            for (; sp + 1 < p.nsplit; sp += 2) { s0 += src[size_t(sp) * kc]; s1 += src[size_t(sp + 1) * kc]; }
            if (sp < p.nsplit) s0 += src[size_t(sp) * kc];
        }
        u_s[f][e] = s0 + s1;
    }
    __syncthreads();
    if (fg != 0 || !ok) return;
    float t[4][5];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            float s = 0.f;
#pragma unroll
            for (int f = 0; f < 5; ++f) wmac(s, w4_G(f, a), u_s[f * 5 + b][e]);
            t[a][b] = s;
        }
    float* g = p.grad + size_t(i) * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float s = 0.f;
#pragma unroll
            for (int f = 0; f < 5; ++f) wmac(s, w4_G(f, b), t[a][f]);
            if (p.accumulate) g[a * 4 + b] += s; else g[a * 4 + b] = s;
        }
}

}  // namespace

#include "wino_tile.h"

namespace {

__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const ng::WinoG p) {
    __shared__ __attribute__((aligned(16))) char lds[ng::WINO_LDS_BYTES];
    ng::wino_tile(p, blockIdx.x, lds);
}

// y = sum over the frequency splits of the partial outputs (+ bias): float4 per thread, splits in order
__global__ __launch_bounds__(256) void wino_split_reduce_kernel(const ng::WinoG p) {
    const long long n4 = p.ws_stride / 4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 s = *reinterpret_cast<const f32x4*>(p.ws + i * 4);
        for (int k = 1; k < p.fsplit; ++k) s += *reinterpret_cast<const f32x4*>(p.ws + k * p.ws_stride + i * 4);
        if (p.bias != nullptr) s += *reinterpret_cast<const f32x4*>(p.bias + (i * 4) % p.K);
        *reinterpret_cast<f32x4*>(p.y + i * 4) = s;
    }
}

}  // namespace

static inline int wino_r(int r) { return r == 0 ? 3 : r; }

extern "C" int64_t nirgan_wino_ws_elems_r(int B, int H, int W, int C, int K, int r) {
    r = wino_r(r);
    if (B <= 0 || H <= 1 || W <= 1 || C <= 0 || K <= 0 || (r != 3 && r != 4)) return 0;
    const long long nf = (r + 1) * (r + 1);
    return nf * B * ((H + 1) / 2) * ((W + 1) / 2) * C + nf * K * C;
}

extern "C" int64_t nirgan_wino_ws_elems(int B, int H, int W, int C, int K) { return nirgan_wino_ws_elems_r(B, H, W, C, K, 3); }

extern "C" int nirgan_wino_weights_r(const float* w, int K, int C, int r, int transpose_flip, float* U, void* stream) {
    r = wino_r(r);
    NG_REQUIRE(w && U && K > 0 && C > 0, "wino_weights: bad arguments");
    NG_REQUIRE(r == 3 || r == 4, "wino_weights: filter size %d (3 or 4)", r);
    WinoW p{w, U, K, C, transpose_flip ? 1 : 0};
    const long long n = (long long)K * C;
    if (r == 3) hipLaunchKernelGGL(wino_weight_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino4_weight_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino_weights");
}

extern "C" int nirgan_wino_weights(const float* w, int K, int C, int transpose_flip, float* U, void* stream) {
    return nirgan_wino_weights_r(w, K, C, 3, transpose_flip, U, stream);
}

extern "C" int nirgan_wino_weights_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream) {
    NG_REQUIRE(jobs_device && njobs >= 1 && njobs <= 256 && total_blocks >= 1, "wino_weights_batch: bad arguments");
    hipLaunchKernelGGL(wino_weights_batch_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(jobs_device), njobs);
    return nirgan_check_launch("wino_weights_batch");
}

static int wino_check(const nirgan_wino_desc* d) {
    NG_REQUIRE(d && d->x && d->U && d->V && d->y && d->zero_page, "wino_conv: null pointer");
    const int r = wino_r(d->r);
    NG_REQUIRE(r == 3 || r == 4, "wino_conv: filter size %d (3 or 4)", r);
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1, "wino_conv: empty problem (H=%d W=%d)", d->H, d->W);
    NG_REQUIRE(d->C % 32 == 0 && d->C > 0 && d->K > 0 && d->K % 128 == 0, "wino_conv: C %% 32 == 0 and K %% 128 == 0 (C=%d K=%d)", d->C, d->K);
    NG_REQUIRE(d->x_hp == d->H + r - 1 && d->x_wp == d->W + r - 1, "wino_conv: the input must be (H+%d) x (W+%d) for the %dx%d filter (%dx%d for %dx%d)",
               r - 1, r - 1, r, r, d->x_hp, d->x_wp, d->H, d->W);
    NG_REQUIRE(ng_aligned16(d->x) && ng_aligned16(d->U) && ng_aligned16(d->V) && ng_aligned16(d->y) && ng_aligned16(d->zero_page), "wino_conv: pointers must be 16-byte aligned");
    const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2), nf = (r + 1) * (r + 1);
    NG_REQUIRE(T < (1ll << 31) / d->C, "wino_conv: problem too large for 32-bit tile offsets");
    NG_REQUIRE(d->V_elems >= nf * T * d->C, "wino_conv: V workspace too small");
    if (d->fsplit > 1) {
        NG_REQUIRE(d->fsplit <= nf, "wino_conv: fsplit=%d exceeds the %lld frequencies", d->fsplit, nf);
        NG_REQUIRE(d->split_ws && ng_aligned16(d->split_ws) && d->split_ws_elems >= (long long)d->fsplit * d->B * d->H * d->W * d->K,
                   "wino_conv: split workspace missing or too small");
    }
    return NIRGAN_OK;
}

// validation + parameters of the GEMM stage (shared with nirgan_wino_wgrad_pair, igemm_wgrad.hip)
int ng_wino_gemm_params(const nirgan_wino_desc* d, ng::WinoG* g) {
    const int rc = wino_check(d);
    if (rc != NIRGAN_OK) return rc;
    ng::build_wino_params(d, *g);
    return NIRGAN_OK;
}

static int wino_input_impl(const nirgan_wino_desc* d, const nirgan_wino_dy_desc* y, void* stream) {
    // the input transform needs x, V and the geometry only (the weight-gradient path transforms the forward input without a GEMM)
    NG_REQUIRE(d && d->x && d->V, "wino_input: null pointer");
    const int r = wino_r(d->r);
    NG_REQUIRE(r == 3 || r == 4, "wino_input: filter size %d (3 or 4)", r);
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->C > 0 && d->C % 4 == 0, "wino_input: bad shape");
    NG_REQUIRE(d->x_hp == d->H + r - 1 && d->x_wp == d->W + r - 1, "wino_input: the input must be (H+%d) x (W+%d)", r - 1, r - 1);
    NG_REQUIRE(ng_aligned16(d->x) && ng_aligned16(d->V), "wino_input: pointers must be 16-byte aligned");
    const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2), nf = (r + 1) * (r + 1);
    NG_REQUIRE(d->V_elems >= nf * T * d->C, "wino_input: V workspace too small");
    WinoIn in;
    in.x = d->x; in.V = d->V; in.B = d->B; in.H = d->H; in.W = d->W; in.C = d->C;
    in.x_row = d->x_wp * d->C; in.x_img = d->x_hp * in.x_row; in.TH = (d->H + 1) / 2; in.TW = (d->W + 1) / 2; in.T = T;
    in.Yt = nullptr; in.yTH = in.yTW = 0; in.yT = 0;
    if (y != nullptr) {
        // the same dY buffer seen twice: zero halo r-1, the data gradient covers (H + r - 1) x (W + r - 1) outputs
        NG_REQUIRE(y->dy == d->x && y->Yt && wino_r(y->r) == r && y->dy_pad == r - 1 && y->B == d->B && y->K == d->C
                   && y->dy_hp == d->x_hp && y->dy_wp == d->x_wp && d->H == y->H + r - 1 && d->W == y->W + r - 1,
                   "wino_input_dy: the two descriptors do not describe the same output-gradient buffer");
        NG_REQUIRE(ng_aligned16(y->Yt), "wino_input_dy: pointers must be 16-byte aligned");
        in.yTH = (y->H + 1) / 2; in.yTW = (y->W + 1) / 2; in.yT = (long long)y->B * in.yTH * in.yTW;
        NG_REQUIRE(y->Yt_elems >= nf * in.yT * y->K, "wino_input_dy: Yt workspace too small");
        in.Yt = y->Yt;
    }
    const long long nthreads = T * (d->C / 4);
    const dim3 grid(unsigned((nthreads + 255) / 256));
    if (r == 3) hipLaunchKernelGGL(wino_input_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in);
    else hipLaunchKernelGGL(wino4_input_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in);
    return nirgan_check_launch("wino_input");
}

extern "C" int nirgan_wino_input(const nirgan_wino_desc* d, void* stream) { return wino_input_impl(d, nullptr, stream); }

extern "C" int nirgan_wino_input_norm(const nirgan_wino_desc* d, const float* y, const float* mean, const float* rstd, int act, float slope, void* stream) {
    NG_REQUIRE(d && d->V && y && mean && rstd, "wino_input_norm: null pointer");
    NG_REQUIRE(wino_r(d->r) == 3, "wino_input_norm: 3x3 filters only");
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->C > 0 && d->C % 4 == 0, "wino_input_norm: bad shape");
    NG_REQUIRE(act == NIRGAN_ACT_NONE || act == NIRGAN_ACT_RELU || act == NIRGAN_ACT_LRELU, "wino_input_norm: activation %d", act);
    NG_REQUIRE(ng_aligned16(y) && ng_aligned16(d->V) && ng_aligned16(mean) && ng_aligned16(rstd), "wino_input_norm: pointers must be 16-byte aligned");
    const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    NG_REQUIRE(d->V_elems >= 16 * T * d->C, "wino_input_norm: V workspace too small");
    WinoInN in{y, mean, rstd, d->V, d->B, d->H, d->W, d->C, (d->H + 1) / 2, (d->W + 1) / 2, T, act, slope};
    const long long nthreads = T * (d->C / 4);
    hipLaunchKernelGGL(wino_input_norm_kernel, dim3(unsigned((nthreads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in);
    return nirgan_check_launch("wino_input_norm");
}

extern "C" int nirgan_wino_input_dy(const nirgan_wino_desc* d, const nirgan_wino_dy_desc* y, void* stream) {
    NG_REQUIRE(y != nullptr, "wino_input_dy: null pointer");
    return wino_input_impl(d, y, stream);
}

extern "C" int nirgan_wino_gemm(const nirgan_wino_desc* d, void* stream) {
    ng::WinoG g;
    const int rc = ng_wino_gemm_params(d, &g);
    if (rc != NIRGAN_OK) return rc;
    hipLaunchKernelGGL(wino_gemm_kernel, dim3(g.mtiles * g.ntiles * g.fsplit), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    if (g.fsplit > 1) {
        const long long n4 = g.ws_stride / 4;
        const long long blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(wino_split_reduce_kernel, dim3(unsigned(blocks < 8192 ? blocks : 8192)), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    }
    return nirgan_check_launch("wino_gemm");
}

extern "C" int nirgan_wino_conv3x3(const nirgan_wino_desc* d, void* stream) {
    const int rc = nirgan_wino_input(d, stream);
    return rc != NIRGAN_OK ? rc : nirgan_wino_gemm(d, stream);
}

extern "C" int nirgan_wino_dy(const nirgan_wino_dy_desc* d, void* stream) {
    NG_REQUIRE(d && d->dy && d->Yt, "wino_dy: null pointer");
    const int r = wino_r(d->r);
    NG_REQUIRE(r == 3 || r == 4, "wino_dy: filter size %d (3 or 4)", r);
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1 && d->K > 0 && d->K % 4 == 0 && d->dy_pad >= 0, "wino_dy: bad shape");
    NG_REQUIRE(d->dy_hp == d->H + 2 * d->dy_pad && d->dy_wp == d->W + 2 * d->dy_pad, "wino_dy: dy geometry mismatch");
    NG_REQUIRE(ng_aligned16(d->dy) && ng_aligned16(d->Yt), "wino_dy: pointers must be 16-byte aligned");
    WinoDy p;
    p.dy = d->dy; p.Yt = d->Yt; p.B = d->B; p.H = d->H; p.W = d->W; p.K = d->K;
    p.d_row = d->dy_wp * d->K; p.d_img = d->dy_hp * p.d_row; p.d_org = d->dy_pad * p.d_row + d->dy_pad * d->K;
    p.TH = (d->H + 1) / 2; p.TW = (d->W + 1) / 2; p.T = (long long)d->B * p.TH * p.TW;
    NG_REQUIRE(d->Yt_elems >= (long long)(r + 1) * (r + 1) * p.T * d->K, "wino_dy: workspace too small");
    const long long n = p.T * (d->K / 4);
    const dim3 grid(unsigned((n + 255) / 256));
    if (r == 3) hipLaunchKernelGGL(wino_dy_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino4_dy_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino_dy");
}

extern "C" int nirgan_wino_wgrad_finish_r(const float* slabs, int nsplit, int K, int C, int r, float* grad, int accumulate, void* stream) {
    r = wino_r(r);
    NG_REQUIRE(slabs && grad && nsplit >= 1 && K > 0 && C > 0, "wino_wgrad_finish: bad arguments");
    NG_REQUIRE(r == 3 || r == 4, "wino_wgrad_finish: filter size %d (3 or 4)", r);
    WinoFin p{slabs, nsplit, K, C, grad, accumulate ? 1 : 0};
    const long long n = (long long)K * C;
    const dim3 grid(unsigned((n + 63) / 64));
    if (r == 3) hipLaunchKernelGGL(wino_wgrad_finish_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wino4_wgrad_finish_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino_wgrad_finish");
}

extern "C" int nirgan_wino_wgrad_finish(const float* slabs, int nsplit, int K, int C, float* grad, int accumulate, void* stream) {
    return nirgan_wino_wgrad_finish_r(slabs, nsplit, K, C, 3, grad, accumulate, stream);
}
