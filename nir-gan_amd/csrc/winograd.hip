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
__global__ __launch_bounds__(256) void wino_weight_kernel(const WinoW p) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
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
}

}  // namespace

#include "wino_tile.h"

namespace {

__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const ng::WinoG p) {
    __shared__ __attribute__((aligned(16))) char lds[ng::WINO_LDS_BYTES];
    ng::wino_tile(p, blockIdx.x, lds);
}

}  // namespace

extern "C" int64_t nirgan_wino_ws_elems(int B, int H, int W, int C, int K) {
    if (B <= 0 || H <= 1 || W <= 1 || C <= 0 || K <= 0) return 0;
    return 16ll * B * ((H + 1) / 2) * ((W + 1) / 2) * C + 16ll * K * C;
}

extern "C" int nirgan_wino_weights(const float* w, int K, int C, int transpose_flip, float* U, void* stream) {
    NG_REQUIRE(w && U && K > 0 && C > 0, "wino_weights: bad arguments");
    WinoW p{w, U, K, C, transpose_flip ? 1 : 0};
    const long long n = (long long)K * C;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("wino_weights");
}

static int wino_check(const nirgan_wino_desc* d) {
    NG_REQUIRE(d && d->x && d->U && d->V && d->y && d->zero_page, "wino_conv3x3: null pointer");
    NG_REQUIRE(d->B > 0 && d->H > 1 && d->W > 1, "wino_conv3x3: empty problem (H=%d W=%d)", d->H, d->W);
    NG_REQUIRE(d->C % 32 == 0 && d->C > 0 && d->K > 0 && d->K % 128 == 0, "wino_conv3x3: C %% 32 == 0 and K %% 128 == 0 (C=%d K=%d)", d->C, d->K);
    NG_REQUIRE(d->x_hp == d->H + 2 && d->x_wp == d->W + 2, "wino_conv3x3: the input must carry a halo of exactly 1 (%dx%d for %dx%d)", d->x_hp, d->x_wp, d->H, d->W);
    NG_REQUIRE(ng_aligned16(d->x) && ng_aligned16(d->U) && ng_aligned16(d->V) && ng_aligned16(d->y) && ng_aligned16(d->zero_page), "wino_conv3x3: pointers must be 16-byte aligned");
    const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    NG_REQUIRE(16 * T * d->C < (1ll << 31) * 4 && T < (1ll << 31) / d->C, "wino_conv3x3: problem too large for 32-bit tile offsets");
    NG_REQUIRE(d->V_elems >= 16 * T * d->C, "wino_conv3x3: V workspace too small");
    return NIRGAN_OK;
}

// validation + parameters of the GEMM stage (shared with nirgan_wino_wgrad_pair, igemm_wgrad.hip)
int ng_wino_gemm_params(const nirgan_wino_desc* d, ng::WinoG* g) {
    const int rc = wino_check(d);
    if (rc != NIRGAN_OK) return rc;
    ng::build_wino_params(d, *g);
    return NIRGAN_OK;
}

extern "C" int nirgan_wino_input(const nirgan_wino_desc* d, void* stream) {
    const int rc = wino_check(d);
    if (rc != NIRGAN_OK) return rc;
    const long long T = (long long)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    WinoIn in;
    in.x = d->x; in.V = d->V; in.B = d->B; in.H = d->H; in.W = d->W; in.C = d->C;
    in.x_row = d->x_wp * d->C; in.x_img = d->x_hp * in.x_row; in.TH = (d->H + 1) / 2; in.TW = (d->W + 1) / 2; in.T = T;
    const long long nthreads = T * (d->C / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3(unsigned((nthreads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in);
    return nirgan_check_launch("wino_input");
}

extern "C" int nirgan_wino_gemm(const nirgan_wino_desc* d, void* stream) {
    ng::WinoG g;
    const int rc = ng_wino_gemm_params(d, &g);
    if (rc != NIRGAN_OK) return rc;
    hipLaunchKernelGGL(wino_gemm_kernel, dim3(g.mtiles * g.ntiles), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    return nirgan_check_launch("wino_gemm");
}

extern "C" int nirgan_wino_conv3x3(const nirgan_wino_desc* d, void* stream) {
    const int rc = nirgan_wino_input(d, stream);
    return rc != NIRGAN_OK ? rc : nirgan_wino_gemm(d, stream);
}
