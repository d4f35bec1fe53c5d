// Image-quality metrics of the train / validation loop on the device, one pass over pred and target:
// sum |d|, sum d^2 and the sum of the SSIM map (Gaussian window, sigma 1.5, reflect border, "same" size).
// HBM-bound: each input element is read once per tile (+ the window halo), nothing is written but per-block
// partial sums; a second tiny launch adds the partials in block order (deterministic, no float atomics).
//
// Per 32x32 output tile: the (32+2r)^2 input patches of both images go to LDS (reflect indexing at the image
// border), a horizontal pass produces the five filtered moments x, y, x^2, y^2, xy for the (32+2r) x 32 strip,
// a vertical pass finishes them at the 4 output pixels of each thread.
#include "common.h"

namespace {

constexpr int TILE = 32;
constexpr int MAXR = 5;                       // window <= 11
constexpr int PW = TILE + 2 * MAXR;           // patch width

struct MetricsP {
    const float* a; const float* b;
    int planes, H, W, r;
    float k[2 * MAXR + 1];
    float c1, c2, eps;
    float* partials;                          // [blocks][3]
    int tiles_x, tiles_y;
};

__global__ __launch_bounds__(256) void metrics_kernel(const MetricsP p) {
    __shared__ float sa[PW][PW + 1], sb[PW][PW + 1];
    __shared__ float hm[5][PW][TILE + 1];
    __shared__ float red[4][3];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int plane = bid / p.tiles_y;
    const int r = p.r, pw = TILE + 2 * r;
    const float* A = p.a + size_t(plane) * p.H * p.W;
    const float* B = p.b + size_t(plane) * p.H * p.W;
    const int h0 = ty * TILE - r, w0 = tx * TILE - r;
    for (int i = tid; i < pw * pw; i += 256) {
        const int y = i / pw, x = i - y * pw;
        // reflect without edge repeat (image larger than the window radius: checked on the host).  Patch positions past
        // the image's last partial tile feed no output: clamp them into the reflectable band first.
        const int ph = h0 + y < p.H + r ? h0 + y : p.H - 1 + r, pwc = w0 + x < p.W + r ? w0 + x : p.W - 1 + r;
        const int hh = ng_reflect(ph, p.H), ww = ng_reflect(pwc, p.W);
        sa[y][x] = A[size_t(hh) * p.W + ww];
        sb[y][x] = B[size_t(hh) * p.W + ww];
    }
    __syncthreads();
    for (int i = tid; i < pw * TILE; i += 256) {
        const int y = i / TILE, x = i - y * TILE;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        for (int t = 0; t <= 2 * r; ++t) {
            const float wv = p.k[t], u = sa[y][x + t], v = sb[y][x + t];
            m0 += wv * u; m1 += wv * v; m2 += wv * u * u; m3 += wv * v * v; m4 += wv * u * v;
        }
        hm[0][y][x] = m0; hm[1][y][x] = m1; hm[2][y][x] = m2; hm[3][y][x] = m3; hm[4][y][x] = m4;
    }
    __syncthreads();
    float s_l1 = 0.f, s_l2 = 0.f, s_ssim = 0.f;
    for (int i = tid; i < TILE * TILE; i += 256) {
        const int y = i / TILE, x = i - y * TILE;
        const int oh = ty * TILE + y, ow = tx * TILE + x;
        if (oh >= p.H || ow >= p.W) continue;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        for (int t = 0; t <= 2 * r; ++t) {
            const float wv = p.k[t];
            m0 += wv * hm[0][y + t][x]; m1 += wv * hm[1][y + t][x]; m2 += wv * hm[2][y + t][x];
            m3 += wv * hm[3][y + t][x]; m4 += wv * hm[4][y + t][x];
        }
        const float mu1_sq = m0 * m0, mu2_sq = m1 * m1, mu12 = m0 * m1;
        const float s1 = m2 - mu1_sq, s2 = m3 - mu2_sq, s12 = m4 - mu12;
        const float num = (2.f * mu12 + p.c1) * (2.f * s12 + p.c2);
        const float den = (mu1_sq + mu2_sq + p.c1) * (s1 + s2 + p.c2);
        s_ssim += num / (den + p.eps);
        const float d = sa[y + r][x + r] - sb[y + r][x + r];
        s_l1 += fabsf(d);
        s_l2 += d * d;
    }
    s_l1 = ng_wave_sum(s_l1); s_l2 = ng_wave_sum(s_l2); s_ssim = ng_wave_sum(s_ssim);
    if ((tid & 63) == 0) { red[tid >> 6][0] = s_l1; red[tid >> 6][1] = s_l2; red[tid >> 6][2] = s_ssim; }
    __syncthreads();
    if (tid < 3) p.partials[size_t(blockIdx.x) * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// one block: sums[j] = sum over blocks (fixed order per lane, then a fixed tree) * scale
__global__ __launch_bounds__(256) void metrics_finalize_kernel(const float* __restrict__ partials, int nblocks, float scale, float* __restrict__ sums) {
    __shared__ float red[4][3];
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        s[0] += partials[size_t(i) * 3]; s[1] += partials[size_t(i) * 3 + 1]; s[2] += partials[size_t(i) * 3 + 2];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] = ng_wave_sum(s[j]);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s[0]; red[threadIdx.x >> 6][1] = s[1]; red[threadIdx.x >> 6][2] = s[2]; }
    __syncthreads();
    if (threadIdx.x < 3) sums[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * scale;
}

}  // namespace

extern "C" int64_t nirgan_image_metrics_ws_elems(int planes, int H, int W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    return int64_t(planes) * ((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE) * 3;
}

extern "C" int nirgan_image_metrics(const nirgan_metrics_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr && d->pred && d->target && d->ws && d->means, "image_metrics: null pointer");
    NG_REQUIRE(d->planes > 0 && d->H > 0 && d->W > 0, "image_metrics: empty problem");
    NG_REQUIRE(d->window >= 1 && d->window <= 2 * MAXR + 1 && (d->window & 1), "image_metrics: window=%d must be odd and <= %d", d->window, 2 * MAXR + 1);
    const int r = d->window / 2;
    NG_REQUIRE(d->H > r && d->W > r, "image_metrics: image smaller than the window radius (reflect border)");
    NG_REQUIRE(d->sigma > 0.f && d->max_val > 0.f, "image_metrics: sigma and max_val must be positive");
    MetricsP p;
    p.a = d->pred; p.b = d->target; p.planes = d->planes; p.H = d->H; p.W = d->W; p.r = r;
    double sum = 0.0, kv[2 * MAXR + 1];
    for (int t = 0; t < d->window; ++t) {
        const double x = double(t - r);
        kv[t] = exp(-(x * x) / (2.0 * double(d->sigma) * double(d->sigma)));
        sum += kv[t];
    }
    for (int t = 0; t < 2 * MAXR + 1; ++t) p.k[t] = t < d->window ? float(kv[t] / sum) : 0.f;
    p.c1 = (0.01f * d->max_val) * (0.01f * d->max_val);
    p.c2 = (0.03f * d->max_val) * (0.03f * d->max_val);
    p.eps = d->eps;
    p.tiles_x = (d->W + TILE - 1) / TILE; p.tiles_y = (d->H + TILE - 1) / TILE;
    const int64_t blocks = int64_t(d->planes) * p.tiles_x * p.tiles_y;
    NG_REQUIRE(blocks < (int64_t(1) << 31), "image_metrics: too many tiles");
    NG_REQUIRE(d->ws_elems >= blocks * 3, "image_metrics: workspace too small (nirgan_image_metrics_ws_elems)");
    p.partials = d->ws;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(metrics_kernel, dim3(unsigned(blocks)), dim3(256), 0, st, p);
    const float scale = 1.f / (float(d->planes) * float(d->H) * float(d->W));
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(256), 0, st, d->ws, int(blocks), scale, d->means);
    return nirgan_check_launch("image_metrics");
}
