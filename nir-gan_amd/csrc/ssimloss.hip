// SSIM loss of the generator objective, value and gradient: model/pix2pix.py:233-237 adds
// lambda_ssim * ssim_loss(pred, nir) with utils/losses.py:10-30 = 1 - kornia.metrics.ssim(img1, img2, 11).mean()
// (Gaussian window sigma 1.5, reflect border, "same" size, C1 = (0.01 L)^2, C2 = (0.03 L)^2, eps in the denominator).
//
// With mu = G*x, nu = G*y, m_xx = G*x^2, m_yy = G*y^2, m_xy = G*xy (G = reflect-padded Gaussian filter):
//   A1 = 2 mu nu + C1, A2 = 2 (m_xy - mu nu) + C2, B1 = mu^2 + nu^2 + C1, B2 = (m_xx - mu^2) + (m_yy - nu^2) + C2
//   S = A1 A2 / D, D = B1 B2 + eps
//   a = dS/dmu   = (2 nu (A2 - A1) - 2 mu S (B2 - B1)) / D
//   b = dS/dm_xx = -S B1 / D
//   c = dS/dm_xy = 2 A1 / D
//   d(sum S)/dx(q) = G^T[a](q) + 2 x(q) G^T[b](q) + y(q) G^T[c](q)
// and G^T (adjoint of filter-after-reflect-pad) = the full zero-padded correlation on the padded domain, folded back
// through the reflect images of q.  Three HBM-bound launches over single-channel planes (16 x 256^2: well under a
// microsecond of traffic each; the cost is launch latency):
//   ssim_maps_kernel     S partial sums + the maps a, b, c                      (same tiling as metrics.hip)
//   ssim_adjoint_kernel  F_m = w (*) zero-padded m on the (H+2r) x (W+2r) domain, m in {a, b, c}
//   ssim_fold_kernel     grad(q) += -weight/N * (fold F_a + 2 x fold F_b + y fold F_c);  loss += weight (1 - mean S)
#include "common.h"

namespace {

constexpr int TILE = 32;
constexpr int MAXR = 5;                       // window <= 11
constexpr int PW = TILE + 2 * MAXR;

struct SsimP {
    const float* x; const float* y;
    int planes, H, W, r;
    float k[2 * MAXR + 1];
    float c1, c2, eps;
    float* maps;                              // [3][planes][H][W]
    float* F;                                 // [3][planes][H+2r][W+2r]
    float* partials;                          // [blocks of the maps launch]
    int tiles_x, tiles_y, ptiles_x, ptiles_y;
    float weight; float* loss; float* value; float* grad;
    int nblocks;
};

__global__ __launch_bounds__(256) void ssim_maps_kernel(const SsimP p) {
    __shared__ float sa[PW][PW + 1], sb[PW][PW + 1];
    __shared__ float hm[5][PW][TILE + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int plane = bid / p.tiles_y;
    const int r = p.r, pw = TILE + 2 * r;
    const size_t hw = size_t(p.H) * p.W;
    const float* A = p.x + size_t(plane) * hw;
    const float* B = p.y + size_t(plane) * hw;
    const int h0 = ty * TILE - r, w0 = tx * TILE - r;
    for (int i = tid; i < pw * pw; i += 256) {
        const int y = i / pw, x = i - y * pw;
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
    float s_ssim = 0.f;
    const size_t mplane = size_t(p.planes) * hw;
    for (int i = tid; i < TILE * TILE; i += 256) {
        const int y = i / TILE, x = i - y * TILE;
        const int oh = ty * TILE + y, ow = tx * TILE + x;
        if (oh >= p.H || ow >= p.W) continue;
        float mu = 0.f, nu = 0.f, mxx = 0.f, myy = 0.f, mxy = 0.f;
        for (int t = 0; t <= 2 * r; ++t) {
            const float wv = p.k[t];
            mu += wv * hm[0][y + t][x]; nu += wv * hm[1][y + t][x]; mxx += wv * hm[2][y + t][x];
            myy += wv * hm[3][y + t][x]; mxy += wv * hm[4][y + t][x];
        }
        const float A1 = 2.f * mu * nu + p.c1, A2 = 2.f * (mxy - mu * nu) + p.c2;
        const float B1 = mu * mu + nu * nu + p.c1, B2 = (mxx - mu * mu) + (myy - nu * nu) + p.c2;
        const float inv = 1.f / (B1 * B2 + p.eps);
        const float S = A1 * A2 * inv;
        s_ssim += S;
        const size_t o = size_t(plane) * hw + size_t(oh) * p.W + ow;
        p.maps[o] = (2.f * nu * (A2 - A1) - 2.f * mu * S * (B2 - B1)) * inv;
        p.maps[mplane + o] = -S * B1 * inv;
        p.maps[2 * mplane + o] = 2.f * A1 * inv;
    }
    s_ssim = ng_wave_sum(s_ssim);
    if ((tid & 63) == 0) red[tid >> 6] = s_ssim;
    __syncthreads();
    if (tid == 0) p.partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grid: ptiles_x * ptiles_y * planes * 3 blocks; F_m(i) = sum_d w[d] m(i - d) on the padded domain, m zero outside the image
__global__ __launch_bounds__(256) void ssim_adjoint_kernel(const SsimP p) {
    __shared__ float sm[PW][PW + 1];
    __shared__ float hm[PW][TILE + 1];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % p.ptiles_x; bid /= p.ptiles_x;
    const int ty = bid % p.ptiles_y; bid /= p.ptiles_y;
    const int plane = bid % p.planes, map = bid / p.planes;
    const int r = p.r, pw = TILE + 2 * r;
    const int Hp = p.H + 2 * r, Wp = p.W + 2 * r;
    const size_t hw = size_t(p.H) * p.W;
    const float* M = p.maps + (size_t(map) * p.planes + plane) * hw;
    // padded output (iy, ix) in [0, Hp) x [0, Wp) reads image rows iy - 2r .. iy
    const int h0 = ty * TILE - 2 * r, w0 = tx * TILE - 2 * r;
    for (int i = tid; i < pw * pw; i += 256) {
        const int y = i / pw, x = i - y * pw;
        const int hh = h0 + y, ww = w0 + x;
        sm[y][x] = (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W) ? M[size_t(hh) * p.W + ww] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < pw * TILE; i += 256) {
        const int y = i / TILE, x = i - y * TILE;
        float s = 0.f;
        for (int t = 0; t <= 2 * r; ++t) s += p.k[t] * sm[y][x + t];
        hm[y][x] = s;
    }
    __syncthreads();
    float* F = p.F + (size_t(map) * p.planes + plane) * Hp * Wp;
    for (int i = tid; i < TILE * TILE; i += 256) {
        const int y = i / TILE, x = i - y * TILE;
        const int oy = ty * TILE + y, ox = tx * TILE + x;
        if (oy >= Hp || ox >= Wp) continue;
        float s = 0.f;
        for (int t = 0; t <= 2 * r; ++t) s += p.k[t] * hm[y + t][x];
        F[size_t(oy) * Wp + ox] = s;
    }
}

// padded coordinates holding a reflect image of interior coordinate h (halo r < H): the pixel itself and its mirrors
__device__ __forceinline__ int reflect_images(int h, int H, int r, int* out) {
    int n = 0;
    out[n++] = h + r;
    if (h >= 1 && h <= r) out[n++] = r - h;
    if (h >= H - 1 - r && h <= H - 2) out[n++] = r + 2 * (H - 1) - h;
    return n;
}

__global__ __launch_bounds__(256) void ssim_fold_kernel(const SsimP p) {
    const int r = p.r, Hp = p.H + 2 * r, Wp = p.W + 2 * r;
    const size_t hw = size_t(p.H) * p.W, total = size_t(p.planes) * hw;
    const float scale = -p.weight / float(total);
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += size_t(gridDim.x) * 256) {
        const int plane = int(i / hw);
        const int rem = int(i - size_t(plane) * hw);
        const int h = rem / p.W, w = rem - h * p.W;
        int hs[3], ws[3];
        const int nh = reflect_images(h, p.H, r, hs), nw = reflect_images(w, p.W, r, ws);
        float f[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float* F = p.F + (size_t(m) * p.planes + plane) * Hp * Wp;
            float s = 0.f;
            for (int a = 0; a < nh; ++a)
                for (int b = 0; b < nw; ++b) s += F[size_t(hs[a]) * Wp + ws[b]];
            f[m] = s;
        }
        p.grad[i] += scale * (f[0] + 2.f * p.x[i] * f[1] + p.y[i] * f[2]);
    }
}

// one block: 1 - (sum of the partials in a fixed order) / N
__global__ __launch_bounds__(256) void ssim_value_kernel(const SsimP p) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < p.nblocks; i += 256) s += p.partials[i];
    s = ng_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = 1.f - ((red[0] + red[1]) + (red[2] + red[3])) / (float(p.planes) * float(p.H) * float(p.W));
        if (p.value) *p.value = v;
        if (p.loss) atomicAdd(p.loss, p.weight * v);      // micro-batches on separate streams add their shares (two terms: order-free)
    }
}

int64_t tiles_of(int n) { return (n + TILE - 1) / TILE; }

}  // namespace

extern "C" int64_t nirgan_ssim_loss_ws_elems(int planes, int H, int W, int window) {
    if (planes <= 0 || H <= 0 || W <= 0 || window < 1 || window > 2 * MAXR + 1 || !(window & 1)) return 0;
    const int r = window / 2;
    return 3ll * planes * H * W + 3ll * planes * (H + 2 * r) * (W + 2 * r) + int64_t(planes) * tiles_of(H) * tiles_of(W);
}

extern "C" int nirgan_ssim_loss(const nirgan_ssim_loss_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr && d->pred && d->target && d->ws, "ssim_loss: null pointer");
    NG_REQUIRE(d->planes > 0 && d->H > 0 && d->W > 0, "ssim_loss: empty problem");
    NG_REQUIRE(d->window >= 1 && d->window <= 2 * MAXR + 1 && (d->window & 1), "ssim_loss: window=%d must be odd and <= %d", d->window, 2 * MAXR + 1);
    const int r = d->window / 2;
    NG_REQUIRE(d->H > r && d->W > r, "ssim_loss: image smaller than the window radius (reflect border)");
    NG_REQUIRE(d->sigma > 0.f && d->max_val > 0.f, "ssim_loss: sigma and max_val must be positive");
    NG_REQUIRE(d->ws_elems >= nirgan_ssim_loss_ws_elems(d->planes, d->H, d->W, d->window), "ssim_loss: workspace too small (nirgan_ssim_loss_ws_elems)");
    SsimP p;
    p.x = d->pred; p.y = d->target; p.planes = d->planes; p.H = d->H; p.W = d->W; p.r = r;
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
    p.tiles_x = int(tiles_of(d->W)); p.tiles_y = int(tiles_of(d->H));
    p.ptiles_x = int(tiles_of(d->W + 2 * r)); p.ptiles_y = int(tiles_of(d->H + 2 * r));
    const int64_t hw = int64_t(d->H) * d->W, hwp = int64_t(d->H + 2 * r) * (d->W + 2 * r);
    p.maps = d->ws;
    p.F = d->ws + 3 * d->planes * hw;
    p.partials = p.F + 3 * d->planes * hwp;
    p.weight = d->weight; p.loss = d->loss; p.value = d->value; p.grad = d->grad_pred;
    const int64_t blocks = int64_t(d->planes) * p.tiles_x * p.tiles_y;
    const int64_t pblocks = 3ll * d->planes * p.ptiles_x * p.ptiles_y;
    NG_REQUIRE(blocks < (1ll << 31) && pblocks < (1ll << 31), "ssim_loss: too many tiles");
    p.nblocks = int(blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ssim_maps_kernel, dim3(unsigned(blocks)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(ssim_value_kernel, dim3(1), dim3(256), 0, st, p);
    if (d->grad_pred) {
        hipLaunchKernelGGL(ssim_adjoint_kernel, dim3(unsigned(pblocks)), dim3(256), 0, st, p);
        const int64_t total = int64_t(d->planes) * hw;
        const int64_t fb = (total + 255) / 256;
        hipLaunchKernelGGL(ssim_fold_kernel, dim3(unsigned(fb < 4096 ? fb : 4096)), dim3(256), 0, st, p);
    }
    return nirgan_check_launch("ssim_loss");
}
