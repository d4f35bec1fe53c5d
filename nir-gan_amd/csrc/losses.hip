// LSGAN, L1 and spectral-index (NDVI/NDWI/GNDVI/SAVI/MSAVI/EVI) losses: value and gradient
// wrt the prediction in one streaming pass over NCHW tiles.  HBM-bound.  No float atomics between blocks: LSGAN (a few 10^4 patch
// values) is ONE block; the pixel losses leave per-block partial sums in a workspace that ng_partials_finish adds up in block order,
// so the loss scalars are bitwise reproducible.
#include "common.h"

namespace {

template <int NV>
__device__ __forceinline__ void block_partials(float (&v)[NV], float* dst) {      // dst[NV]: this block's sums, fixed association
    __shared__ float part[16][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = ng_wave_sum(v[i]);
        if (lane == 0) part[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += part[w][threadIdx.x];
        dst[threadIdx.x] = t;
    }
}

// one block of 1024 threads, four elements in flight per thread and trip
__global__ __launch_bounds__(1024) void lsgan_kernel(const float* __restrict__ pred, int64_t n, float target, float weight,
                                                     float* loss_out, float* __restrict__ grad) {
    const float inv = 1.f / float(n);
    float acc[1] = {0.f};
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += 4096) {
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + j * 1024;
            d[j] = i < n ? pred[i] - target : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + j * 1024;
            acc[0] += d[j] * d[j];
            if (grad && i < n) grad[i] = weight * 2.f * d[j] * inv;
        }
    }
    acc[0] *= weight * inv;
    __shared__ float total[1];
    block_partials<1>(acc, total);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_out, total[0]);          // ONE add per launch: commutes with another stream's
}

struct PixP {
    const float* rgb; const float* nir; const float* pred;
    int B, HW;
    float w[7];   // l1, ndvi, ndwi, gndvi, savi, msavi, evi
    int criterion, log_all;
    const float* extra; int extra_cs, extra_c; float extra_scale;
    float* sums; float* grad; float* ws;
};

// value and derivative of criterion(a, f) wrt f, where a = idx(nir), f = idx(pred)
__device__ __forceinline__ void crit(int criterion, float a, float f, float& val, float& dval) {
    const float d = f - a;
    if (criterion == 0) {
        val = fabsf(d);
        dval = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    } else {
        val = d * d;
        dval = 2.f * d;
    }
}

__global__ __launch_bounds__(256) void pix_loss_kernel(const PixP p) {
    const int64_t n = int64_t(p.B) * p.HW;
    const float inv = 1.f / float(n);
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const int64_t b = i / p.HW, px = i - b * p.HW;
        // rgb == nullptr: plain L1 (torch.nn.L1Loss on two single-band tensors); no index term is evaluated then
        const float R = p.rgb ? p.rgb[(b * 3 + 0) * p.HW + px] : 0.f;
        const float G = p.rgb ? p.rgb[(b * 3 + 1) * p.HW + px] : 0.f;
        const float Bl = p.rgb ? p.rgb[(b * 3 + 2) * p.HW + px] : 0.f;
        const float x = p.nir[i], y = p.pred[i];
        float g = 0.f, v, dv;
        {   // L1 (torch.nn.L1Loss)
            const float d = y - x;
            acc[0] += fabsf(d);
            g += p.w[0] * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (p.log_all || p.w[1] != 0.f) {   // NDVI (n - R) / (n + R + 1e-6)
            const float dn = x + R + 1e-6f, dp = y + R + 1e-6f;
            crit(p.criterion, (x - R) / dn, (y - R) / dp, v, dv);
            acc[1] += v;
            if (p.w[1] != 0.f) g += p.w[1] * dv * ((dp - (y - R)) / (dp * dp));
        }
        if (p.log_all || p.w[2] != 0.f) {   // NDWI (n - G) / (n + G + 1e-6)
            const float dn = x + G + 1e-6f, dp = y + G + 1e-6f;
            crit(p.criterion, (x - G) / dn, (y - G) / dp, v, dv);
            acc[2] += v;
            if (p.w[2] != 0.f) g += p.w[2] * dv * ((dp - (y - G)) / (dp * dp));
        }
        if (p.log_all || p.w[3] != 0.f) {   // GNDVI (n - G) / (ndvi0(n) + G), ndvi0 without epsilon
            const float nd = (x - R) / (x + R), ndp = (y - R) / (y + R);
            const float den = nd + G, denp = ndp + G;
            crit(p.criterion, (x - G) / den, (y - G) / denp, v, dv);
            acc[3] += v;
            const float dndp = 2.f * R / ((y + R) * (y + R));
            if (p.w[3] != 0.f) g += p.w[3] * dv * ((denp - (y - G) * dndp) / (denp * denp));
        }
        if (p.log_all || p.w[4] != 0.f) {   // SAVI 1.5 (n - R) / (n + R + 0.5)
            const float dn = x + R + 0.5f, dp = y + R + 0.5f;
            crit(p.criterion, 1.5f * (x - R) / dn, 1.5f * (y - R) / dp, v, dv);
            acc[4] += v;
            if (p.w[4] != 0.f) g += p.w[4] * dv * (1.5f * (dp - (y - R)) / (dp * dp));
        }
        if (p.log_all || p.w[5] != 0.f) {   // MSAVI (2n + 1 - sqrt((2n+1)^2 - 8 (n - R))) / 2
            const float tn = 2.f * x + 1.f, tp = 2.f * y + 1.f;
            const float sn = sqrtf(tn * tn - 8.f * (x - R)), sp = sqrtf(tp * tp - 8.f * (y - R));
            crit(p.criterion, (tn - sn) * 0.5f, (tp - sp) * 0.5f, v, dv);
            acc[5] += v;
            if (p.w[5] != 0.f) g += p.w[5] * dv * (0.5f * (2.f - (4.f * tp - 8.f) / (2.f * sp)));
        }
        if (p.log_all || p.w[6] != 0.f) {   // EVI 2.5 (n - R) / ((n + 6)(R - 7.5)(B + 1) + 1e-6)
            const float c = (R - 7.5f) * (Bl + 1.f);
            const float dn = (x + 6.f) * c + 1e-6f, dp = (y + 6.f) * c + 1e-6f;
            crit(p.criterion, 2.5f * ((x - R) / dn), 2.5f * ((y - R) / dp), v, dv);
            acc[6] += v;
            if (p.w[6] != 0.f) g += p.w[6] * dv * (2.5f * (dp - (y - R) * c) / (dp * dp));
        }
        if (p.grad) {
            g *= inv;
            if (p.extra) g += p.extra_scale * p.extra[i * p.extra_cs + p.extra_c];
            p.grad[i] = g;
        }
    }
    block_partials<7>(acc, p.ws + size_t(blockIdx.x) * 7);
}

}  // namespace

extern "C" int nirgan_lsgan(const float* pred, int64_t n, float target, float weight, float* loss_out, float* grad, void* stream) {
    NG_REQUIRE(pred && loss_out && n > 0, "lsgan: bad arguments");
    hipLaunchKernelGGL(lsgan_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), pred, n, target, weight, loss_out, grad);
    return nirgan_check_launch("lsgan");
}

extern "C" int nirgan_pix_loss(const nirgan_pix_loss_desc* d, void* stream) {
    NG_REQUIRE(d && d->nir && d->pred && d->sums, "pix_loss: null pointer");
    NG_REQUIRE(d->rgb || (!d->log_all && d->w_ndvi == 0.f && d->w_ndwi == 0.f && d->w_gndvi == 0.f && d->w_savi == 0.f &&
                          d->w_msavi == 0.f && d->w_evi == 0.f), "pix_loss: the spectral indices need rgb");
    NG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "pix_loss: bad shape");
    NG_REQUIRE(d->criterion == 0 || d->criterion == 1, "pix_loss: criterion must be 0 (l1) or 1 (l2)");
    NG_REQUIRE(!d->extra || (d->extra_c >= 0 && d->extra_c < d->extra_cs), "pix_loss: extra channel out of range");
    PixP p;
    p.rgb = d->rgb; p.nir = d->nir; p.pred = d->pred; p.B = d->B; p.HW = d->H * d->W;
    p.w[0] = d->w_l1; p.w[1] = d->w_ndvi; p.w[2] = d->w_ndwi; p.w[3] = d->w_gndvi; p.w[4] = d->w_savi; p.w[5] = d->w_msavi; p.w[6] = d->w_evi;
    p.criterion = d->criterion; p.log_all = d->log_all;
    p.extra = d->extra; p.extra_cs = d->extra_cs; p.extra_c = d->extra_c; p.extra_scale = d->extra_scale;
    p.sums = d->sums; p.grad = d->grad_pred;
    const int64_t n = int64_t(d->B) * p.HW;
    int64_t g = (n + 255) / 256;
    g = g < NIRGAN_PIX_LOSS_WS_ELEMS / 8 ? g : NIRGAN_PIX_LOSS_WS_ELEMS / 8;
    NG_REQUIRE(d->ws && d->ws_elems >= NIRGAN_PIX_LOSS_WS_ELEMS, "pix_loss: workspace of NIRGAN_PIX_LOSS_WS_ELEMS floats required (block partial sums)");
    p.ws = d->ws;
    hipLaunchKernelGGL(pix_loss_kernel, dim3(int(g)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return ng_partials_finish(d->ws, int(g), 7, d->sums, static_cast<hipStream_t>(stream));
}
