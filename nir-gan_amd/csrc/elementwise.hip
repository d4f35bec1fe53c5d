// Flat fp32 streaming kernels: Adam, SatCLIP injection (bilinear resize + modulation),
// column sums, fill, axpy, and the descriptor-list runner.  All HBM-bound.
#include "common.h"

namespace {

inline int grid_for(int64_t total, int cap = 4096) {
    const int64_t g = (total + 255) / 256;
    return int(g < cap ? (g < 1 ? 1 : g) : cap);
}

// torch.optim.Adam (single tensor form): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float b1, float b2, float eps,
                                                   float step_size, float inv_sqrt_bc2) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n4; i += int64_t(gridDim.x) * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = mm[k] * b1 + (1.f - b1) * gg[k];
            vv[k] = vv[k] * b2 + (1.f - b2) * gg[k] * gg[k];
            const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
            pp[k] -= step_size * (mm[k] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gg = g[i];
        const float mm = m[i] * b1 + (1.f - b1) * gg;
        const float vv = v[i] * b2 + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        p[i] -= step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    }
}

// align_corners=False source coordinate (ATen area_pixel_compute_source_index, cubic=false)
__device__ __forceinline__ void bl_coord(int o, float scale, int S, int& i0, int& i1, float& l1) {
    float s = scale * (float(o) + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = int(s);
    i0 = i0 < S - 1 ? i0 : S - 1;
    i1 = i0 < S - 1 ? i0 + 1 : i0;
    l1 = s - float(i0);
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ src, int B, int SH, int SW, float* __restrict__ dst, int OH, int OW) {
    const float sh = float(SH) / float(OH), sw = float(SW) / float(OW);
    const int64_t total = int64_t(B) * OH * OW;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int ow = int(i % OW), oh = int((i / OW) % OH), b = int(i / (int64_t(OW) * OH));
        int h0, h1, w0, w1; float lh, lw;
        bl_coord(oh, sh, SH, h0, h1, lh);
        bl_coord(ow, sw, SW, w0, w1, lw);
        const float* s = src + int64_t(b) * SH * SW;
        dst[i] = (1.f - lh) * ((1.f - lw) * s[h0 * SW + w0] + lw * s[h0 * SW + w1]) +
                 lh * ((1.f - lw) * s[h1 * SW + w0] + lw * s[h1 * SW + w1]);
    }
}

// adjoint of the resize as a GATHER over the source grid (no atomics: every source pixel sums the destination pixels whose
// footprint holds it, in raster order -- bitwise reproducible).  A destination row oh reads source rows h0 = floor(src), h1 = h0 + 1
// (clamped) with src = (oh + 0.5) * sh - 0.5: source row h can only be h0 or h1 of rows with src in (h - 1, h + 1), plus the clamped ends.
__device__ __forceinline__ void bl_window(int h, float s, int S, int O, int& lo, int& hi) {
    const float inv = 1.f / s;
    int a = int(floorf((float(h) - 1.f + 0.5f) * inv - 0.5f)) - 1, b = int(ceilf((float(h) + 1.f + 0.5f) * inv - 0.5f)) + 1;
    if (h == 0) a = 0;                      // negative source coordinates clamp onto row 0
    if (h == S - 1) b = O - 1;              // and rows past the end onto the last one
    lo = a < 0 ? 0 : a;
    hi = b > O - 1 ? O - 1 : b;
}

__global__ void bilinear_bwd_kernel(const float* __restrict__ dd, int B, int OH, int OW, float* __restrict__ ds, int SH, int SW) {
    const float sh = float(SH) / float(OH), sw = float(SW) / float(OW);
    const int64_t total = int64_t(B) * SH * SW;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int w = int(i % SW), h = int((i / SW) % SH), b = int(i / (int64_t(SW) * SH));
        int olo, ohi, wlo, whi;
        bl_window(h, sh, SH, OH, olo, ohi);
        bl_window(w, sw, SW, OW, wlo, whi);
        const float* d = dd + int64_t(b) * OH * OW;
        float acc = 0.f;
        for (int oh = olo; oh <= ohi; ++oh) {
            int h0, h1; float lh;
            bl_coord(oh, sh, SH, h0, h1, lh);
            const float ch = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
            if (ch == 0.f) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
                int w0, w1; float lw;
                bl_coord(ow, sw, SW, w0, w1, lw);
                const float cw = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
                if (cw != 0.f) acc += ch * cw * d[int64_t(oh) * OW + ow];
            }
        }
        ds[i] = acc;
    }
}

struct InjP {
    const float* z; const float* e; const float* scale; int style;
    int HW, W, C;
    float* out; int o_row, o_img, o_org;
    // backward
    const float* g; const float* a; int a_row, a_img, a_org;
    float* dz; float* de; float* dscale; float* ws;
};

// one wave per pixel group: lanes over channel quads
__global__ __launch_bounds__(256) void inject_fwd_kernel(const InjP p, int64_t npix) {
    const int q4 = p.C / 4;
    const float s = p.scale ? p.scale[0] : 1.f;
    const int64_t total = npix * q4;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int q = int(i % q4);
        const int64_t pix = i / q4;
        const int b = int(pix / p.HW), r = int(pix - int64_t(b) * p.HW);
        const int h = r / p.W, w = r - h * p.W;
        const float ev = p.e[pix];
        f32x4 v = *reinterpret_cast<const f32x4*>(p.z + pix * p.C + q * 4);
        if (p.style == 0) v = v * (1.f + s * ev); else v = v + s * ev;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        *reinterpret_cast<f32x4*>(p.out + int64_t(b) * p.o_img + p.o_org + int64_t(h) * p.o_row + int64_t(w) * p.C + q * 4) = v;
    }
}

// block = 256 threads = (256/q4) pixels x q4 quads per iteration; de needs a sum over channels
__global__ __launch_bounds__(256) void inject_bwd_kernel(const InjP p, int64_t npix) {
    __shared__ float red[256];
    const int q4 = p.C / 4 < 256 ? p.C / 4 : 256;
    const int ppi = 256 / q4;                     // pixels per iteration
    const int tid = threadIdx.x, q = tid % q4, pl = tid / q4;
    const float s = p.scale ? p.scale[0] : 1.f;
    float ds_acc = 0.f;
    for (int64_t base = int64_t(blockIdx.x) * ppi; base < npix; base += int64_t(gridDim.x) * ppi) {
        const int64_t pix = base + pl;
        float de_part = 0.f;
        if (pl < ppi && pix < npix) {
            const int b = int(pix / p.HW), r = int(pix - int64_t(b) * p.HW);
            const int h = r / p.W, w = r - h * p.W;
            const float ev = p.e[pix];
            for (int qq = q; qq < p.C / 4; qq += q4) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(p.g + pix * p.C + qq * 4);
                const f32x4 av = *reinterpret_cast<const f32x4*>(p.a + int64_t(b) * p.a_img + p.a_org + int64_t(h) * p.a_row + int64_t(w) * p.C + qq * 4);
                const f32x4 zv = *reinterpret_cast<const f32x4*>(p.z + pix * p.C + qq * 4);
                f32x4 gm, dz;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gm[k] = av[k] > 0.f ? gv[k] : 0.f;
                    if (p.style == 0) {
                        dz[k] = gm[k] * (1.f + s * ev);
                        de_part += gm[k] * zv[k] * s;
                        ds_acc += gm[k] * zv[k] * ev;
                    } else {
                        dz[k] = gm[k];
                        de_part += gm[k] * s;
                        ds_acc += gm[k] * ev;
                    }
                }
                *reinterpret_cast<f32x4*>(p.dz + pix * p.C + qq * 4) = dz;
            }
        }
        red[tid] = de_part;
        __syncthreads();
        if (q == 0 && pl < ppi && pix < npix) {
            float t = 0.f;
            for (int k = 0; k < q4; ++k) t += red[pl * q4 + k];
            p.de[pix] = t;
        }
        __syncthreads();
    }
    ds_acc = ng_wave_sum(ds_acc);
    if ((tid & 63) == 0) red[tid >> 6] = ds_acc;
    __syncthreads();
    if (tid == 0 && p.dscale) p.ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);      // summed in block order by ng_partials_finish
}

__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int64_t rows, int cols, float* __restrict__ out, int accumulate) {
    // block = 64 columns x 16 row lanes; a lane adds the rows rl, rl + 16, ... with four independent partial sums (four loads in flight),
    // the lanes are combined through LDS in lane order: a fixed association, whatever the launch.  (4 row lanes walking 256 rows each,
    // one load in flight, cost 61 us for the PatchGAN's 1 024 x 64 live-bias rows.)
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int64_t r = rl;
        for (; r + 48 < rows; r += 64) {
            s0 += x[r * cols + c];
            s1 += x[(r + 16) * cols + c];
            s2 += x[(r + 32) * cols + c];
            s3 += x[(r + 48) * cols + c];
        }
        for (; r < rows; r += 16) s0 += x[r * cols + c];
    }
    red[rl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ void fill_kernel(float* __restrict__ dst, int64_t n, float v) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) dst[i] = v;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] += a * x[i];
}


// model/generator_inject.py:133-134: x * post_correction_param (a learnable 0-dim parameter read from device memory)
__global__ __launch_bounds__(256) void param_scale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ param, float* __restrict__ out, int64_t n) {
    const float c = *param;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) out[i] = x[i] * c;
}
// its backward: gx = gout * c; ws[block] = sum over the block's elements of gout * x (block order fixed: summed by ng_partials_finish)
__global__ __launch_bounds__(256) void param_scale_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x, const float* __restrict__ param,
                                                               float* __restrict__ gx, float* __restrict__ ws, int64_t n) {
    __shared__ float red[4];
    const float c = *param;
    float acc = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const float g = gout[i];
        gx[i] = g * c;
        acc += g * x[i];
    }
    acc = ng_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// per-block partial sums -> dst, fixed association (see common.h::ng_partials_finish)
__global__ __launch_bounds__(256) void partials_finish_kernel(const float* __restrict__ ws, int rows, int nv, float* dst) {
    __shared__ float part[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = tid; r < rows; r += 256)
        for (int i = 0; i < nv; ++i) acc[i] += ws[size_t(r) * nv + i];
    for (int i = 0; i < nv; ++i) {
        const float v = ng_wave_sum(acc[i]);
        if (lane == 0) part[wave][i] = v;
    }
    __syncthreads();
    if (tid < nv) atomicAdd(dst + tid, (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
}

}  // namespace

int ng_partials_finish(const float* ws, int rows, int nv, float* dst, hipStream_t st) {
    NG_REQUIRE(ws && dst && rows > 0 && nv > 0 && nv <= 8, "partials_finish: bad arguments");
    hipLaunchKernelGGL(partials_finish_kernel, dim3(1), dim3(256), 0, st, ws, rows, nv, dst);
    return nirgan_check_launch("partials_finish");
}

extern "C" int nirgan_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                           float eps, int step, void* stream) {
    NG_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam: bad arguments");
    NG_REQUIRE(ng_aligned16(p) && ng_aligned16(g) && ng_aligned16(m) && ng_aligned16(v), "adam: pointers must be 16-byte aligned");
    const double bc1 = 1.0 - pow(double(beta1), double(step));
    const double bc2 = 1.0 - pow(double(beta2), double(step));
    const float step_size = float(double(lr) / bc1);
    const float inv_sqrt_bc2 = float(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 2048)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       p, g, m, v, n, beta1, beta2, eps, step_size, inv_sqrt_bc2);
    return nirgan_check_launch("adam");
}

extern "C" int nirgan_bilinear_fwd(const float* src, int B, int SH, int SW, float* dst, int OH, int OW, void* stream) {
    NG_REQUIRE(src && dst && B > 0 && SH > 0 && SW > 0 && OH > 0 && OW > 0, "bilinear_fwd: bad arguments");
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for(int64_t(B) * OH * OW)), dim3(256), 0, static_cast<hipStream_t>(stream), src, B, SH, SW, dst, OH, OW);
    return nirgan_check_launch("bilinear_fwd");
}

extern "C" int nirgan_bilinear_bwd(const float* ddst, int B, int OH, int OW, float* dsrc, int SH, int SW, void* stream) {
    NG_REQUIRE(ddst && dsrc && B > 0 && SH > 0 && SW > 0 && OH > 0 && OW > 0, "bilinear_bwd: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(int64_t(B) * SH * SW)), dim3(256), 0, st, ddst, B, OH, OW, dsrc, SH, SW);
    return nirgan_check_launch("bilinear_bwd");
}

extern "C" int nirgan_inject_fwd(const nirgan_inject_fwd_desc* d, void* stream) {
    NG_REQUIRE(d && d->z && d->e && d->out, "inject_fwd: null pointer");
    NG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C >= 4 && d->C % 4 == 0, "inject_fwd: bad shape");
    NG_REQUIRE(d->o_hp == d->H + 2 * d->o_pad && d->o_wp == d->W + 2 * d->o_pad, "inject_fwd: output geometry mismatch");
    InjP p = {};
    p.z = d->z; p.e = d->e; p.scale = d->scale; p.style = d->style; p.HW = d->H * d->W; p.W = d->W; p.C = d->C;
    p.out = d->out; p.o_row = d->o_wp * d->C; p.o_img = d->o_hp * p.o_row; p.o_org = d->o_pad * p.o_row + d->o_pad * d->C;
    const int64_t npix = int64_t(d->B) * p.HW;
    hipLaunchKernelGGL(inject_fwd_kernel, dim3(grid_for(npix * (d->C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), p, npix);
    return nirgan_check_launch("inject_fwd");
}

extern "C" int nirgan_inject_bwd(const nirgan_inject_bwd_desc* d, void* stream) {
    NG_REQUIRE(d && d->g && d->a && d->z && d->e && d->dz && d->de, "inject_bwd: null pointer");
    NG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C >= 4 && d->C % 4 == 0, "inject_bwd: bad shape");
    NG_REQUIRE(d->a_hp == d->H + 2 * d->a_pad && d->a_wp == d->W + 2 * d->a_pad, "inject_bwd: mask geometry mismatch");
    InjP p = {};
    p.z = d->z; p.e = d->e; p.scale = d->scale; p.style = d->style; p.HW = d->H * d->W; p.W = d->W; p.C = d->C;
    p.g = d->g; p.a = d->a; p.a_row = d->a_wp * d->C; p.a_img = d->a_hp * p.a_row; p.a_org = d->a_pad * p.a_row + d->a_pad * d->C;
    p.dz = d->dz; p.de = d->de; p.dscale = d->dscale;
    const int64_t npix = int64_t(d->B) * p.HW;
    const int q4 = d->C / 4 < 256 ? d->C / 4 : 256;
    const int ppi = 256 / q4;
    int64_t g = (npix + ppi - 1) / ppi;
    g = g < 2048 ? g : 2048;
    NG_REQUIRE(!d->dscale || (d->ws && d->ws_elems >= g), "inject_bwd: dscale needs a workspace of %lld floats (block partial sums, fixed-order finish)", (long long)g);
    p.ws = d->ws;
    hipLaunchKernelGGL(inject_bwd_kernel, dim3(int(g)), dim3(256), 0, static_cast<hipStream_t>(stream), p, npix);
    if (d->dscale) return ng_partials_finish(d->ws, int(g), 1, d->dscale, static_cast<hipStream_t>(stream));
    return nirgan_check_launch("inject_bwd");
}

extern "C" int nirgan_param_scale_fwd(const float* x, const float* param, float* out, int64_t n, void* stream) {
    NG_REQUIRE(x && param && out && n > 0, "param_scale_fwd: bad arguments");
    hipLaunchKernelGGL(param_scale_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, param, out, n);
    return nirgan_check_launch("param_scale_fwd");
}

extern "C" int nirgan_param_scale_bwd(const float* gout, const float* x, const float* param, float* gx, float* dparam, float* ws, int64_t ws_elems,
                                      int64_t n, void* stream) {
    NG_REQUIRE(gout && x && param && gx && dparam && ws && n > 0, "param_scale_bwd: bad arguments");
    int64_t g = (n + 255) / 256;
    g = g < 1024 ? g : 1024;                        // (a fixed grid for a fixed n: the partial sums' association never changes)
    NG_REQUIRE(ws_elems >= g, "param_scale_bwd: workspace holds %lld floats, %lld needed", (long long)ws_elems, (long long)g);
    hipLaunchKernelGGL(param_scale_bwd_kernel, dim3(int(g)), dim3(256), 0, static_cast<hipStream_t>(stream), gout, x, param, gx, ws, n);
    return ng_partials_finish(ws, int(g), 1, dparam, static_cast<hipStream_t>(stream));
}

extern "C" int nirgan_colsum(const float* x, int64_t rows, int cols, float* out, int accumulate, void* stream) {
    NG_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(1024), 0, static_cast<hipStream_t>(stream), x, rows, cols, out, accumulate);
    return nirgan_check_launch("colsum");
}

extern "C" int nirgan_fill(float* dst, int64_t n, float value, void* stream) {
    NG_REQUIRE(dst && n > 0, "fill: bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, n, value);
    return nirgan_check_launch("fill");
}

extern "C" int nirgan_axpy(float* y, const float* x, int64_t n, float alpha, void* stream) {
    NG_REQUIRE(y && x && n > 0, "axpy: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), y, x, n, alpha);
    return nirgan_check_launch("axpy");
}

extern "C" int nirgan_run_plan(const nirgan_plan_entry* entries, int n, void* stream) {
    NG_REQUIRE(entries || n == 0, "run_plan: null entries");
    for (int i = 0; i < n; ++i) {
        int rc;
        switch (entries[i].op) {
            case NIRGAN_OP_CONV: rc = nirgan_conv_igemm(static_cast<const nirgan_conv_desc*>(entries[i].desc), stream); break;
            case NIRGAN_OP_WGRAD: rc = nirgan_wgrad_igemm(static_cast<const nirgan_wgrad_desc*>(entries[i].desc), stream); break;
            case NIRGAN_OP_IN_FWD: rc = nirgan_instnorm_fwd(static_cast<const nirgan_in_fwd_desc*>(entries[i].desc), stream); break;
            case NIRGAN_OP_IN_BWD: rc = nirgan_instnorm_bwd(static_cast<const nirgan_in_bwd_desc*>(entries[i].desc), stream); break;
            default: nirgan_set_error("run_plan: unknown op %d at entry %d", entries[i].op, i); return NIRGAN_ERR_ARG;
        }
        if (rc != NIRGAN_OK) return rc;
    }
    return NIRGAN_OK;
}
