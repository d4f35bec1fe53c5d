// Device helpers of the instance-norm kernels that other translation units fuse into their own loaders (csrc/wino6.hip evaluates the
// second pass of the instance-norm backward inside its dY transform).
#pragma once
#include "common.h"

namespace {

__host__ __device__ inline int in_nrg(int C) {
    const int q4 = C / 4;
    return q4 >= 256 ? 1 : 256 / q4;
}

inline int in_nchunk(int B, int HW, int C) {
    const int nrg = in_nrg(C);
    int want = 2048 / B;          // 2 048 blocks: eight per CU, all resident (32 waves per CU); 1 024 left the passes latency-bound at 16 waves (bf16 step 12.56 -> 11.92 ms of op time)
    if (want < 1) want = 1;
    int cap = HW / (nrg * 8);
    if (cap < 1) cap = 1;
    return want < cap ? want : cap;
}

// padded-buffer coordinates that hold a copy of interior coordinate h under a reflect halo
// of width P (P < H): the interior itself plus its mirror images.  Returns the count (<= 3).
__device__ __forceinline__ int halo_images(int h, int H, int P, int* out) {
    int n = 0;
    out[n++] = h + P;
    if (h >= 1 && h <= P) out[n++] = P - h;
    if (h >= H - 1 - P && h <= H - 2) out[n++] = P + 2 * (H - 1) - h;
    return n;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// y (a convolution's output in front of its instance norm) is fp32, or -- bf16 operand mode, y16 -- stored as bf16 by the producing
// launch (nirgan_conv_desc.out_bf16; the statistics come from its fp32 accumulators): element offsets, 2 or 4 bytes each
__device__ __forceinline__ const float* y_at(const float* y, size_t elems, int y16) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(y) + elems * (y16 ? 2 : 4));
}
__device__ __forceinline__ f32x4 ldy4(const float* yb, size_t off, int y16) {
    if (!y16) return ld4(yb + off);
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const char*>(yb) + off * 2);
    return __builtin_convertvector(v, f32x4);
}
typedef __bf16 in_bf16x4 __attribute__((ext_vector_type(4)));
// the bf16 twin of a buffer: same element offset, value rounded to nearest even
__device__ __forceinline__ void st4_twin(unsigned short* twin, size_t off, f32x4 v) {
    if (twin) *reinterpret_cast<in_bf16x4*>(twin + off) = __builtin_convertvector(v, in_bf16x4);
}

struct InBwd {
    const float* g; int g_row, g_img, g_pad, g_fold;
    const float* g2;
    const float* a; int a_row, a_img, a_org;
    int act; float slope;
    const float* y; const float* mean; const float* rstd; int norm;
    int HW, W, H, C;
    float* dy; int d_row, d_img, d_org;
    float* gsum_out;
    float* dbias;
    float* ws; int nchunk, ppc;
    unsigned short* dy16;
    int y16, g16;              // y / g stored as bf16 (nirgan_in_bwd_desc.y_bf16 / g_bf16): read through ldy4
    int pchunks;               // chunks of partial sums per sample in ws (= nchunk, or the producer's count: nirgan_in_bwd_desc.sums_chunks)
};

// gradient wrt the block output at pixel (h, w) = pix of sample b: the (reflect-folded) halo'd gradient plus the dense skip gradient
__device__ __forceinline__ f32x4 in_bwd_gsum(const InBwd& p, const float* gb, const float* g2b, int h, int w, int pix, int q) {
    f32x4 ga = {0, 0, 0, 0};
    if (gb) {
        if (p.g_fold) {
            int hs[3], wsx[3];
            const int nh = halo_images(h, p.H, p.g_pad, hs), nw = halo_images(w, p.W, p.g_pad, wsx);
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) ga += ldy4(gb, size_t(hs[i]) * p.g_row + size_t(wsx[j]) * p.C + q * 4, p.g16);
        } else {
            ga = ldy4(gb, size_t(h + p.g_pad) * p.g_row + size_t(w + p.g_pad) * p.C + q * 4, p.g16);
        }
    }
    if (g2b) ga += ld4(g2b + size_t(pix) * p.C + q * 4);
    return ga;
}


// host: descriptor -> kernel parameters of the backward passes (validation lives in nirgan_instnorm_bwd)
inline InBwd in_bwd_params(const nirgan_in_bwd_desc* d) {
    InBwd p;
    p.g16 = d->g_bf16 ? 1 : 0;
    p.g = d->g; p.g_row = d->g_wp * d->C; p.g_img = d->g_hp * p.g_row; p.g_pad = d->g_pad; p.g_fold = d->g_fold;
    p.g2 = d->g2;
    p.a = d->a; p.a_row = d->a_wp * d->C; p.a_img = d->a_hp * p.a_row; p.a_org = d->a_pad * p.a_row + d->a_pad * d->C;
    p.act = d->act; p.slope = d->slope;
    p.y = d->y; p.y16 = d->y_bf16 ? 1 : 0; p.mean = d->mean; p.rstd = d->rstd; p.norm = d->norm;
    p.H = d->H; p.W = d->W; p.HW = d->H * d->W; p.C = d->C;
    p.dy = d->dy; p.d_row = d->d_wp * d->C; p.d_img = d->d_hp * p.d_row; p.d_org = d->d_pad * p.d_row + d->d_pad * d->C;
    p.gsum_out = d->gsum_out; p.dbias = d->dbias;
    p.ws = d->ws; p.nchunk = in_nchunk(d->B, p.HW, d->C);
    p.ppc = (p.HW + p.nchunk - 1) / p.nchunk;
    p.dy16 = static_cast<unsigned short*>(d->dy_bf16);
    p.pchunks = d->norm && d->sums_chunks > 0 ? d->sums_chunks : p.nchunk;
    return p;
}

// dy of the instance-norm backward at pixel (h, w) = pix of sample b, channel quad q -- EXACTLY the arithmetic of in_bwd_pass2_kernel
// (m1, m2 = mean(g_z), mean(g_z * z) of (b, q) from the finalize pass; gb / g2b / gsb / yb = the sample's base pointers)
__device__ __forceinline__ f32x4 in_bwd_dy(const InBwd& p, const float* gb, const float* g2b, const float* gsb, const float* yb,
                                           const f32x4 mean, const f32x4 rstd, const f32x4 m1, const f32x4 m2, int h, int w, int q) {
    const int pix = h * p.W + w;
    f32x4 gz = gsb ? ld4(gsb + size_t(pix) * p.C + q * 4) : in_bwd_gsum(p, gb, g2b, h, w, pix, q);
    const f32x4 z = (ldy4(yb, size_t(pix) * p.C + q * 4, p.y16) - mean) * rstd;
    if (p.act == NIRGAN_ACT_RELU || p.act == NIRGAN_ACT_LRELU) {
        const float neg = p.act == NIRGAN_ACT_RELU ? 0.f : p.slope;
#pragma unroll
        for (int i = 0; i < 4; ++i) gz[i] = z[i] > 0.f ? gz[i] : gz[i] * neg;
    }
    return rstd * (gz - m1 - z * m2);
}

}  // namespace
