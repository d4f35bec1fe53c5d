// InstanceNorm2d(affine=False) + activation + residual + halo write, forward and backward,
// on dense NHWC fp32.  HBM-bound: lanes run along C (float4 per lane, fully coalesced rows),
// pixel chunks spread over the grid, per-(b,c) sums combined through a small partial buffer
// in a fixed order (deterministic).  Sums are taken about a per-channel shift (the value at
// pixel 0) so that E[x^2]-E[x]^2 does not cancel.
#include "instnorm_dev.h"
#include <type_traits>

namespace {

// block-level sum over the row groups of two float4 accumulators; result valid for tid < q4
__device__ __forceinline__ void rg_reduce2(f32x4& s1, f32x4& s2, f32x4* lds, int tid, int q4, int nrg) {
    lds[tid] = s1;
    lds[256 + tid] = s2;
    __syncthreads();
    if (tid < q4) {
        f32x4 a = lds[tid], b = lds[256 + tid];
        for (int r = 1; r < nrg; ++r) {
            a += lds[r * q4 + tid];
            b += lds[256 + r * q4 + tid];
        }
        s1 = a;
        s2 = b;
    }
    __syncthreads();
}

struct InFwd {
    const float* y; int y16; int HW, W, H, C;      // y16: y is stored as bf16 (instnorm_dev.h::ldy4)
    int norm; float eps;
    float* mean; float* rstd;
    int act; float slope;
    const float* residual; int r_row, r_img, r_org;
    float* out; int o_row, o_img, o_pad, border;
    float* ws; int nchunk, ppc;
    unsigned short* out16;
    int pre_chunks; const float* shift;      // partial sums left by the producer of y: chunk count and the shift they are taken about
};

__global__ __launch_bounds__(256) void in_stats_kernel(const InFwd p) {
    __shared__ f32x4 lds[512];
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4;
    const float* yb = y_at(p.y, size_t(b) * p.HW * p.C, p.y16);
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rg < nrg) {
        const f32x4 k = ldy4(yb, q * 4, p.y16);
        const int start = chunk * p.ppc;
        int end = start + p.ppc;
        end = end < p.HW ? end : p.HW;
        for (int pix = start + rg; pix < end; pix += nrg) {
            const f32x4 v = ldy4(yb, size_t(pix) * p.C + q * 4, p.y16) - k;
            s1 += v;
            s2 += v * v;
        }
    }
    rg_reduce2(s1, s2, lds, tid, q4, nrg);
    if (tid < q4) {
        float* w = p.ws + (size_t(b) * p.nchunk + chunk) * 2 * p.C;
        st4(w + tid * 4, s1);
        st4(w + p.C + tid * 4, s2);
    }
}

// sum of the chunk partials of sample b (blockIdx.x): row group rg adds chunks rg, rg+nrg, ... (independent loads, 8 in
// flight), the groups are combined through LDS in group order -- a fixed order, so the result is reproducible.
// Result valid for tid < q4.  (One thread per channel quad walking all chunks serially cost ~17 us of pure latency.)
// A block takes the channels [c0, c0 + cw) (cw = 64 for C >= 64 and C % 64 == 0 -- blockIdx.y slices of 64 channels: 16 row groups
// instead of 4 at C = 256, a quarter of the serial loads per thread -- else the whole C); result valid for tid < cw / 4.
__device__ __forceinline__ int fin_cw(int C) { return (C >= 64 && C % 64 == 0) ? 64 : C; }
__device__ __forceinline__ void chunk_sums(const float* w, int nchunk, int C, int tid, f32x4* lds, f32x4& s1, f32x4& s2, int c0 = 0, int cw = 0) {
    cw = cw > 0 ? cw : C;
    const int q4 = cw / 4, nrg = in_nrg(cw);
    const int q = tid % q4, rg = tid / q4;
    s1 = f32x4{0, 0, 0, 0};
    s2 = f32x4{0, 0, 0, 0};
    if (rg < nrg) {
#pragma unroll 8
        for (int c = rg; c < nchunk; c += nrg) {
            s1 += ld4(w + size_t(c) * 2 * C + c0 + q * 4);
            s2 += ld4(w + size_t(c) * 2 * C + C + c0 + q * 4);
        }
    }
    rg_reduce2(s1, s2, lds, tid, q4, nrg);
}

// The same for partial sums a PRODUCER left (convolution epilogue, Winograd output transform): per chunk and channel [k, s1, s2, n] =
// shift, sum (v - k), sum (v - k)^2, count.  Every chunk is re-based onto the first chunk's shift K (values of the data itself, so
// k - K is of the order of the spread): sum (v - K) = s1 + n (k - K), sum (v - K)^2 = s2 + 2 (k - K) s1 + n (k - K)^2 -- no term is
// large against the variance, whatever the channel's mean.  Row groups add their chunks in a fixed order.  Result valid for tid < cw / 4.
__device__ __forceinline__ void chunk_sums_rebased(const float* w, int nchunk, int C, int tid, f32x4* lds, f32x4& s1, f32x4& s2, f32x4& K, int c0, int cw) {
    const int q4 = cw / 4, nrg = in_nrg(cw);
    const int q = tid % q4, rg = tid / q4;
    s1 = f32x4{0, 0, 0, 0};
    s2 = f32x4{0, 0, 0, 0};
    K = ld4(w + c0 + q * 4);
    if (rg < nrg) {
#pragma unroll 4
        for (int c = rg; c < nchunk; c += nrg) {
            const float* wc = w + size_t(c) * 4 * C + c0 + q * 4;
            const f32x4 dk = ld4(wc) - K, a1 = ld4(wc + C), a2 = ld4(wc + 2 * C), n = ld4(wc + 3 * C);
            s1 += a1 + n * dk;
            s2 += a2 + 2.f * dk * a1 + n * dk * dk;
        }
    }
    rg_reduce2(s1, s2, lds, tid, q4, nrg);
}

// one block per sample: chunk partials -> mean, rstd
__global__ __launch_bounds__(256) void in_finalize_kernel(const InFwd p, int B) {
    __shared__ f32x4 lds[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    f32x4 s1, s2, k = {0.f, 0.f, 0.f, 0.f};
    const int cw = fin_cw(p.C), c0 = blockIdx.y * cw;
    if (p.pre_chunks > 0) chunk_sums_rebased(p.ws + size_t(b) * p.pre_chunks * 4 * p.C, p.pre_chunks, p.C, tid, lds, s1, s2, k, c0, cw);
    else chunk_sums(p.ws + size_t(b) * p.nchunk * 2 * p.C, p.nchunk, p.C, tid, lds, s1, s2, c0, cw);
    if (tid >= cw / 4) return;
    const int q = c0 / 4 + tid;
    const float inv = 1.f / float(p.HW);
    if (p.pre_chunks > 0) { if (p.shift != nullptr) k += ld4(p.shift + q * 4); }      // the producer's values exclude its bias
    else k = ldy4(y_at(p.y, size_t(b) * p.HW * p.C, p.y16), q * 4, p.y16);
    const f32x4 m = s1 * inv;
    f32x4 var = s2 * inv - m * m, rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        var[j] = var[j] > 0.f ? var[j] : 0.f;
        rstd[j] = 1.f / sqrtf(var[j] + p.eps);
    }
    st4(p.mean + size_t(b) * p.C + q * 4, k + m);
    st4(p.rstd + size_t(b) * p.C + q * 4, rstd);
}

__device__ __forceinline__ f32x4 act4(f32x4 z, int act, float slope) {
    if (act == NIRGAN_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = z[i] > 0.f ? z[i] : 0.f;
    } else if (act == NIRGAN_ACT_LRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = z[i] > 0.f ? z[i] : z[i] * slope;
    }
    return z;
}

__global__ __launch_bounds__(256) void in_apply_kernel(const InFwd p) {
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4;
    if (rg >= nrg) return;
    const float* yb = y_at(p.y, size_t(b) * p.HW * p.C, p.y16);
    f32x4 mean = {0, 0, 0, 0}, rstd = {1, 1, 1, 1};
    if (p.norm) {
        mean = ld4(p.mean + size_t(b) * p.C + q * 4);
        rstd = ld4(p.rstd + size_t(b) * p.C + q * 4);
    }
    const int start = chunk * p.ppc;
    int end = start + p.ppc;
    end = end < p.HW ? end : p.HW;
    float* ob = p.out ? p.out + size_t(b) * p.o_img : nullptr;
    unsigned short* ob16 = p.out16 ? p.out16 + size_t(b) * p.o_img : nullptr;
    const float* rb = p.residual ? p.residual + size_t(b) * p.r_img + p.r_org : nullptr;
    // (h, w) of the thread's pixel advance with it: one integer division per thread instead of one per pixel (a 32-bit division is ~25
    // VALU instructions; these passes run at four waves per SIMD, where that issue time is not hidden)
    int h = (start + rg) / p.W, w = (start + rg) - h * p.W;
    for (int pix = start + rg; pix < end; pix += nrg, w += nrg) {
        while (w >= p.W) { w -= p.W; ++h; }
        f32x4 v = (ldy4(yb, size_t(pix) * p.C + q * 4, p.y16) - mean) * rstd;
        v = act4(v, p.act, p.slope);
        if (rb) v += ld4(rb + size_t(h) * p.r_row + size_t(w) * p.C + q * 4);
        if (p.border == NIRGAN_BORDER_REFLECT && p.o_pad > 0) {
            int hs[3], wsx[3];
            const int nh = halo_images(h, p.H, p.o_pad, hs), nw = halo_images(w, p.W, p.o_pad, wsx);
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) {
                    const size_t off = size_t(hs[i]) * p.o_row + size_t(wsx[j]) * p.C + q * 4;
                    if (ob) st4(ob + off, v);
                    st4_twin(ob16, off, v);
                }
        } else {
            const size_t off = size_t(h + p.o_pad) * p.o_row + size_t(w + p.o_pad) * p.C + q * 4;
            if (ob) st4(ob + off, v);
            st4_twin(ob16, off, v);
        }
    }
}

// with norm: the sums of g_z and g_z * z only (pass 2 rebuilds g_z from the same inputs instead of reading it back: one tensor write
// less); without norm: dy = g_z, written here
__global__ __launch_bounds__(256) void in_bwd_pass1_kernel(const InBwd p) {
    __shared__ f32x4 lds[512];
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rg < nrg) {
        f32x4 mean = {0, 0, 0, 0}, rstd = {1, 1, 1, 1};
        if (p.norm) {
            mean = ld4(p.mean + size_t(b) * p.C + q * 4);
            rstd = ld4(p.rstd + size_t(b) * p.C + q * 4);
        }
        const float* gb = p.g ? y_at(p.g, size_t(b) * p.g_img, p.g16) : nullptr;
        const float* g2b = p.g2 ? p.g2 + size_t(b) * p.HW * p.C : nullptr;
        const float* yb = p.y ? y_at(p.y, size_t(b) * p.HW * p.C, p.y16) : nullptr;
        float* db = p.dy + size_t(b) * p.d_img + p.d_org;
        const int start = chunk * p.ppc;
        int end = start + p.ppc;
        end = end < p.HW ? end : p.HW;
        int h = (start + rg) / p.W, w = (start + rg) - h * p.W;          // advanced with the pixel (see in_apply_kernel)
        for (int pix = start + rg; pix < end; pix += nrg, w += nrg) {
            while (w >= p.W) { w -= p.W; ++h; }
            const f32x4 ga = in_bwd_gsum(p, gb, g2b, h, w, pix, q);
            if (p.gsum_out) st4(p.gsum_out + (size_t(b) * p.HW + pix) * p.C + q * 4, ga);
            // z = (y - mean) * rstd is recomputed exactly as the forward computed it: its sign is the activation
            // mask (ReLU / LeakyReLU), so the activated tensor does not have to be read back
            f32x4 z = {0, 0, 0, 0};
            const bool masked = p.act == NIRGAN_ACT_RELU || p.act == NIRGAN_ACT_LRELU;
            if (p.norm || masked) z = (ldy4(yb, size_t(pix) * p.C + q * 4, p.y16) - mean) * rstd;
            f32x4 gz = ga;
            if (masked) {
                const float neg = p.act == NIRGAN_ACT_RELU ? 0.f : p.slope;
#pragma unroll
                for (int i = 0; i < 4; ++i) gz[i] = z[i] > 0.f ? gz[i] : gz[i] * neg;
            }
            if (!p.norm) {                                                                      // no second pass: this IS dy
                const size_t off = size_t(h) * p.d_row + size_t(w) * p.C + q * 4;
                st4(db + off, gz);
                st4_twin(p.dy16 ? p.dy16 + size_t(b) * p.d_img + p.d_org : nullptr, off, gz);
            }
            s1 += gz;
            if (p.norm) s2 += gz * z;
        }
    }
    rg_reduce2(s1, s2, lds, tid, q4, nrg);
    if (tid < q4) {
        if (p.norm) {
            float* w = p.ws + (size_t(b) * p.nchunk + chunk) * 2 * p.C;
            st4(w + tid * 4, s1);
            st4(w + p.C + tid * 4, s2);
        } else if (p.dbias) {
            // live bias (no instance norm behind it): this block's channel sums as one row of ws; nirgan_colsum adds the rows in order
            st4(p.ws + (size_t(b) * p.nchunk + chunk) * p.C + tid * 4, s1);
        }
    }
}

// one block per sample: mean(g_z), mean(g_z * z) from the chunk partials, stored behind them in ws
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(const InBwd p, int B) {
    __shared__ f32x4 lds[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    f32x4 s1, s2;
    const int cw = fin_cw(p.C), c0 = blockIdx.y * cw;
    chunk_sums(p.ws + size_t(b) * p.pchunks * 2 * p.C, p.pchunks, p.C, tid, lds, s1, s2, c0, cw);
    if (tid >= cw / 4) return;
    const float inv = 1.f / float(p.HW);
    float* m = p.ws + size_t(B) * p.pchunks * 2 * p.C + size_t(b) * 2 * p.C + c0;
    st4(m + tid * 4, s1 * inv);
    st4(m + p.C + tid * 4, s2 * inv);
}

__global__ __launch_bounds__(256) void in_bwd_pass2_kernel(const InBwd p, int B) {
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4;
    if (rg >= nrg) return;
    const float* mm = p.ws + size_t(B) * p.pchunks * 2 * p.C + size_t(b) * 2 * p.C;
    const f32x4 m1 = ld4(mm + q * 4), m2 = ld4(mm + p.C + q * 4);
    const f32x4 mean = ld4(p.mean + size_t(b) * p.C + q * 4);
    const f32x4 rstd = ld4(p.rstd + size_t(b) * p.C + q * 4);
    const float* yb = y_at(p.y, size_t(b) * p.HW * p.C, p.y16);
    // g_z again, from what pass 1 read: the folded sum it stored for the skip path when there is one, else the halo'd gradient itself
    const float* gsb = p.gsum_out ? p.gsum_out + size_t(b) * p.HW * p.C : nullptr;
    const float* gb = p.g ? y_at(p.g, size_t(b) * p.g_img, p.g16) : nullptr;
    const float* g2b = p.g2 ? p.g2 + size_t(b) * p.HW * p.C : nullptr;
    float* db = p.dy ? p.dy + size_t(b) * p.d_img + p.d_org : nullptr;
    const int start = chunk * p.ppc;
    int end = start + p.ppc;
    end = end < p.HW ? end : p.HW;
    int h = (start + rg) / p.W, w = (start + rg) - h * p.W;              // advanced with the pixel (see in_apply_kernel)
    for (int pix = start + rg; pix < end; pix += nrg, w += nrg) {
        while (w >= p.W) { w -= p.W; ++h; }
        const size_t off = size_t(h) * p.d_row + size_t(w) * p.C + q * 4;
        const f32x4 r = in_bwd_dy(p, gb, g2b, gsb, yb, mean, rstd, m1, m2, h, w, q);
        if (db) st4(db + off, r);
        st4_twin(p.dy16 ? p.dy16 + size_t(b) * p.d_img + p.d_org : nullptr, off, r);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the three per-pixel passes (apply, backward pass 1, backward pass 2) without divergent control flow.  The kernels above walk a
// pixel's reflect images through two small arrays and nested loops with per-lane trip counts, address everything with 64-bit products
// and branch on six run-time flags per pixel: ~15 branches and several hundred issue cycles per 8-16 bytes moved.  In the bf16 operand
// mode (8-byte accesses) that, not HBM, set their rate: 2.5-3 TB/s on a residual-trunk layer where a plain stream of the same bytes runs
// at 5.7-6.3 (scripts/diag/stream_bf16.hip), 21 % of the bf16 step.  Here: the element types and the halo / fold forms are template
// parameters, a pixel has at most ONE mirror image per dimension (host: H, W > 2 pad + 1; smaller images keep the general kernels), so
// the images are at most three masked accesses, and offsets inside a sample are 32-bit.  Same arithmetic per element and the same
// summation order as the general kernels (a gradient's images are added main, column mirror, row mirror, corner).
template <bool I16> __device__ __forceinline__ f32x4 ldx4(const void* base, int off) {
    if constexpr (I16) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        return __builtin_convertvector(*reinterpret_cast<const bf16x4_t*>(static_cast<const unsigned short*>(base) + off), f32x4);
    } else {
        return ld4(static_cast<const float*>(base) + off);
    }
}
// mirror image of interior coordinate h under a reflect halo of width P (-1: none); H > 2 P + 1
__device__ __forceinline__ int mirror1(int h, int H, int P) {
    return (h >= 1 && h <= P) ? P - h : ((h >= H - 1 - P && h <= H - 2) ? P + 2 * (H - 1) - h : -1);
}

// A thread's pixels are taken IN_U at a time: the IN_U independent loads of every stream are issued before the first use, so a wave has
// IN_U x 512 B (bf16: 8 B per lane) per stream in flight instead of one load.  From cold HBM (inside a step the tensors have left the
// last-level cache: scripts/bench_instnorm.py with SETS=6) one load in flight per wave left the bf16 passes latency-bound at 3.4 TB/s
// against 5.0 for the same passes on fp32 storage.
constexpr int IN_U = 4;
template <int U> using in_uc = std::integral_constant<int, U>;

// the thread's next U pixels: (h, w) and the dense element offset of each, the cursor advanced past them
template <int U>
__device__ __forceinline__ void in_next_pixels(int& h, int& w, int& yoff, int W, int nrg, int ystep, int (&hh)[U], int (&ww)[U], int (&yo)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        while (w >= W) { w -= W; ++h; }
        hh[u] = h; ww[u] = w; yo[u] = yoff;
        w += nrg; yoff += ystep;
    }
}

// OUT: 0 = fp32 only, 1 = fp32 + bf16 twin, 2 = twin only
template <bool Y16, int OUT, bool REFLECT, bool RESID>
__global__ __launch_bounds__(256) void in_apply_fast_kernel(const InFwd p) {
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4;
    if (rg >= nrg) return;
    const void* yb = y_at(p.y, size_t(b) * p.HW * p.C, p.y16);
    f32x4 mean = {0, 0, 0, 0}, rstd = {1, 1, 1, 1};
    if (p.norm) {
        mean = ld4(p.mean + size_t(b) * p.C + q * 4);
        rstd = ld4(p.rstd + size_t(b) * p.C + q * 4);
    }
    const float neg = p.act == NIRGAN_ACT_LRELU ? p.slope : 1.f;
    const bool relu = p.act == NIRGAN_ACT_RELU;
    const int start = chunk * p.ppc;
    int end = start + p.ppc;
    end = end < p.HW ? end : p.HW;
    float* const ob = OUT != 2 ? p.out + size_t(b) * p.o_img : nullptr;
    unsigned short* const ob16 = OUT != 0 ? p.out16 + size_t(b) * p.o_img : nullptr;
    const float* const rb = RESID ? p.residual + size_t(b) * p.r_img + p.r_org : nullptr;
    const int P = p.o_pad, C = p.C, q0 = q * 4;
    auto store = [&](int off, const f32x4 v) {
        if constexpr (OUT != 2) st4(ob + off, v);
        if constexpr (OUT != 0) *reinterpret_cast<in_bf16x4*>(ob16 + off) = __builtin_convertvector(v, in_bf16x4);
    };
    int h = (start + rg) / p.W, w = (start + rg) - h * p.W;
    int yoff = (start + rg) * C + q0;
    const int ystep = nrg * C;
    auto group = [&](auto uc) {
        constexpr int U = decltype(uc)::value;
        int hh[U], ww[U], yo[U];
        in_next_pixels<U>(h, w, yoff, p.W, nrg, ystep, hh, ww, yo);
        f32x4 v[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ldx4<Y16>(yb, yo[u]);
        if constexpr (RESID) {
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = ld4(rb + hh[u] * p.r_row + ww[u] * C + q0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 x = (v[u] - mean) * rstd;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = x[i] > 0.f ? x[i] : (relu ? 0.f : x[i] * neg);      // (ReLU stores +0, as act4 does)
            if constexpr (RESID) x += r[u];
            const int orow = (hh[u] + P) * p.o_row, ocol = (ww[u] + P) * C + q0;
            store(orow + ocol, x);
            if constexpr (REFLECT) {
                const int hm = mirror1(hh[u], p.H, P), wm = mirror1(ww[u], p.W, P);
                if (wm >= 0) store(orow + wm * C + q0, x);
                if (hm >= 0) {
                    store(hm * p.o_row + ocol, x);
                    if (wm >= 0) store(hm * p.o_row + wm * C + q0, x);
                }
            }
        }
    };
    int pix = start + rg;
    for (; pix + (IN_U - 1) * nrg < end; pix += IN_U * nrg) group(in_uc<IN_U>{});
    for (; pix < end; pix += nrg) group(in_uc<1>{});
}

// the gradient wrt the block output at U pixels: the halo'd gradient's interior images loaded first (independent), the rare mirror images
// added behind them in the general kernels' order, then the dense skip gradient
template <int U, bool G16, bool FOLD>
__device__ __forceinline__ void gsum_fast(const InBwd& p, const void* gb, const float* g2b, const int (&hh)[U], const int (&ww)[U], const int (&yo)[U], int q0, f32x4 (&ga)[U]) {
    const int P = p.g_pad, C = p.C;
    f32x4 g2v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ga[u] = ldx4<G16>(gb, (hh[u] + P) * p.g_row + (ww[u] + P) * C + q0);
    if (g2b) {
#pragma unroll
        for (int u = 0; u < U; ++u) g2v[u] = ld4(g2b + yo[u]);
    }
    if constexpr (FOLD) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int grow = (hh[u] + P) * p.g_row, gcol = (ww[u] + P) * C + q0;
            const int hm = mirror1(hh[u], p.H, P), wm = mirror1(ww[u], p.W, P);
            if (wm >= 0) ga[u] += ldx4<G16>(gb, grow + wm * C + q0);
            if (hm >= 0) {
                ga[u] += ldx4<G16>(gb, hm * p.g_row + gcol);
                if (wm >= 0) ga[u] += ldx4<G16>(gb, hm * p.g_row + wm * C + q0);
            }
        }
    }
    if (g2b) {
#pragma unroll
        for (int u = 0; u < U; ++u) ga[u] += g2v[u];
    }
}

// with norm and a halo'd gradient g: the sums of g_z and g_z * z (the general kernel's norm branch)
template <bool Y16, bool G16, bool FOLD>
__global__ __launch_bounds__(256) void in_bwd_pass1_fast_kernel(const InBwd p) {
    __shared__ f32x4 lds[512];
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4, q0 = q * 4;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rg < nrg) {
        const f32x4 mean = ld4(p.mean + size_t(b) * p.C + q0), rstd = ld4(p.rstd + size_t(b) * p.C + q0);
        const void* gb = y_at(p.g, size_t(b) * p.g_img, p.g16);
        const float* g2b = p.g2 ? p.g2 + size_t(b) * p.HW * p.C : nullptr;
        const void* yb = y_at(p.y, size_t(b) * p.HW * p.C, p.y16);
        float* gso = p.gsum_out ? p.gsum_out + size_t(b) * p.HW * p.C : nullptr;
        const float neg = p.act == NIRGAN_ACT_RELU ? 0.f : (p.act == NIRGAN_ACT_LRELU ? p.slope : 1.f);
        const int start = chunk * p.ppc;
        int end = start + p.ppc;
        end = end < p.HW ? end : p.HW;
        int h = (start + rg) / p.W, w = (start + rg) - h * p.W;
        int yoff = (start + rg) * p.C + q0;
        const int ystep = nrg * p.C;
        auto group = [&](auto uc) {
            constexpr int U = decltype(uc)::value;
            int hh[U], ww[U], yo[U];
            in_next_pixels<U>(h, w, yoff, p.W, nrg, ystep, hh, ww, yo);
            f32x4 yv[U], ga[U];
#pragma unroll
            for (int u = 0; u < U; ++u) yv[u] = ldx4<Y16>(yb, yo[u]);
            gsum_fast<U, G16, FOLD>(p, gb, g2b, hh, ww, yo, q0, ga);
#pragma unroll
            for (int u = 0; u < U; ++u) {                                  // (pixel order: the sums are those of the general kernel)
                if (gso) st4(gso + yo[u], ga[u]);
                const f32x4 z = (yv[u] - mean) * rstd;
                f32x4 gz = ga[u];
#pragma unroll
                for (int i = 0; i < 4; ++i) gz[i] = z[i] > 0.f ? gz[i] : gz[i] * neg;
                s1 += gz;
                s2 += gz * z;
            }
        };
        int pix = start + rg;
        for (; pix + (IN_U - 1) * nrg < end; pix += IN_U * nrg) group(in_uc<IN_U>{});
        for (; pix < end; pix += nrg) group(in_uc<1>{});
    }
    rg_reduce2(s1, s2, lds, tid, q4, nrg);
    if (tid < q4) {
        float* w = p.ws + (size_t(b) * p.nchunk + chunk) * 2 * p.C;
        st4(w + tid * 4, s1);
        st4(w + p.C + tid * 4, s2);
    }
}

// DY: 0 = fp32 only, 1 = fp32 + twin, 2 = twin only
template <bool Y16, bool G16, bool FOLD, int DY>
__global__ __launch_bounds__(256) void in_bwd_pass2_fast_kernel(const InBwd p, int B) {
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int q4 = p.C / 4, nrg = in_nrg(p.C);
    const int q = tid % q4, rg = tid / q4, q0 = q * 4;
    if (rg >= nrg) return;
    const float* mm = p.ws + size_t(B) * p.pchunks * 2 * p.C + size_t(b) * 2 * p.C;
    const f32x4 m1 = ld4(mm + q0), m2 = ld4(mm + p.C + q0);
    const f32x4 mean = ld4(p.mean + size_t(b) * p.C + q0), rstd = ld4(p.rstd + size_t(b) * p.C + q0);
    const void* yb = y_at(p.y, size_t(b) * p.HW * p.C, p.y16);
    const float* gsb = p.gsum_out ? p.gsum_out + size_t(b) * p.HW * p.C : nullptr;
    const void* gb = p.g ? y_at(p.g, size_t(b) * p.g_img, p.g16) : nullptr;
    const float* g2b = p.g2 ? p.g2 + size_t(b) * p.HW * p.C : nullptr;
    float* const db = DY != 2 ? p.dy + size_t(b) * p.d_img + p.d_org : nullptr;
    unsigned short* const db16 = DY != 0 ? p.dy16 + size_t(b) * p.d_img + p.d_org : nullptr;
    const float neg = p.act == NIRGAN_ACT_RELU ? 0.f : (p.act == NIRGAN_ACT_LRELU ? p.slope : 1.f);
    const int start = chunk * p.ppc;
    int end = start + p.ppc;
    end = end < p.HW ? end : p.HW;
    int h = (start + rg) / p.W, w = (start + rg) - h * p.W;
    int yoff = (start + rg) * p.C + q0;
    const int ystep = nrg * p.C;
    // g_z again, from what pass 1 read: the folded sum it stored for the skip path when there is one (SUMS), else the halo'd gradient
    auto group = [&](auto uc, auto sums) {
        constexpr int U = decltype(uc)::value;
        int hh[U], ww[U], yo[U];
        in_next_pixels<U>(h, w, yoff, p.W, nrg, ystep, hh, ww, yo);
        f32x4 yv[U], gz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) yv[u] = ldx4<Y16>(yb, yo[u]);
        if constexpr (decltype(sums)::value) {
#pragma unroll
            for (int u = 0; u < U; ++u) gz[u] = ld4(gsb + yo[u]);
        } else {
            gsum_fast<U, G16, FOLD>(p, gb, g2b, hh, ww, yo, q0, gz);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 z = (yv[u] - mean) * rstd;
#pragma unroll
            for (int i = 0; i < 4; ++i) gz[u][i] = z[i] > 0.f ? gz[u][i] : gz[u][i] * neg;
            const f32x4 r = rstd * (gz[u] - m1 - z * m2);
            const int off = hh[u] * p.d_row + ww[u] * p.C + q0;
            if constexpr (DY != 2) st4(db + off, r);
            if constexpr (DY != 0) *reinterpret_cast<in_bf16x4*>(db16 + off) = __builtin_convertvector(r, in_bf16x4);
        }
    };
    int pix = start + rg;
    if (gsb) {
        for (; pix + (IN_U - 1) * nrg < end; pix += IN_U * nrg) group(in_uc<IN_U>{}, std::true_type{});
        for (; pix < end; pix += nrg) group(in_uc<1>{}, std::true_type{});
    } else {
        for (; pix + (IN_U - 1) * nrg < end; pix += IN_U * nrg) group(in_uc<IN_U>{}, std::false_type{});
        for (; pix < end; pix += nrg) group(in_uc<1>{}, std::false_type{});
    }
}

// whether the fast forms cover a launch: every in-sample offset fits 31 bits, at most one mirror image per dimension
inline bool in_fast_ok(int H, int W, int C, int pad, long long img_elems) {
    return img_elems < (1ll << 31) && (pad == 0 || (H > 2 * pad + 1 && W > 2 * pad + 1));
}

}  // namespace

extern "C" int64_t nirgan_instnorm_ws_elems(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return int64_t(B) * in_nchunk(B, H * W, C) * 2 * C + int64_t(B) * 2 * C;
}

extern "C" int nirgan_instnorm_fwd(const nirgan_in_fwd_desc* d, void* stream) {
    NG_REQUIRE(d && d->y && (d->out || d->out_bf16 || d->norm), "instnorm_fwd: null pointer");
    // neither output: mean / rstd only, the consumer normalises on the fly (nirgan_wino6_input_norm); out_bf16 alone: every consumer reads the
    // bf16 twin (bf16 operand mode), the fp32 tensor is not stored
    const bool stats_only = d->out == nullptr && d->out_bf16 == nullptr;
    NG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C >= 4 && d->C % 4 == 0 && d->C <= 1024, "instnorm_fwd: bad shape B=%d H=%d W=%d C=%d", d->B, d->H, d->W, d->C);
    NG_REQUIRE(ng_aligned16(d->y) && ng_aligned16(d->out) && ng_aligned16(d->residual), "instnorm_fwd: pointers must be 16-byte aligned");
    NG_REQUIRE(stats_only || (d->o_pad >= 0 && d->o_hp == d->H + 2 * d->o_pad && d->o_wp == d->W + 2 * d->o_pad), "instnorm_fwd: output halo geometry mismatch");
    NG_REQUIRE(d->border != NIRGAN_BORDER_REFLECT || (d->o_pad < d->H && d->o_pad < d->W), "instnorm_fwd: reflect halo wider than the image");
    NG_REQUIRE(!d->residual || (d->r_hp == d->H + 2 * d->r_pad && d->r_wp == d->W + 2 * d->r_pad), "instnorm_fwd: residual geometry mismatch");
    InFwd p;
    p.y = d->y; p.y16 = d->y_bf16 ? 1 : 0; p.H = d->H; p.W = d->W; p.HW = d->H * d->W; p.C = d->C;
    p.norm = d->norm; p.eps = d->eps; p.mean = d->mean; p.rstd = d->rstd; p.act = d->act; p.slope = d->slope;
    p.residual = d->residual; p.r_row = d->r_wp * d->C; p.r_img = d->r_hp * p.r_row; p.r_org = d->r_pad * p.r_row + d->r_pad * d->C;
    p.out = d->out; p.o_row = d->o_wp * d->C; p.o_img = d->o_hp * p.o_row; p.o_pad = d->o_pad; p.border = d->border;
    p.ws = d->ws; p.nchunk = in_nchunk(d->B, p.HW, d->C); p.ppc = (p.HW + p.nchunk - 1) / p.nchunk;
    p.out16 = static_cast<unsigned short*>(d->out_bf16);
    NG_REQUIRE(!d->out_bf16 || (d->C % 8 == 0 && (reinterpret_cast<uintptr_t>(d->out_bf16) & 15) == 0), "instnorm_fwd: bf16 twin needs C %% 8 == 0 and 16-byte alignment");
    hipStream_t st = static_cast<hipStream_t>(stream);
    p.pre_chunks = d->norm && d->stats_chunks > 0 ? d->stats_chunks : 0;
    p.shift = d->stats_shift;
    if (d->norm) {
        NG_REQUIRE(ng_aligned16(d->stats_shift), "instnorm_fwd: stats_shift must be 16-byte aligned");
        NG_REQUIRE(d->mean && d->rstd && d->ws && d->ws_elems >= int64_t(d->B) * (p.pre_chunks > 0 ? int64_t(p.pre_chunks) * 4 : int64_t(p.nchunk) * 2) * d->C, "instnorm_fwd: mean/rstd/ws missing or too small");
        if (p.pre_chunks == 0) hipLaunchKernelGGL(in_stats_kernel, dim3(p.nchunk, d->B), dim3(256), 0, st, p);
        hipLaunchKernelGGL(in_finalize_kernel, dim3(d->B, (d->C >= 64 && d->C % 64 == 0) ? d->C / 64 : 1), dim3(256), 0, st, p, d->B);
    }
    if (!stats_only) {
        const dim3 grid(p.nchunk, d->B);
        const bool reflect = d->border == NIRGAN_BORDER_REFLECT && d->o_pad > 0;
        const long long big = (long long)p.o_img > (long long)p.r_img ? p.o_img : p.r_img;
        if (in_fast_ok(d->H, d->W, d->C, reflect ? d->o_pad : 0, big > (long long)p.HW * p.C ? big : (long long)p.HW * p.C)) {
            const int out = d->out && d->out_bf16 ? 1 : (d->out ? 0 : 2);
#define NG_APPLY(Y16, OUT, REFLECT, RESID) hipLaunchKernelGGL((in_apply_fast_kernel<Y16, OUT, REFLECT, RESID>), grid, dim3(256), 0, st, p)
#define NG_APPLY_R(Y16, OUT, REFLECT) do { if (d->residual) NG_APPLY(Y16, OUT, REFLECT, true); else NG_APPLY(Y16, OUT, REFLECT, false); } while (0)
#define NG_APPLY_B(Y16, OUT) do { if (reflect) NG_APPLY_R(Y16, OUT, true); else NG_APPLY_R(Y16, OUT, false); } while (0)
#define NG_APPLY_O(Y16) do { if (out == 0) NG_APPLY_B(Y16, 0); else if (out == 1) NG_APPLY_B(Y16, 1); else NG_APPLY_B(Y16, 2); } while (0)
            if (p.y16) NG_APPLY_O(true); else NG_APPLY_O(false);
#undef NG_APPLY_O
#undef NG_APPLY_B
#undef NG_APPLY_R
#undef NG_APPLY
        } else {
            hipLaunchKernelGGL(in_apply_kernel, grid, dim3(256), 0, st, p);
        }
    }
    return nirgan_check_launch("instnorm_fwd");
}

extern "C" int nirgan_instnorm_bwd(const nirgan_in_bwd_desc* d, void* stream) {
    NG_REQUIRE(d && (d->dy || d->norm) && (d->g || d->g2), "instnorm_bwd: null pointer");
    // with norm, neither dy nor dy_bf16: the two reductions only, the consumer evaluates dy on the fly (nirgan_wino6_input_dy_norm); dy_bf16 alone:
    // every consumer reads the bf16 twin, the fp32 tensor is not stored
    const bool sums_only = d->dy == nullptr && d->dy_bf16 == nullptr;
    NG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C >= 4 && d->C % 4 == 0 && d->C <= 1024, "instnorm_bwd: bad shape");
    NG_REQUIRE(!d->g || (d->g_hp == d->H + 2 * d->g_pad && d->g_wp == d->W + 2 * d->g_pad), "instnorm_bwd: g geometry mismatch");
    NG_REQUIRE(!d->g_fold || (d->g_pad < d->H && d->g_pad < d->W), "instnorm_bwd: fold halo wider than the image");
    NG_REQUIRE(sums_only || (d->d_hp == d->H + 2 * d->d_pad && d->d_wp == d->W + 2 * d->d_pad), "instnorm_bwd: dy geometry mismatch");
    const bool masked = d->act == NIRGAN_ACT_RELU || d->act == NIRGAN_ACT_LRELU;
    NG_REQUIRE(!(masked || d->norm) || d->y, "instnorm_bwd: y (pre-activation input of the block) required for the mask / statistics");
    NG_REQUIRE(!d->norm || (d->mean && d->rstd && d->ws), "instnorm_bwd: mean/rstd/ws required when norm");
    InBwd p = in_bwd_params(d);
    NG_REQUIRE(!d->dy_bf16 || (d->C % 8 == 0 && (reinterpret_cast<uintptr_t>(d->dy_bf16) & 15) == 0), "instnorm_bwd: bf16 twin needs C %% 8 == 0 and 16-byte alignment");
    NG_REQUIRE(!d->norm || d->ws_elems >= int64_t(d->B) * p.pchunks * 2 * d->C + int64_t(d->B) * 2 * d->C, "instnorm_bwd: ws too small");
    // sums_chunks > 0: the producer of the gradient already left the partial sums of the first pass in ws -- nirgan_wino6_output in its
    // fused mode (the folded gradient then sits in gsum_out) or a convolution launch with nirgan_conv_desc.fuse_* (the gradient is g
    // itself: no fold, no second gradient)
    const bool pre = d->norm && d->sums_chunks > 0;
    NG_REQUIRE(!pre || d->gsum_out != nullptr || (d->g != nullptr && !d->g_fold && d->g2 == nullptr), "instnorm_bwd: sums_chunks needs the folded gradient in gsum_out, or a plain g (no fold, no g2)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    NG_REQUIRE(d->norm || !d->dbias || (d->ws && d->ws_elems >= int64_t(d->B) * p.nchunk * d->C),
               "instnorm_bwd: dbias needs ws (one row of channel sums per block, summed in fixed order)");
    // the branch-light forms (with norm and a halo'd gradient g): see in_apply_fast_kernel
    const long long gbig = (long long)p.g_img > (long long)p.d_img ? p.g_img : p.d_img;
    const bool fast = d->norm && d->g != nullptr && in_fast_ok(d->H, d->W, d->C, d->g_fold ? d->g_pad : 0, gbig > (long long)p.HW * p.C ? gbig : (long long)p.HW * p.C);
    const dim3 bgrid(p.nchunk, d->B);
#define NG_P1(Y16, G16, FOLD) hipLaunchKernelGGL((in_bwd_pass1_fast_kernel<Y16, G16, FOLD>), bgrid, dim3(256), 0, st, p)
#define NG_P1_F(Y16, G16) do { if (d->g_fold) NG_P1(Y16, G16, true); else NG_P1(Y16, G16, false); } while (0)
#define NG_P1_G(Y16) do { if (p.g16) NG_P1_F(Y16, true); else NG_P1_F(Y16, false); } while (0)
    if (!pre) {
        if (fast) { if (p.y16) NG_P1_G(true); else NG_P1_G(false); }
        else hipLaunchKernelGGL(in_bwd_pass1_kernel, bgrid, dim3(256), 0, st, p);
    }
#undef NG_P1_G
#undef NG_P1_F
#undef NG_P1
    if (!d->norm && d->dbias) return nirgan_colsum(d->ws, int64_t(d->B) * p.nchunk, d->C, d->dbias, 1, stream);
    if (d->norm) {
        hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(d->B, (d->C >= 64 && d->C % 64 == 0) ? d->C / 64 : 1), dim3(256), 0, st, p, d->B);
        if (!sums_only) {
            // (pass 2 may take the folded gradient from gsum_out instead of g: the fast form needs one of the two and 31-bit offsets)
            const bool fast2 = (d->g != nullptr ? fast : (d->gsum_out != nullptr && in_fast_ok(d->H, d->W, d->C, 0, (long long)p.d_img > (long long)p.HW * p.C ? p.d_img : (long long)p.HW * p.C)));
            const int dyo = d->dy && d->dy_bf16 ? 1 : (d->dy ? 0 : 2);
            const int B_ = d->B;
#define NG_P2(Y16, G16, FOLD, DY) hipLaunchKernelGGL((in_bwd_pass2_fast_kernel<Y16, G16, FOLD, DY>), bgrid, dim3(256), 0, st, p, B_)
#define NG_P2_D(Y16, G16, FOLD) do { if (dyo == 0) NG_P2(Y16, G16, FOLD, 0); else if (dyo == 1) NG_P2(Y16, G16, FOLD, 1); else NG_P2(Y16, G16, FOLD, 2); } while (0)
#define NG_P2_F(Y16, G16) do { if (d->g_fold && d->g) NG_P2_D(Y16, G16, true); else NG_P2_D(Y16, G16, false); } while (0)
#define NG_P2_G(Y16) do { if (p.g16) NG_P2_F(Y16, true); else NG_P2_F(Y16, false); } while (0)
            if (fast2) { if (p.y16) NG_P2_G(true); else NG_P2_G(false); }
            else hipLaunchKernelGGL(in_bwd_pass2_kernel, bgrid, dim3(256), 0, st, p, d->B);
#undef NG_P2_G
#undef NG_P2_F
#undef NG_P2_D
#undef NG_P2
        }
    }
    return nirgan_check_launch("instnorm_bwd");
}
