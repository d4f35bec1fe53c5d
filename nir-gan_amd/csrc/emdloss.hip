// emd_loss of the reference's utils/losses.py:64-78, value and gradient wrt the prediction:
//     p = softmax(pred.reshape(B, -1), 1), t = softmax(target.reshape(B, -1), 1)
//     emd = mean | cumsum(p, 1) - cumsum(t, 1) |                      (over all B * N elements)
// One workgroup of 1024 threads per sample; a thread owns a contiguous segment of the N = C*H*W elements.  Scans and sums
// run in double (torch's CPU cumsum accumulates float in double too; a float running sum of 65 536 terms would carry an
// error of the size of the CDF differences themselves).  Backward, with s_i = sign(c_i - d_i) / (B N):
//     dL/dp_j = sum_{i >= j} s_i =: g_j        (suffix sum)          dL/dx_k = p_k (g_k - sum_j p_j g_j)
// Integer / scan work, HBM-bound and tiny (16 x 65 536 floats): launch-latency sized.
#include "common.h"

namespace {

constexpr int NT = 1024;

struct EmdP {
    const float* x; const float* y;
    int B; long long N;
    float weight;
    double* partial;        // [B] per-sample sums of |c - d|
    float* grad;            // [B][N] or nullptr: += weight * dL/dx
    float* scratch;         // [B][N] floats (suffix sums g_j, unscaled), used when grad != nullptr
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum of a double (all threads get the result); red: 16 doubles
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max_f(float v, float* red) {
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) s = fmaxf(s, red[i]);
    return s;
}

// exclusive prefix (or suffix) of one double per thread over the block, in thread order; buf: NT doubles
__device__ __forceinline__ double block_exclusive(double v, double* buf, bool suffix) {
    __syncthreads();
    buf[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 64) {                       // wave 0: lane l owns threads 16 l .. 16 l + 15
        const int l = threadIdx.x;
        double loc[16], s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { loc[i] = buf[l * 16 + i]; s += loc[i]; }
        // exclusive scan of the 64 lane totals
        double inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double n = __shfl_up(inc, o, 64);
            if (l >= o) inc += n;
        }
        const double total = __shfl(inc, 63, 64);
        double run = suffix ? total - inc : inc - s;          // sum of the lanes before (after) this one
        if (!suffix) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { buf[l * 16 + i] = run; run += loc[i]; }
        } else {
#pragma unroll
            for (int i = 15; i >= 0; --i) { buf[l * 16 + i] = run; run += loc[i]; }
        }
    }
    __syncthreads();
    return buf[threadIdx.x];
}

__global__ __launch_bounds__(NT) void emd_kernel(const EmdP p) {
    __shared__ double dbuf[NT];
    __shared__ double dred[NT / 64];
    __shared__ float fred[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = p.x + size_t(b) * p.N;
    const float* y = p.y + size_t(b) * p.N;
    const long long seg = (p.N + NT - 1) / NT;
    const long long lo = tid * seg < p.N ? tid * seg : p.N, hi = lo + seg < p.N ? lo + seg : p.N;
    // softmax statistics
    float mx = -INFINITY, my = -INFINITY;
    for (long long i = tid; i < p.N; i += NT) { mx = fmaxf(mx, x[i]); my = fmaxf(my, y[i]); }
    mx = block_max_f(mx, fred);
    my = block_max_f(my, fred);
    double sx = 0.0, sy = 0.0;
    for (long long i = tid; i < p.N; i += NT) { sx += double(expf(x[i] - mx)); sy += double(expf(y[i] - my)); }
    sx = block_sum_d(sx, dred);
    sy = block_sum_d(sy, dred);
    const double ix = 1.0 / sx, iy = 1.0 / sy;
    // CDFs: per-thread segment sums, block scan, then the walk
    double ax = 0.0, ay = 0.0;
    for (long long i = lo; i < hi; ++i) { ax += double(expf(x[i] - mx)) * ix; ay += double(expf(y[i] - my)) * iy; }
    double cx = block_exclusive(ax, dbuf, false);
    double cy = block_exclusive(ay, dbuf, false);
    double acc = 0.0, nsign = 0.0;
    float* gs = p.grad ? p.scratch + size_t(b) * p.N : nullptr;
    for (long long i = lo; i < hi; ++i) {
        cx += double(expf(x[i] - mx)) * ix;
        cy += double(expf(y[i] - my)) * iy;
        // the reference compares float CDFs: torch.cumsum returns float32
        const float d = float(cx) - float(cy);
        acc += double(fabsf(d));
        const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        if (gs) { gs[i] = s; nsign += double(s); }
    }
    acc = block_sum_d(acc, dred);
    if (tid == 0) p.partial[b] = acc;
    if (!p.grad) return;
    // suffix sums of the signs: g_j = sum_{i >= j} s_i (unscaled), kept in scratch; dot = sum_j p_j g_j
    double run = block_exclusive(nsign, dbuf, true);
    double dot = 0.0;
    for (long long i = hi - 1; i >= lo; --i) {
        run += double(gs[i]);
        gs[i] = float(run);                                   // integers up to N: exact in float for N < 2^24
        dot += double(expf(x[i] - mx)) * ix * run;
    }
    dot = block_sum_d(dot, dred);
    const double scale = double(p.weight) / (double(p.B) * double(p.N));
    float* g = p.grad + size_t(b) * p.N;
    __syncthreads();
    for (long long i = tid; i < p.N; i += NT) {
        const double pk = double(expf(x[i] - mx)) * ix;
        g[i] += float(scale * pk * (double(gs[i]) - dot));
    }
}

__global__ void emd_value_kernel(const double* partial, int B, long long N, float weight, float* loss, float* value) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += partial[b];
    const float v = float(s / (double(B) * double(N)));
    if (value) *value = v;
    if (loss) atomicAdd(loss, weight * v);
}

}  // namespace

extern "C" int64_t nirgan_emd_loss_ws_bytes(int B, int64_t N, int with_grad) {
    if (B <= 0 || N <= 0) return 0;
    return int64_t(B) * 8 + (with_grad ? int64_t(B) * N * 4 : 0);
}

extern "C" int nirgan_emd_loss(const nirgan_emd_loss_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr && d->pred && d->target && d->ws, "emd_loss: null pointer");
    NG_REQUIRE(d->B > 0 && d->B <= 65535 && d->N > 0 && d->N < (1ll << 24), "emd_loss: B=%d N=%lld out of range (N < 2^24)", d->B, (long long)d->N);
    NG_REQUIRE(d->ws_bytes >= nirgan_emd_loss_ws_bytes(d->B, d->N, d->grad_pred != nullptr), "emd_loss: workspace too small (nirgan_emd_loss_ws_bytes)");
    NG_REQUIRE((reinterpret_cast<uintptr_t>(d->ws) & 7) == 0, "emd_loss: workspace must be 8-byte aligned");
    EmdP p;
    p.x = d->pred; p.y = d->target; p.B = d->B; p.N = d->N; p.weight = d->weight;
    p.partial = static_cast<double*>(d->ws);
    p.scratch = reinterpret_cast<float*>(p.partial + d->B);
    p.grad = d->grad_pred;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(emd_kernel, dim3(d->B), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(emd_value_kernel, dim3(1), dim3(64), 0, st, p.partial, d->B, (long long)d->N, d->weight, d->loss, d->value);
    return nirgan_check_launch("emd_loss");
}
