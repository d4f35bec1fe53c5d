// Histogram matching of predicted NIR tiles to a reference band: create_synthetic_dataset.py:34-47
// calls skimage.exposure.match_histograms(img, ref, channel_axis=None) per tile on the CPU.  Published algorithm
// (skimage/exposure/histogram_matching.py::_match_cumulative_cdf, float images):
//     src_values, src_lookup, src_counts = unique(source);  tmpl_values, tmpl_counts = unique(template)
//     out = interp(cumsum(src_counts)/n, cumsum(tmpl_counts)/n, tmpl_values)[src_lookup]
// Device restatement: both planes are sorted (bitonic network, 2048-element phases in LDS, wider strides as
// global compare-exchange passes; the source carries its pixel index), then every sorted source element finds
// its cumulative count r by an upper-bound search in its own plane and evaluates np.interp at r/n against the
// sorted template, whose "unique" runs are located with two more binary searches.  Integer/byte work + HBM:
// nothing here is a contraction.
#include "common.h"

namespace {

constexpr int CHUNK = 2048;          // elements sorted per workgroup in LDS (256 threads x 8)

__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- fill: keys (and pixel indices) of one plane, padded with the largest key up to the power of two P
template <typename T>
__global__ __launch_bounds__(256) void hm_fill_kernel(const float* __restrict__ src, int N, int P, T* __restrict__ dst) {
    const int plane = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
        const uint32_t key = i < N ? f2key(src[size_t(plane) * N + i]) : 0xffffffffu;
        if constexpr (sizeof(T) == 8) dst[size_t(plane) * P + i] = (uint64_t(key) << 32) | uint32_t(i);
        else dst[size_t(plane) * P + i] = key;
    }
}

template <typename T>
__device__ __forceinline__ void cmpx(T& a, T& b, bool ascending) {
    if ((a > b) == ascending) { const T t = a; a = b; b = t; }
}

// ---- all stages with stride < CHUNK of the merges k = kfirst .. klast (k doubles), inside LDS
template <typename T>
__global__ __launch_bounds__(256) void hm_local_kernel(T* __restrict__ data, int P, int kfirst, int klast) {
    __shared__ T s[CHUNK];
    const int plane = blockIdx.y;
    T* base = data + size_t(plane) * P + size_t(blockIdx.x) * CHUNK;
    for (int i = threadIdx.x; i < CHUNK; i += 256) s[i] = base[i];
    __syncthreads();
    const int g0 = blockIdx.x * CHUNK;
    for (int k = kfirst; k <= klast; k <<= 1) {
        for (int j = (k >> 1) < CHUNK ? (k >> 1) : (CHUNK >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < CHUNK / 2; t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // element with bit j clear
                const bool asc = ((g0 + i) & k) == 0;
                cmpx(s[i], s[i + j], asc);
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < CHUNK; i += 256) base[i] = s[i];
}

// ---- one compare-exchange pass of merge k at stride j >= CHUNK
template <typename T>
__global__ __launch_bounds__(256) void hm_global_kernel(T* __restrict__ data, int P, int k, int j) {
    const int plane = blockIdx.y;
    T* base = data + size_t(plane) * P;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < P / 2; t += gridDim.x * 256) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        T a = base[i], b = base[i + j];
        const bool asc = (i & k) == 0;
        if ((a > b) == asc) { base[i] = b; base[i + j] = a; }
    }
}

// first index in [0, n) whose value is > v (values by float comparison, so -0.0 == +0.0 as numpy's unique has it)
__device__ __forceinline__ int upper_bound_pairs(const uint64_t* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key2f(uint32_t(a[mid] >> 32)) <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int upper_bound_keys(const uint32_t* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key2f(a[mid]) <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int lower_bound_keys(const uint32_t* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key2f(a[mid]) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void hm_match_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ tmpl,
                                                       int N, int P, float* __restrict__ out) {
    const int plane = blockIdx.y;
    const uint64_t* s = src + size_t(plane) * P;
    const uint32_t* t = tmpl + size_t(plane) * P;
    float* o = out + size_t(plane) * N;
    const double inv_n = 1.0 / double(N);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < N; p += gridDim.x * 256) {
        const uint64_t e = s[p];
        const float v = key2f(uint32_t(e >> 32));
        const int idx = int(uint32_t(e));
        const int r = upper_bound_pairs(s, N, v);            // cumulative count of the source value: 1..N
        const float tv = key2f(t[r - 1]);
        const int hi_end = upper_bound_keys(t, N, tv);       // cumulative count of the template value tv
        double res;
        if (hi_end == r) {
            res = double(tv);                                // x == xp[i]
        } else {
            const int lo_start = lower_bound_keys(t, N, tv); // cumulative count of the previous unique template value
            if (lo_start == 0) {
                res = double(tv);                            // x < xp[0]: np.interp clamps to fp[0]
            } else {
                const double prev = double(key2f(t[lo_start - 1]));
                const double x = double(r) * inv_n, xp0 = double(lo_start) * inv_n, xp1 = double(hi_end) * inv_n;
                const double slope = (double(tv) - prev) / (xp1 - xp0);
                res = slope * (x - xp0) + prev;
            }
        }
        o[idx] = float(res);
    }
}

int next_pow2(int n) {
    int p = CHUNK;
    while (p < n) p <<= 1;
    return p;
}

template <typename T>
void sort_planes(T* data, int B, int P, hipStream_t st) {
    const dim3 lgrid(P / CHUNK, B), ggrid(P / 2 / 256 < 1024 ? P / 2 / 256 : 1024, B);
    hipLaunchKernelGGL((hm_local_kernel<T>), lgrid, dim3(256), 0, st, data, P, 2, CHUNK);
    for (int k = 2 * CHUNK; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= CHUNK; j >>= 1) hipLaunchKernelGGL((hm_global_kernel<T>), ggrid, dim3(256), 0, st, data, P, k, j);
        hipLaunchKernelGGL((hm_local_kernel<T>), lgrid, dim3(256), 0, st, data, P, k, k);
    }
}

}  // namespace

extern "C" int64_t nirgan_hist_match_ws_bytes(int B, int N) {
    if (B <= 0 || N <= 0 || N > (1 << 26)) return 0;
    return int64_t(B) * next_pow2(N) * 12;
}

extern "C" int nirgan_hist_match(const nirgan_hist_match_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr && d->image && d->reference && d->out && d->ws, "hist_match: null pointer");
    NG_REQUIRE(d->B > 0 && d->B <= 65535 && d->N > 0 && d->N <= (1 << 26), "hist_match: B=%d N=%d out of range", d->B, d->N);
    const int P = next_pow2(d->N);
    NG_REQUIRE(d->ws_bytes >= int64_t(d->B) * P * 12, "hist_match: workspace too small (nirgan_hist_match_ws_bytes)");
    NG_REQUIRE((reinterpret_cast<uintptr_t>(d->ws) & 7) == 0, "hist_match: workspace must be 8-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t* src = static_cast<uint64_t*>(d->ws);
    uint32_t* tmpl = reinterpret_cast<uint32_t*>(src + size_t(d->B) * P);
    const dim3 fgrid(P / 256 < 1024 ? P / 256 : 1024, d->B);
    hipLaunchKernelGGL((hm_fill_kernel<uint64_t>), fgrid, dim3(256), 0, st, d->image, d->N, P, src);
    hipLaunchKernelGGL((hm_fill_kernel<uint32_t>), fgrid, dim3(256), 0, st, d->reference, d->N, P, tmpl);
    sort_planes<uint64_t>(src, d->B, P, st);
    sort_planes<uint32_t>(tmpl, d->B, P, st);
    const dim3 mgrid((d->N + 255) / 256 < 2048 ? (d->N + 255) / 256 : 2048, d->B);
    hipLaunchKernelGGL(hm_match_kernel, mgrid, dim3(256), 0, st, src, tmpl, d->N, P, d->out);
    return nirgan_check_launch("hist_match");
}
