// Implicit-GEMM weight gradient on the gfx950 matrix pipe, deterministic split over pixels.
//
//   slab[s][n][J] = sum_{m in split s} P[pix(m)][n] * Q[pix(m)*q_stride + tap(J)][c(J)],  J = t*run + c
//   GEMM view: rows n (TN = 128 or 64 per block), columns J (128 per block), reduction over
//   pixels m in steps of 32.  4 waves as 2x2; fragments come straight from the [m][n] / [m][J]
//   LDS images: for v_mfma_f32_32x32x2_f32 the two k lanes-halves read pixel rows 2kk and
//   2kk+1, and one ds_read_b64 hands a lane two adjacent n (or J): MFMA tile e covers rows
//   n = base + 2*i + e (a stride-2 interleave, undone at the store).  Both images are filled
//   by LDS-DMA, rows are >= 256 B so no swizzle is needed (a half-wave reads 256 contiguous
//   bytes).  Two LDS stages, one barrier per 32-pixel step.
//   The split partial sums go to slabs (plain stores) and are summed by reduce_rows in a fixed
//   order, so gradients are bitwise reproducible run to run.
#include "igemm_tiles.h"

namespace {

template <int TN>
__global__ __launch_bounds__(256, 2) void wgrad_igemm_kernel(const ng::WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ng::wgrad_tile<TN>(p, blockIdx.x, smem);
}

// horizontally fused launch: the data-gradient tiles of a stride-1 convolution followed by the tiles of its
// weight gradient (both consume the same dY).  One grid: the weight-gradient blocks fill the partly empty
// last round of the data-gradient, and the other way round.
__global__ __launch_bounds__(256, 2) void conv_wgrad_pair_kernel(const ng::ConvParams cp, const ng::WgradParams wp, const int conv_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (int(blockIdx.x) < conv_blocks)
        ng::conv_tile<128>(cp, blockIdx.x, smem);
    else
        ng::wgrad_tile<128>(wp, int(blockIdx.x) - conv_blocks, smem);
}

__global__ void reduce_rows_kernel(const float* __restrict__ slabs, int nsplit, int N, int K,
                                   const int32_t* __restrict__ map, float* __restrict__ dst,
                                   int64_t dst_elems, int dst_row_stride, int accumulate) {
    const int64_t total = int64_t(N) * K;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int n = int(i / K), k = int(i - int64_t(n) * K);
        const int mk = map[k];
        if (mk < 0) continue;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += slabs[int64_t(sp) * total + i];
        const int64_t o = int64_t(n) * dst_row_stride + mk;
        if (o < dst_elems) dst[o] = accumulate ? dst[o] + s : s;
    }
}

__global__ void pack_rows_kernel(const float* __restrict__ src, int64_t src_elems, int src_row_stride,
                                 const int32_t* __restrict__ map, float* __restrict__ dst, int N, int K) {
    const int64_t total = int64_t(N) * K;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int n = int(i / K), k = int(i - int64_t(n) * K);
        const int mk = map[k];
        float v = 0.f;
        if (mk >= 0) {
            const int64_t o = int64_t(n) * src_row_stride + mk;
            if (o < src_elems) v = src[o];
        }
        dst[i] = v;
    }
}

}  // namespace


extern "C" int nirgan_wgrad_igemm(const nirgan_wgrad_desc* d, void* stream) {
    ng::WgradParams p;
    const int rc = ng::build_wgrad_params(d, p);
    if (rc != NIRGAN_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->N > 64) {
        static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_igemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess; }();
        (void)once;
        hipLaunchKernelGGL(wgrad_igemm_kernel<128>, dim3(p.ntiles_n * p.ntiles_k * p.nsplit), dim3(256), 65536, st, p);
    } else {
        hipLaunchKernelGGL(wgrad_igemm_kernel<64>, dim3(p.ntiles_k * p.nsplit), dim3(256), 49152, st, p);
    }
    return nirgan_check_launch("wgrad_igemm");
}

extern "C" int nirgan_conv_wgrad_pair(const nirgan_conv_desc* c, const nirgan_wgrad_desc* w, void* stream) {
    ng::ConvParams cp;
    ng::WgradParams wp;
    int rc = ng::build_conv_params(c, cp);
    if (rc != NIRGAN_OK) return rc;
    rc = ng::build_wgrad_params(w, wp);
    if (rc != NIRGAN_OK) return rc;
    if (c->N <= 64 || w->N <= 64) {      // narrow variants: two ordinary launches
        rc = nirgan_conv_igemm(c, stream);
        return rc != NIRGAN_OK ? rc : nirgan_wgrad_igemm(w, stream);
    }
    static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess; }();
    (void)once;
    const int conv_blocks = cp.mtiles * cp.ntiles;
    const int wgrad_blocks = wp.ntiles_n * wp.ntiles_k * wp.nsplit;
    hipLaunchKernelGGL(conv_wgrad_pair_kernel, dim3(conv_blocks + wgrad_blocks), dim3(256), 65536, static_cast<hipStream_t>(stream), cp, wp, conv_blocks);
    return nirgan_check_launch("conv_wgrad_pair");
}

extern "C" int nirgan_reduce_rows(const float* slabs, int nsplit, int N, int K, const int32_t* map,
                                  float* dst, int64_t dst_elems, int dst_row_stride, int accumulate, void* stream) {
    NG_REQUIRE(slabs && map && dst && nsplit >= 1 && N > 0 && K > 0, "reduce_rows: bad arguments");
    const int64_t total = int64_t(N) * K;
    const int grid = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slabs, nsplit, N, K, map, dst, dst_elems, dst_row_stride, accumulate);
    return nirgan_check_launch("reduce_rows");
}

extern "C" int nirgan_pack_rows(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                                float* dst, int N, int K, void* stream) {
    NG_REQUIRE(src && map && dst && N > 0 && K > 0, "pack_rows: bad arguments");
    const int64_t total = int64_t(N) * K;
    const int grid = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, src_elems, src_row_stride, map, dst, N, K);
    return nirgan_check_launch("pack_rows");
}
