// Diagnostic (never shipped): what a pure streaming pass over a bf16 tensor reaches on the MI355X as a function of bytes per lane and of the
// independent accesses a thread keeps in flight -- the shape of the bf16 mode's instance-norm apply pass on a residual-trunk layer
// (16 x 64 x 64 x 256 bf16 in, the same out; 33.5 MB each way).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/diag/stream_bf16 scripts/diag/stream_bf16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));

template <int V, int ILP>
__global__ __launch_bounds__(256) void k(const __bf16* __restrict__ x, __bf16* __restrict__ y, long n, float a, float b) {
    typedef __bf16 bv __attribute__((ext_vector_type(V)));
    typedef float fv __attribute__((ext_vector_type(V)));
    const long stride = (long)gridDim.x * 256 * V;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    for (; i + (ILP - 1) * stride < n; i += ILP * stride) {
        bv v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) v[j] = *reinterpret_cast<const bv*>(x + i + j * stride);
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            fv f = __builtin_convertvector(v[j], fv) * a + b;
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] = f[e] > 0.f ? f[e] : 0.f;
            *reinterpret_cast<bv*>(y + i + j * stride) = __builtin_convertvector(f, bv);
        }
    }
    for (; i < n; i += stride) {
        fv f = __builtin_convertvector(*reinterpret_cast<const bv*>(x + i), fv) * a + b;
        *reinterpret_cast<bv*>(y + i) = __builtin_convertvector(f, bv);
    }
}

template <int V, int ILP> void run(const __bf16* x, __bf16* y, long n, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<V, ILP>), dim3(blocks), dim3(256), 0, 0, x, y, n, 1.5f, 0.1f);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((k<V, ILP>), dim3(blocks), dim3(256), 0, 0, x, y, n, 1.5f, 0.1f);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %2d B/lane  ILP %d  %5d blocks: %6.1f us  %5.2f TB/s\n", V * 2, ILP, blocks, ms * 1e3 / 20, 4.0 * n / (ms / 20 * 1e-3) / 1e12);
}

int main() {
    const long n = 16L * 64 * 64 * 256;
    __bf16 *x, *y; hipMalloc(&x, n * 2 * 8); hipMalloc(&y, n * 2 * 8);
    hipMemset(x, 0x3c, n * 2 * 8);
    // eight different tensors in rotation would defeat the 256 MB Infinity Cache; one pair (67 MB) stays resident as the step's does not:
    // so offset each launch?  Keep it simple: report the cache-resident figure and a 2 x 268 MB one
    printf("33.5 MB in + 33.5 MB out (Infinity-Cache resident between launches):\n");
    for (int blocks : {1024, 2048, 4096, 8192}) { run<4, 1>(x, y, n, blocks); run<4, 4>(x, y, n, blocks); run<8, 1>(x, y, n, blocks); run<8, 2>(x, y, n, blocks); run<8, 4>(x, y, n, blocks); }
    printf("268 MB in + 268 MB out (beyond the Infinity Cache):\n");
    for (int blocks : {2048, 8192}) { run<4, 1>(x, y, n * 8, blocks); run<4, 4>(x, y, n * 8, blocks); run<8, 1>(x, y, n * 8, blocks); run<8, 4>(x, y, n * 8, blocks); }
    return 0;
}
