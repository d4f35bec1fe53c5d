// Diagnostic build of the 128-row convolution tile with in-kernel stamps (never shipped) on an N = 64 layer: one sub-pixel phase of a
// stride-2 data gradient (16 x 128 x 128 pixels, 2 x 2 taps of 128 channels, 64 output channels).  -DNGD_BF16: both operands stored as
// bf16 (conv_tile<64, 1, true, true>), else exact fp32.  Where does a wave spend its cycles: loop / epilogue / waits?
#define NG_DIAG 1
#include "../../nir-gan_amd/csrc/igemm_tiles.h"
#include <vector>
#include <algorithm>
#include <cstring>

void nirgan_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

#ifndef NGD_BN
#define NGD_BN 64
#endif
__global__ __launch_bounds__(256, 2) void k(const ng::ConvParams p) {
    __shared__ __attribute__((aligned(16))) char st0[(128 + NGD_BN) * 128];
    __shared__ __attribute__((aligned(16))) char st1[(128 + NGD_BN) * 128];
#ifdef NGD_BF16
    ng::conv_tile<NGD_BN, 1, true, true>(p, blockIdx.x, st0, st1);
#else
    ng::conv_tile<NGD_BN, 0>(p, blockIdx.x, st0, st1);
#endif
}

int main() {
    const int B = 16, H = 128, C = 128, N = NGD_BN, T = 2;
#ifdef NGD_BF16
    const int es = 2;
#else
    const int es = 4;
#endif
    const size_t in_n = size_t(B) * (H + 1) * (H + 1) * C, out_n = size_t(B) * H * H * N, w_n = size_t(N) * T * T * C;
    char *in, *w; float* out; float* zero; unsigned long long* dbg;
    hipMalloc(&in, in_n * es); hipMalloc(&w, w_n * es); hipMalloc(&out, out_n * 4); hipMalloc(&zero, 256);
    hipMemset(in, 0x3c, in_n * es); hipMemset(w, 0x3c, w_n * es); hipMemset(zero, 0, 256);
    nirgan_conv_desc d = {};
    d.in = reinterpret_cast<float*>(in); d.in_elems = in_n; d.in_hp = H + 1; d.in_wp = H + 1; d.in_cs = C; d.run = C; d.in_stride = 1; d.ntaps = T * T;
    for (int t = 0; t < T * T; ++t) { d.tap_dh[t] = t / T; d.tap_dw[t] = t % T; }
    d.w = reinterpret_cast<float*>(w); d.w_elems = w_n; d.out = out; d.out_elems = out_n; d.out_hp = H; d.out_wp = H; d.out_cs = N; d.out_stride = 1;
    d.B = B; d.OH = H; d.OW = H; d.N = N; d.zero_page = zero;
#ifdef NGD_BF16
    d.precision = 1; d.w_bf16 = 1; d.in_bf16 = 1;
#ifdef NGD_OUT16
    d.out_bf16 = 1;                 // the output stored as bf16 (a y in front of an instance norm, a data gradient): same buffer, half used
#endif
#endif
    d.algo = NIRGAN_CONV_TILE128;
    ng::ConvParams p;
    if (ng::build_conv_params(&d, p) != 0) return 1;
    const int nb = p.mtiles * p.ntiles;
    hipMalloc(&dbg, size_t(nb) * 4 * 6 * 8); hipMemset(dbg, 0, size_t(nb) * 4 * 6 * 8);
    p.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, p);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, p);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * B * H * H * N * T * T * C;
    printf("N=%d %s: %d blocks, %.1f us per launch (stamped build), %.1f TF/s\n", N, es == 2 ? "bf16" : "fp32", nb, ms * 100, flops / (ms * 1e-4) / 1e12);
    std::vector<unsigned long long> r(size_t(nb) * 24);
    hipMemcpy(r.data(), dbg, r.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nb * 4; ++i) { tmin = std::min(tmin, r[i * 6]); tmax = std::max(tmax, r[i * 6 + 2]); }
    printf("kernel span %.1f us (100 MHz stamps)\n", double(tmax - tmin) / 100.0);
    double s[5] = {};
    for (int i = 0; i < nb * 4; ++i) {
        const unsigned long long* o = &r[i * 6];
        s[0] += double(o[1] - o[0]); s[1] += double(o[2] - o[1]); s[2] += double(o[3]); s[3] += double(o[4]); s[4] += double(o[5]);
    }
    const double n = nb * 4.0;
    printf("per wave, cycles: set-up %.0f  K loop %.0f (waits %.0f, bodies %.0f)  epilogue %.0f\n", s[4] / n, s[0] / n, s[2] / n, s[3] / n, s[1] / n);
    for (int b : {0, 1, nb / 2}) {
        const unsigned long long* o = &r[size_t(b) * 24];
        printf("block %d wave0: start %llu loop %llu epi %llu wait %llu body %llu\n", b, o[0] - tmin, o[1] - o[0], o[2] - o[1], o[3], o[4]);
    }
    return 0;
}
