// Diagnostic build of the bf16 256 x 256 x 64 tile with in-kernel stamps (never shipped): where does a wave of the residual-block
// 3x3 256 -> 256 convolution (bs 16, 64 x 64, both operands bf16) spend its cycles -- prologue (address set-up, first LDS-DMA burst),
// K loop, statistics, epilogue -- and what clock does the chip hold (s_memtime against s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/diag/tile256_stamp scripts/diag/tile256_stamp.hip
#define NG_DIAG256 1
#ifndef RING
#define RING 8
#endif
#include "../../nir-gan_amd/csrc/igemm_tile256.h"
#include <vector>
#include <cstring>
#include <algorithm>

void nirgan_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

__global__ __launch_bounds__(512, 2) void k(const ng::ConvParams p) {
    __shared__ __attribute__((aligned(16))) char lds[RING * ng::T256_HALF];
    ng::conv_tile256<false, RING>(p, ng_xcd_remap(blockIdx.x, gridDim.x), lds);
}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int B = 16, H = 64, C = 256;
    const int out16 = argc > 1 ? atoi(argv[1]) : 1;
    const size_t in_n = size_t(B) * (H + 2) * (H + 2) * C, out_n = size_t(B) * H * H * C, w_n = size_t(C) * 9 * C;
    unsigned short *in, *w; float *out, *zero; unsigned long long* dbg;
    hipMalloc(&in, in_n * 2); hipMalloc(&w, w_n * 2); hipMalloc(&out, out_n * 4); hipMalloc(&zero, 256);
    std::vector<unsigned short> h(in_n); for (size_t i = 0; i < in_n; ++i) h[i] = bf16_of(float((i * 2654435761u) % 1000) / 500.f - 1.f);
    hipMemcpy(in, h.data(), in_n * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw(w_n); for (size_t i = 0; i < w_n; ++i) hw[i] = bf16_of(float((i * 40503u) % 1000) / 25000.f - 0.02f);
    hipMemcpy(w, hw.data(), w_n * 2, hipMemcpyHostToDevice); hipMemset(zero, 0, 256);
    nirgan_conv_desc d = {};
    d.in = (const float*)in; d.in_elems = in_n; d.in_hp = H + 2; d.in_wp = H + 2; d.in_cs = C; d.run = C; d.in_stride = 1; d.ntaps = 9;
    for (int t = 0; t < 9; ++t) { d.tap_dh[t] = t / 3; d.tap_dw[t] = t % 3; }
    d.w = (const float*)w; d.w_elems = w_n; d.out = out; d.out_elems = out_n; d.out_hp = H; d.out_wp = H; d.out_cs = C; d.out_stride = 1;
    d.B = B; d.OH = H; d.OW = H; d.N = C; d.zero_page = zero; d.precision = 1; d.w_bf16 = 1; d.in_bf16 = 1; d.out_bf16 = out16;
    ng::ConvParams p;
    if (ng::build_conv_params(&d, p) != 0) return 1;
    if (!ng::conv_tile256_ok(p)) { printf("not eligible\n"); return 1; }
    const int nb = ((p.M + 255) >> 8) * (p.N >> 8);
    hipMalloc(&dbg, size_t(nb) * 8 * 7 * 8); hipMemset(dbg, 0, size_t(nb) * 8 * 7 * 8);
    p.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, p);      // warm the clock governor
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, p);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> r(size_t(nb) * 8 * 7);
    hipMemcpy(r.data(), dbg, r.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int i = 0; i < nb * 8; ++i) { rmin = std::min(rmin, r[i * 7 + 5]); rmax = std::max(rmax, r[i * 7 + 6]); }
    printf("blocks %d, out16 %d: %.1f us per launch (events, with stamps); last launch spans %.2f us of real time\n", nb, out16, ms * 1e3 / 20, double(rmax - rmin) / 100.0);
    const char* names[4] = {"set-up", "first DMA burst + K loop", "statistics", "epilogue"};
    for (int half = 0; half < 2; ++half) {
        double s[4] = {}, tot = 0, real = 0; int cnt = 0;
        for (int b = 0; b < nb; ++b)
            for (int wv = half * 4; wv < half * 4 + 4; ++wv) {
                const unsigned long long* o = &r[(size_t(b) * 8 + wv) * 7];
                for (int q = 0; q < 4; ++q) s[q] += double(o[q + 1] - o[q]);
                tot += double(o[4] - o[0]); real += double(o[6] - o[5]); ++cnt;
            }
        printf("waves %d-%d: %.0f cycles in %.2f us = %.2f GHz;", half * 4, half * 4 + 3, tot / cnt, real / cnt / 100.0, tot / real / 10.0);
        for (int q = 0; q < 4; ++q) printf("  %s %.0f", names[q], s[q] / cnt);
        printf("\n");
    }
    return 0;
}
