// Diagnostic build of the bf16 256-wide weight-gradient tile with in-kernel stamps (never shipped): residual-block layer (bs 16, 64 x 64,
// 256 -> 256, 3x3), one unit per workgroup; argument = number of splits (units = 9 x splits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/diag/wgrad256_stamp scripts/diag/wgrad256_stamp.hip
#define NG_DIAG256 1
#ifndef RING
#define RING 8
#endif
#include "../../nir-gan_amd/csrc/igemm_tile256.h"
#include <vector>
#include <cstring>
#include <algorithm>

void nirgan_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

__global__ __launch_bounds__(512, 2) void k(const ng::WgradParams p, const int units) {
    __shared__ __attribute__((aligned(16))) char lds[RING * ng::T256_HALF];
    for (int u = ng_xcd_remap(blockIdx.x, gridDim.x); u < units; u += gridDim.x) ng::wgrad_tile256<RING>(p, u, lds);
}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int B = 16, H = 64, C = 256;
    const int nsplit = argc > 1 ? atoi(argv[1]) : 28;
    const int rows = ((B * H * H / 64 + nsplit - 1) / nsplit) * 64;
    const size_t x_n = size_t(B) * (H + 2) * (H + 2) * C, dy_n = size_t(B) * (H + 4) * (H + 4) * C, slab_n = size_t(nsplit) * C * 9 * C;
    unsigned short *x, *dy; float *slabs, *zero; unsigned long long* dbg;
    hipMalloc(&x, x_n * 2); hipMalloc(&dy, dy_n * 2); hipMalloc(&slabs, slab_n * 4); hipMalloc(&zero, 256);
    std::vector<unsigned short> h(std::max(x_n, dy_n));
    for (size_t i = 0; i < h.size(); ++i) h[i] = bf16_of(float((i * 2654435761u) % 1000) / 500.f - 1.f);
    hipMemcpy(x, h.data(), x_n * 2, hipMemcpyHostToDevice); hipMemcpy(dy, h.data(), dy_n * 2, hipMemcpyHostToDevice); hipMemset(zero, 0, 256);
    nirgan_wgrad_desc d = {};
    d.p = (const float*)dy; d.p_elems = dy_n; d.p_hp = H + 4; d.p_wp = H + 4; d.p_cs = C; d.p_oh = 2; d.p_ow = 2;
    d.q = (const float*)x; d.q_elems = x_n; d.q_hp = H + 2; d.q_wp = H + 2; d.q_cs = C; d.q_stride = 1; d.run = C; d.ntaps = 9;
    for (int t = 0; t < 9; ++t) { d.tap_dh[t] = t / 3; d.tap_dw[t] = t % 3; }
    d.B = B; d.OH = H; d.OW = H; d.N = C; d.slabs = slabs; d.slab_elems = slab_n; d.nsplit = nsplit; d.rows_per_split = rows; d.zero_page = zero;
    d.precision = 1; d.pq_bf16 = 1;
    ng::WgradParams p;
    if (ng::build_wgrad_params(&d, p) != 0) return 1;
    if (!ng::wgrad_tile256_ok(p)) { printf("not eligible\n"); return 1; }
    const int units = 9 * nsplit, grid = units < 256 ? units : 256;
    hipMalloc(&dbg, size_t(units) * 8 * 7 * 8); hipMemset(dbg, 0, size_t(units) * 8 * 7 * 8);
    p.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, p, units);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, p, units);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> r(size_t(units) * 8 * 7);
    hipMemcpy(r.data(), dbg, r.size() * 8, hipMemcpyDeviceToHost);
    printf("splits %d x %d rows (%d K-tiles per unit), %d units on %d workgroups: %.1f us per launch (events, with stamps)\n", nsplit, rows, rows / 64, units, grid, ms * 1e3 / 20);
    const char* names[4] = {"set-up", "first DMA burst + K loop", "-", "epilogue"};
    for (int half = 0; half < 2; ++half) {
        double s[4] = {}, tot = 0, real = 0; int cnt = 0;
        for (int b = 0; b < units; ++b)
            for (int wv = half * 4; wv < half * 4 + 4; ++wv) {
                const unsigned long long* o = &r[(size_t(b) * 8 + wv) * 7];
                for (int q = 0; q < 4; ++q) s[q] += double(o[q + 1] - o[q]);
                tot += double(o[4] - o[0]); real += double(o[6] - o[5]); ++cnt;
            }
        printf("waves %d-%d: %.0f cycles in %.2f us = %.2f GHz;", half * 4, half * 4 + 3, tot / cnt, real / cnt / 100.0, tot / real / 10.0);
        for (int q = 0; q < 4; ++q) printf("  %s %.0f", names[q], s[q] / cnt);
        printf("  (K loop per K-tile %.0f)\n", s[1] / cnt / (rows / 64));
    }
    return 0;
}
