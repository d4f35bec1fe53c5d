// Diagnostic build of the convolution tile with in-kernel stamps (never shipped): where does a wave of the
// residual-block 3x3 256->256 convolution (bs 16, 64x64) spend its cycles?
#define NG_DIAG 1
#include "../../nir-gan_amd/csrc/igemm_tiles.h"
#include <vector>
#include <algorithm>

void nirgan_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

__global__ __launch_bounds__(256, 2) void k(const ng::ConvParams p) {
    __shared__ __attribute__((aligned(16))) char st0[32768];
    __shared__ __attribute__((aligned(16))) char st1[32768];
    ng::conv_tile<128, 0>(p, blockIdx.x, st0, st1);
}

int main() {
    const int B = 16, H = 64, C = 256;
    const size_t in_n = size_t(B) * (H + 2) * (H + 2) * C, out_n = size_t(B) * H * H * C, w_n = size_t(C) * 9 * C;
    float *in, *w, *out, *zero; unsigned long long* dbg;
    hipMalloc(&in, in_n * 4); hipMalloc(&w, w_n * 4); hipMalloc(&out, out_n * 4); hipMalloc(&zero, 256);
    std::vector<float> h(in_n); for (size_t i = 0; i < in_n; ++i) h[i] = float((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(w_n); for (size_t i = 0; i < w_n; ++i) hw[i] = float((i * 40503u) % 1000) / 25000.f - 0.02f;
    hipMemcpy(w, hw.data(), w_n * 4, hipMemcpyHostToDevice); hipMemset(zero, 0, 256);
    nirgan_conv_desc d = {};
    d.in = in; d.in_elems = in_n; d.in_hp = H + 2; d.in_wp = H + 2; d.in_cs = C; d.run = C; d.in_stride = 1; d.ntaps = 9;
    for (int t = 0; t < 9; ++t) { d.tap_dh[t] = t / 3; d.tap_dw[t] = t % 3; }
    d.w = w; d.w_elems = w_n; d.out = out; d.out_elems = out_n; d.out_hp = H; d.out_wp = H; d.out_cs = C; d.out_stride = 1;
    d.B = B; d.OH = H; d.OW = H; d.N = C; d.zero_page = zero;
    ng::ConvParams p;
    if (ng::build_conv_params(&d, p) != 0) return 1;
    const int nb = p.mtiles * p.ntiles;
    hipMalloc(&dbg, size_t(nb) * 4 * 6 * 8); hipMemset(dbg, 0, size_t(nb) * 4 * 6 * 8);
    p.dbg = dbg;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, p);
    hipDeviceSynchronize();
    std::vector<unsigned long long> r(size_t(nb) * 24);
    hipMemcpy(r.data(), dbg, r.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nb * 4; ++i) { tmin = std::min(tmin, r[i * 6]); tmax = std::max(tmax, r[i * 6 + 2]); }
    printf("blocks %d, kernel span %.1f us (100 MHz ticks? raw %llu)\n", nb, double(tmax - tmin) / 100.0, tmax - tmin);
    double s[2][5] = {};
    int cnt[2] = {};
    for (int i = 0; i < nb * 4; ++i) {
        const unsigned long long* o = &r[i * 6];
        const int slot = 0;                 // (o[5] is the set-up time since round 4)
        s[slot][0] += double(o[1] - o[0]); s[slot][1] += double(o[2] - o[1]); s[slot][2] += double(o[3]); s[slot][3] += double(o[4]);
        s[slot][4] += double(o[0] - tmin);
        cnt[slot]++;
    }
    for (int sl = 0; sl < 2; ++sl)
        if (cnt[sl]) printf("slot parity %d: waves %d  loop %.0f  epilogue %.0f  wait(barrier+vmcnt) %.0f  body %.0f  (per step: wait %.0f body %.0f)  mean start %.0f\n",
                            sl, cnt[sl], s[sl][0] / cnt[sl], s[sl][1] / cnt[sl], s[sl][2] / cnt[sl], s[sl][3] / cnt[sl],
                            s[sl][2] / cnt[sl] / 70.0, s[sl][3] / cnt[sl] / 70.0, s[sl][4] / cnt[sl]);
    // one block's timeline
    for (int b : {0, 1, 600}) {
        const unsigned long long* o = &r[size_t(b) * 24];
        printf("block %d wave0: start %llu loop %llu epi %llu wait %llu body %llu set-up %llu\n", b, o[0] - tmin, o[1] - o[0], o[2] - o[1], o[3], o[4], o[5]);
    }
    return 0;
}
