// Diagnostic (never shipped): residual-block 3x3 256->256 forward conv (bs 16, 64x64) with and without its epilogue,
// fp32 and bf16 operand modes -> upper bound of what overlapping the epilogue with the next tile could buy.
#ifdef SKIP
#define NG_DIAG_SKIP_EPILOGUE 1
#endif
#include "../../nir-gan_amd/csrc/igemm_tiles.h"
#include <vector>
void nirgan_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
template <int PREC>
__global__ __launch_bounds__(256, 2) void k(const ng::ConvParams p) {
    __shared__ __attribute__((aligned(16))) char st0[32768];
    __shared__ __attribute__((aligned(16))) char st1[32768];
    ng::conv_tile<128, PREC>(p, blockIdx.x, st0, st1);
}
int main() {
    const int B = 16, H = 64, C = 256;
    const size_t in_n = size_t(B) * (H + 2) * (H + 2) * C, out_n = size_t(B) * H * H * C, w_n = size_t(C) * 9 * C;
    float *in, *w, *out, *zero;
    (void)hipMalloc(&in, in_n * 4); (void)hipMalloc(&w, w_n * 4); (void)hipMalloc(&out, out_n * 4); (void)hipMalloc(&zero, 256);
    std::vector<float> h(in_n); for (size_t i = 0; i < in_n; ++i) h[i] = float((i * 2654435761u) % 1000) / 500.f - 1.f;
    (void)hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(w_n); for (size_t i = 0; i < w_n; ++i) hw[i] = float((i * 40503u) % 1000) / 25000.f - 0.02f;
    (void)hipMemcpy(w, hw.data(), w_n * 4, hipMemcpyHostToDevice); (void)hipMemset(zero, 0, 256);
    nirgan_conv_desc d = {};
    d.in = in; d.in_elems = in_n; d.in_hp = H + 2; d.in_wp = H + 2; d.in_cs = C; d.run = C; d.in_stride = 1; d.ntaps = 9;
    for (int t = 0; t < 9; ++t) { d.tap_dh[t] = t / 3; d.tap_dw[t] = t % 3; }
    d.w = w; d.w_elems = w_n; d.out = out; d.out_elems = out_n; d.out_hp = H; d.out_wp = H; d.out_cs = C; d.out_stride = 1;
    d.B = B; d.OH = H; d.OW = H; d.N = C; d.zero_page = zero;
    ng::ConvParams p;
    if (ng::build_conv_params(&d, p) != 0) return 1;
    const int nb = p.mtiles * p.ntiles;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int prec = 0; prec < 2; ++prec) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (prec == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, p); else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, p);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
            (void)hipMemset(zero, 0, 256); (void)hipDeviceSynchronize();   // a breather between launches (clock)
        }
#ifdef SKIP
        printf("prec %d WITHOUT epilogue: %.1f us\n", prec, best * 1e3f);
#else
        printf("prec %d with epilogue:    %.1f us\n", prec, best * 1e3f);
#endif
    }
    return 0;
}
