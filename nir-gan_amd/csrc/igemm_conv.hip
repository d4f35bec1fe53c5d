// Implicit-GEMM valid convolution over halo'd NHWC fp32 buffers on the gfx950 matrix pipe.
//
//   M = B*OH*OW output pixels, N = output channels, K = ntaps*run.
//   Block tile 128(M) x BN(N) x 32(K), 4 waves as 2x2, each wave 64 x BN/2 built from
//   v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/issue/SIMD).  A (pixels x K-slice) and
//   B (weights, [N][K] K-contiguous) tiles are staged by LDS-DMA (global_load_lds_dwordx4):
//   one wave-instruction lands 8 rows x 128 B.  Rows are 128 B, so the 16-byte chunk index
//   is XOR-swizzled with (row>>1)&7 on the *source* side (the LDS image stays lane-linear)
//   and on the ds_read_b128 side, which makes the fragment reads bank-conflict free.
//   One ds_read_b128 gives a lane 4 consecutive k of its row: lanes 0-31 take chunk 2g,
//   lanes 32-63 chunk 2g+1, so MFMA j of the group contracts k = 8g+j and 8g+4+j; A and B
//   use the same order, the sum is a permutation of the k order only.
//   Two LDS stages, one barrier per K-step: the DMA of step s+1 flies under the 64 MFMAs
//   (4096 cycles) of step s.  Out-of-range rows/channels read a zero page.
#include "common.h"

namespace {

struct ConvParams {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    const float* zero;
    int in_row, in_img, in_cs, run, in_stride, in_org;
    int ntaps;
    int tap_off[NIRGAN_MAX_TAPS];
    int K;
    int out_cs, out_row, out_img, out_stride, out_org;
    int OW, OHW, M, N;
    int mtiles, ntiles;
};

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    constexpr int BM = 128;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NT = BN / 64;   // 32-column MFMA tiles per wave
    constexpr int BI = BN / 32;   // B loader instructions per wave (8 rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = ng_xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int n0 = (id % p.ntiles) * BN, m0 = (id / p.ntiles) * BM;

    // ---------------- loader state: each lane owns one 16-byte chunk of 4 A rows and BI B rows
    const int lrow = lane >> 3, lchunk = lane & 7;
    int a_base[4], a_col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        const int lc = lchunk ^ ((row >> 1) & 7);
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const int b = m / p.OHW, r = m - b * p.OHW;
        const int oh = r / p.OW, ow = r - oh * p.OW;
        a_col[i] = lc * 4;
        a_base[i] = b * p.in_img + oh * p.in_stride * p.in_row + ow * p.in_stride * p.in_cs + p.in_org + lc * 4;
    }
    int b_base[BI], b_col[BI];
    bool b_ok[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (wave * BI + i) * 8 + lrow;
        const int lc = lchunk ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_ok[i] = n < p.N;
        b_col[i] = lc * 4;
        b_base[i] = (b_ok[i] ? n : 0) * p.K + lc * 4;
    }

    auto issue = [&](int stage, int t, int c0) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + A_BYTES;
        const int toff = p.tap_off[t] + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* src = (c0 + a_col[i] < p.run) ? p.in + (a_base[i] + toff) : p.zero;
            ng_glds16(src, sA + (wave * 4 + i) * 1024);
        }
        const int woff = t * p.run + c0;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const float* src = (b_ok[i] && c0 + b_col[i] < p.run) ? p.w + (b_base[i] + woff) : p.zero;
            ng_glds16(src, sB + (wave * BI + i) * 1024);
        }
    };

    // ---------------- compute state
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    int a_off[2], a_key[2], b_off[NT], b_key[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wr * 64 + mt * 32 + (lane & 31);
        a_off[mt] = row * 128;
        a_key[mt] = (row >> 1) & 7;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wc * (BN / 2) + nt * 32 + (lane & 31);
        b_off[nt] = row * 128;
        b_key[nt] = (row >> 1) & 7;
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    auto compute = [&](int stage) {
        const char* sA = smem + stage * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + half;
            f32x4 a[2], b[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                a[mt] = *reinterpret_cast<const f32x4*>(sA + a_off[mt] + ((chunk ^ a_key[mt]) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const f32x4*>(sB + b_off[nt] + ((chunk ^ b_key[nt]) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
        }
    };

    // ---------------- main loop: K-steps enumerate (tap, 32-float slice of the run)
    const int csteps = (p.run + 31) >> 5;
    const int nk = p.ntaps * csteps;
    int t = 0, c0 = 0;
    issue(0, 0, 0);
    for (int s = 0; s < nk; ++s) {
        int c1 = c0 + 32, t1 = t;
        if (c1 >= p.run) { c1 = 0; t1 = t + 1; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) issue((s + 1) & 1, t1, c1);
        compute(s & 1);
        t = t1; c0 = c1;
    }

    // ---------------- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + wc * (BN / 2) + nt * 32 + (lane & 31);
            const bool col_ok = col < p.N;
            const float bv = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M && col_ok) {
                    const int b = m / p.OHW, rr = m - b * p.OHW;
                    const int oh = rr / p.OW, ow = rr - oh * p.OW;
                    const int off = b * p.out_img + oh * p.out_stride * p.out_row + ow * p.out_stride * p.out_cs + p.out_org + col;
                    p.out[off] = acc[mt][nt][r] + bv;
                }
            }
        }
    }
}

}  // namespace

extern "C" int nirgan_conv_igemm(const nirgan_conv_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr, "conv_igemm: null descriptor");
    NG_REQUIRE(d->in && d->w && d->out && d->zero_page, "conv_igemm: null pointer");
    NG_REQUIRE(ng_aligned16(d->in) && ng_aligned16(d->w) && ng_aligned16(d->zero_page), "conv_igemm: in/w/zero_page must be 16-byte aligned");
    NG_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->N > 0, "conv_igemm: empty problem B=%d OH=%d OW=%d N=%d", d->B, d->OH, d->OW, d->N);
    NG_REQUIRE(d->ntaps >= 1 && d->ntaps <= NIRGAN_MAX_TAPS, "conv_igemm: ntaps=%d out of range", d->ntaps);
    NG_REQUIRE(d->run > 0 && d->run % 4 == 0 && d->in_cs > 0 && d->in_cs % 4 == 0, "conv_igemm: run=%d and in_cs=%d must be positive multiples of 4", d->run, d->in_cs);
    NG_REQUIRE(d->in_stride >= 1 && d->out_stride >= 1, "conv_igemm: strides must be >= 1");
    NG_REQUIRE(d->in_elems < (int64_t(1) << 31) && d->out_elems < (int64_t(1) << 31) && d->w_elems < (int64_t(1) << 31), "conv_igemm: buffers must be < 2^31 floats");
    NG_REQUIRE(d->in_elems >= int64_t(d->B) * d->in_hp * d->in_wp * d->in_cs, "conv_igemm: in_elems too small");
    NG_REQUIRE(d->out_elems >= int64_t(d->B) * d->out_hp * d->out_wp * d->out_cs, "conv_igemm: out_elems too small");
    NG_REQUIRE(d->w_elems >= int64_t(d->N) * d->ntaps * d->run, "conv_igemm: w_elems too small");
    NG_REQUIRE(d->N <= d->out_cs, "conv_igemm: N=%d exceeds out_cs=%d", d->N, d->out_cs);
    int dh0 = d->tap_dh[0], dh1 = d->tap_dh[0], dw0 = d->tap_dw[0], dw1 = d->tap_dw[0];
    for (int t = 1; t < d->ntaps; ++t) {
        dh0 = d->tap_dh[t] < dh0 ? d->tap_dh[t] : dh0; dh1 = d->tap_dh[t] > dh1 ? d->tap_dh[t] : dh1;
        dw0 = d->tap_dw[t] < dw0 ? d->tap_dw[t] : dw0; dw1 = d->tap_dw[t] > dw1 ? d->tap_dw[t] : dw1;
    }
    NG_REQUIRE(d->in_oh + dh0 >= 0 && (d->OH - 1) * d->in_stride + d->in_oh + dh1 < d->in_hp, "conv_igemm: input rows out of range");
    NG_REQUIRE(d->in_ow + dw0 >= 0 && int64_t((d->OW - 1) * d->in_stride + d->in_ow + dw1) * d->in_cs + d->run <= int64_t(d->in_wp) * d->in_cs, "conv_igemm: input columns out of range");
    NG_REQUIRE(d->out_oh >= 0 && (d->OH - 1) * d->out_stride + d->out_oh < d->out_hp && d->out_ow >= 0 && (d->OW - 1) * d->out_stride + d->out_ow < d->out_wp, "conv_igemm: output window out of range");

    ConvParams p;
    p.in = d->in; p.w = d->w; p.bias = d->bias; p.out = d->out; p.zero = d->zero_page;
    p.in_cs = d->in_cs; p.in_row = d->in_wp * d->in_cs; p.in_img = d->in_hp * p.in_row;
    p.run = d->run; p.in_stride = d->in_stride; p.in_org = d->in_oh * p.in_row + d->in_ow * d->in_cs;
    p.ntaps = d->ntaps;
    for (int t = 0; t < NIRGAN_MAX_TAPS; ++t) p.tap_off[t] = t < d->ntaps ? d->tap_dh[t] * p.in_row + d->tap_dw[t] * d->in_cs : 0;
    p.K = d->ntaps * d->run;
    p.out_cs = d->out_cs; p.out_row = d->out_wp * d->out_cs; p.out_img = d->out_hp * p.out_row;
    p.out_stride = d->out_stride; p.out_org = d->out_oh * p.out_row + d->out_ow * d->out_cs;
    p.OW = d->OW; p.OHW = d->OH * d->OW;
    const int64_t M = int64_t(d->B) * p.OHW;
    NG_REQUIRE(M < (int64_t(1) << 31), "conv_igemm: too many output pixels");
    p.M = int(M); p.N = d->N;
    p.mtiles = (p.M + 127) / 128;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->N > 64) {
        p.ntiles = (d->N + 127) / 128;
        static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess; }();
        (void)once;
        hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3(p.mtiles * p.ntiles), dim3(256), 65536, st, p);
    } else {
        p.ntiles = 1;
        hipLaunchKernelGGL(conv_igemm_kernel<64>, dim3(p.mtiles), dim3(256), 49152, st, p);
    }
    return nirgan_check_launch("conv_igemm");
}
