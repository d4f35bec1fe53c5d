// The generator's last layer, Conv2d(64, 1, 7) + bias + tanh (+ crop) over the reflect-padded halo'd NHWC input, and its
// backward, as direct kernels (model/networks.py:366-368: ReflectionPad2d(3), Conv2d(ngf, output_nc, 7), Tanh).
//
// 3136 multiply-adds per output pixel but ONE output channel: there is no N for the matrix pipe.  The tap-plane route
// (1x1 product into 49 planes + shifted gather) moves 228 MB of planes through HBM each way.  Here a wave's 64 lanes ARE
// the 64 input channels:
//   * the 49 weights of a lane's channel live in 49 VGPRs for the whole kernel (no LDS, no re-reads);
//   * x[pixel][lane] is one coalesced 256-byte wave load;
//   * forward: a wave slides a (4+6) x 8 register window along 4 output rows; per column 4 x 49 FMAs per lane, then ONE
//     merged 64-lane reduction of the 4 partial sums; results collect in 4 registers (one lane per output) and leave as
//     64-byte runs after 64 columns;
//     (VALU-bound: 3.3 G lane-FMAs = 84 us at 64 lanes x 4 SIMD x 256 CU x 2.4 GHz; the 281 MB activation is read once from HBM);
//   * backward: dz = dout * tanh' has ONE channel, so both gradients do have a GEMM shape -- data gradient [pixels] x [49 taps] x
//     [64 channels], weight gradient [49 taps] x [pixels] x [64 channels] -- whose 49-wide operand is an im2col of the 4.9 MB dz
//     image: v_mfma_f32_32x32x2_f32 with that operand gathered per lane straight from a zero-bordered copy of dz (cache
//     resident; no LDS, no bounds checks) and the other operand in registers (the weights) or streamed once (x).
//     First version of both on the vector ALU with dz in SGPRs: 152 / 185 us (scalar-load latency bound).
#include "common.h"

namespace {

constexpr int EK = 7, ET = 49, EC = 64;

struct EndP {
    const float* x; int x_wp, x_hp; int64_t x_img;           // halo'd input [B][x_hp][x_wp][64]
    int B, OH, OW, crop, H2, W2;
    const float* w;                                           // [49][64]
    const float* bias; int act;
    float* out;
    const float* dout;
    float* dz; int dz_rows, dz_stride; int64_t dz_img;        // zero-bordered dz image
    float* gx; float* gw; float* gbias; float* ws; int64_t ws_bias0;
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}

// ------------------------------------------------------------------------------------------------ forward
// wave = (image, 4 output rows, 64 output columns)
__global__ __launch_bounds__(256) void endconv_fwd_kernel(const EndP p, const float* __restrict__ x, const float* __restrict__ wt_, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    // (XCD-contiguous walk: a block = the four column segments of one group of 4 output rows and reads 10 input rows, 6 of them shared
    // with the next group -- dispatched round-robin, neighbouring groups sat on different XCDs and every L2 fetched its own copy:
    // 742 MB read for a 281 MB input)
#ifdef NG_END_NOREMAP
    const int wave = uni(blockIdx.x * 4 + (threadIdx.x >> 6));
#else
    const int wave = uni(ng_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
#endif
    const int segs = (p.W2 + 63) / 64, rgs = (p.H2 + 3) / 4;
    if (wave >= p.B * rgs * segs) return;
    const int seg = wave % segs, rg = (wave / segs) % rgs, b = wave / (segs * rgs);
    const int y0 = rg * 4, x0 = seg * 64;

    float w[ET];
#pragma unroll
    for (int t = 0; t < ET; ++t) w[t] = wt_[t * EC + lane];

    const float* xb = x + int64_t(b) * p.x_img;
    // rows y0+crop .. y0+crop+9 (clamped: the extra rows feed masked-off outputs only)
    int rowoff[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        int yy = y0 + p.crop + r;
        yy = yy < p.x_hp ? yy : p.x_hp - 1;
        rowoff[r] = yy * p.x_wp;
    }
    const int cbase = x0 + p.crop;
    auto ldcol = [&](float (&dst)[10], int col) {
        int cc = cbase + col;
        cc = cc < p.x_wp ? cc : p.x_wp - 1;
#pragma unroll
        for (int r = 0; r < 10; ++r) dst[r] = (xb + int64_t(rowoff[r] + cc) * EC)[lane];
    };

    float xw[8][10];
#pragma unroll
    for (int s = 0; s < 7; ++s) ldcol(xw[s], s);

    float outv[4] = {0.f, 0.f, 0.f, 0.f};
    const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0;
    for (int m = 0; m < 4; ++m) {
        float cur = 0.f;
        for (int jj = 0; jj < 16; jj += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ldcol(xw[(u + 7) & 7], m * 16 + jj + u + 7);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int kb = 0; kb < EK; ++kb) {
                    const float(&c)[10] = xw[(u + kb) & 7];
#pragma unroll
                    for (int ka = 0; ka < EK; ++ka) {
                        const float wt = w[ka * EK + kb];
                        a0 = fmaf(c[ka], wt, a0);
                        a1 = fmaf(c[ka + 1], wt, a1);
                        a2 = fmaf(c[ka + 2], wt, a2);
                        a3 = fmaf(c[ka + 3], wt, a3);
                    }
                }
                // merged reduction over the 64 channels: after the two cross-row exchanges each 16-lane row owns one output row
                // (lane rows 0..3 hold output rows 0, 2, 1, 3), then 4 DPP steps inside the row
                const float pk = hi32 ? a1 : a0, ps = hi32 ? a0 : a1;
                const float qk = hi32 ? a3 : a2, qs = hi32 ? a2 : a3;
                const float pp = pk + __shfl_xor(ps, 32, 64);
                const float qq = qk + __shfl_xor(qs, 32, 64);
                const float rk = hi16 ? qq : pp, rs = hi16 ? pp : qq;
                float r = rk + __shfl_xor(rs, 16, 64);
                r = dpp_add<0xB1>(r);        // quad_perm [1,0,3,2]
                r = dpp_add<0x4E>(r);        // quad_perm [2,3,0,1]
                r = dpp_add<0x124>(r);       // row_ror:4
                r = dpp_add<0x128>(r);       // row_ror:8
                if ((lane & 15) == jj + u) cur = r;
            }
        }
        outv[0] = outv[1]; outv[1] = outv[2]; outv[2] = outv[3]; outv[3] = cur;
    }
    // register m, lane l: output row {0,2,1,3}[l >> 4], column 16 m + (l & 15)
    const int rsel = lane >> 4;
    const int orow = y0 + ((rsel & 1) << 1 | (rsel >> 1));
    const float b0 = p.bias ? p.bias[0] : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ocol = x0 + 16 * m + (lane & 15);
        if (orow < p.H2 && ocol < p.W2) {
            const float z = outv[m] + b0;
            out[(int64_t(b) * p.H2 + orow) * p.W2 + ocol] = p.act == NIRGAN_ACT_TANH ? tanhf(z) : z;
        }
    }
}

// ------------------------------------------------------------------------------------------------ dz image
// dzp[b][yy][xx] = dout * act'(out) at (yy - 6 - crop, xx - 6 - crop), 0 outside: every tap of every pixel of the halo'd
// grid reads inside the buffer.  Also the bias gradient (sum dz), accumulated.
__global__ __launch_bounds__(256) void endconv_dz_kernel(const EndP p) {
    const int64_t total = int64_t(p.B) * p.dz_img;
    const int off = 6 + p.crop;
    float s = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int xx = int(i % p.dz_stride), yy = int((i / p.dz_stride) % p.dz_rows), b = int(i / p.dz_img);
        const int y = yy - off, x = xx - off;
        float g = 0.f;
        if (y >= 0 && y < p.H2 && x >= 0 && x < p.W2) {
            const int64_t o = (int64_t(b) * p.H2 + y) * p.W2 + x;
            g = p.dout[o];
            if (p.act == NIRGAN_ACT_TANH) {
                const float t = p.out[o];
                g *= 1.f - t * t;
            }
        }
        p.dz[i] = g;
        s += g;
    }
    if (p.gbias != nullptr) {
        s = ng_wave_sum(s);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) p.ws[p.ws_bias0 + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);      // block order: ng_partials_finish
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
// gx[b][hh][ww][c] = sum_t dzp[hh - ka_t + 6][ww - kb_t + 6] * w[t][c] over the whole halo'd grid: a GEMM with M = pixels, N = 64
// channels and K = 49 taps whose A operand is an im2col of the ONE-channel dz image (4.9 MB, cache resident) -- each lane gathers
// its own dword, no LDS -- and whose B operand (the weights) stays in 50 VGPRs for the whole kernel.
// A wave takes 32 consecutive pixels of an image's flattened halo'd grid per tile: 25 k-steps of 2 taps x 2 channel halves.
__global__ __launch_bounds__(256) void endconv_dgrad_kernel(const EndP p, const float* __restrict__ dz, const float* __restrict__ wt_, float* __restrict__ gx) {
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int nwaves = uni(gridDim.x * 4);
    // B[k = tap 2s + half][n = col (+32)]; the 50th tap does not exist: weight 0
    float wb[25][2];
#pragma unroll
    for (int s = 0; s < 25; ++s) {
        const int t = 2 * s + half;
        wb[s][0] = t < ET ? wt_[(t < ET ? t : 0) * EC + col] : 0.f;
        wb[s][1] = t < ET ? wt_[(t < ET ? t : 0) * EC + 32 + col] : 0.f;
    }
    const int npix = p.x_hp * p.x_wp;
    const int tiles_img = (npix + 31) / 32;
    const int S = p.dz_stride;
    const int ntiles = p.B * tiles_img;
    // the gathers of tile i+1 are issued before the 50 MFMAs of tile i
    auto gather = [&](int tile, float (&a)[25]) {
        tile = tile < ntiles ? tile : ntiles - 1;
        const int b = tile / tiles_img, p0 = (tile - b * tiles_img) * 32;
        int px = p0 + col;
        px = px < npix ? px : npix - 1;
        const int hh = px / p.x_wp, ww = px - hh * p.x_wp;
        // lane offsets into the dz image: taps 2s (lower half-wave) and 2s+1 (upper); the upper tap sits one column to the left, or --
        // when 2s is the last tap of a kernel row -- at the start of the next kernel row (one image row up, six columns right)
        const int zo = (hh + 6) * S + ww + 6;
        const int zoA = zo - half, zoB = zo - half * (S - 6);
        const float* zb = dz + int64_t(b) * p.dz_img;
#pragma unroll
        for (int s = 0; s < 25; ++s) {
            const int t0 = 2 * s, ka = t0 / EK, kb = t0 % EK;
            const int off = s == 24 ? zo : (kb == EK - 1 ? zoB : zoA);          // step 24: tap 48 in both halves (the weight of "tap 49" is 0)
            a[s] = zb[off - (ka * S + kb)];
        }
    };
    auto tile_out = [&](int tile, const float (&a)[25]) {
        const int b = tile / tiles_img, p0 = (tile - b * tiles_img) * 32;
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
        for (int s = 0; s < 25; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wb[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wb[s][1], acc1, 0, 0, 0);
        }
        float* o = gx + int64_t(b) * p.x_img + int64_t(p0) * EC + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (p0 + m < npix) {
                o[m * EC] = acc0[r];
                o[m * EC + 32] = acc1[r];
            }
        }
    };
    // two gather buffers in fixed roles; the empty asm keeps the next tile's gathers above this tile's MFMAs
    int tile = uni(blockIdx.x * 4 + (threadIdx.x >> 6));
    float a0[25], a1[25];
    gather(tile, a0);
    while (true) {
        if (tile >= ntiles) break;
        gather(tile + nwaves, a1);
        asm volatile("" ::: "memory");
        tile_out(tile, a0);
        tile += nwaves;
        if (tile >= ntiles) break;
        gather(tile + nwaves, a0);
        asm volatile("" ::: "memory");
        tile_out(tile, a1);
        tile += nwaves;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// gw[t][c] = sum over the halo'd grid of dzp[hh - ka_t + 6][ww - kb_t + 6] * x[hh][ww][c]: M = 49 taps (two 32-row tiles), N = 64
// channels, K = pixels.  A (the dz im2col) is gathered per lane as above; B is x itself, read exactly once: a lane loads the
// channel pair (2 col, 2 col + 1) of its pixel, so the two N tiles are the even and the odd channels.
// Work unit = 8 k-steps (16 pixels of one image row); units are dealt round-robin to 2 waves per SIMD (balanced to one unit), and
// the loads of units i+1 and i+2 are in flight under the 32 MFMAs of unit i (x streams from HBM: ~64 KB in flight per CU).
// A block adds its 4 waves in LDS and writes one partial [49][64].
__global__ __launch_bounds__(256) void endconv_wgrad_kernel(const EndP p, const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ ws) {
    __shared__ float red[ET * EC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int S = p.dz_stride;
    // taps of this lane's A rows (rows 49..63 of the second tile do not exist: they re-read tap 48 and are dropped); lane offsets
    // relative to dzp[hh][ww], all >= 0
    const int t0 = col, t1 = col + 32 < ET ? col + 32 : ET - 1;
    const int la0 = (6 - t0 / EK) * S + 6 - t0 % EK + half, la1 = (6 - t1 / EK) * S + 6 - t1 % EK + half;
    const int lx = half * EC + 2 * col;
    const int npair = (p.x_wp + 1) / 2, cpr = (npair + 7) / 8;
    const int nunits = p.B * p.x_hp * cpr, nwaves = uni(gridDim.x * 4);
    struct Buf { float a0[8], a1[8]; f32x2 xv[8]; };
    auto load = [&](int unit, Buf& f) {
        const bool live = unit < nunits;                     // past the end: a repeat with x = 0 (the loop runs whole rounds of 3)
        unit = live ? unit : nunits - 1;
        const int row = unit / cpr, q0 = (unit - row * cpr) * 8;
        const int b = row / p.x_hp, hh = row - b * p.x_hp;
        const float* zu = dz + (int64_t(b) * p.dz_img + hh * S);
        const float* xu = x + (int64_t(b) * p.x_img + int64_t(hh) * p.x_wp * EC);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + j < npair ? q0 + j : npair - 1;       // dz addresses clamped; x past the row end reads zeros:
            const int ww = 2 * q;                                    // the first row of the dz image is border
            f.a0[j] = (zu + ww)[la0];
            f.a1[j] = (zu + ww)[la1];
            const bool ok = live && q0 + j < npair && ww + half < p.x_wp;
            const float* xp = ok ? xu + ww * EC + lx : dz + lx;
            f.xv[j] = *reinterpret_cast<const f32x2*>(xp);
        }
    };
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    auto mma = [&](const Buf& f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[j], f.xv[j][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[j], f.xv[j][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[j], f.xv[j][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[j], f.xv[j][1], acc[1][1], 0, 0, 0);
        }
    };
    // three buffers in fixed roles (no register copies); the empty asm keeps the compiler from sinking a unit's loads below the
    // MFMAs they are meant to fly under
    Buf f0, f1, f2;
    int unit = uni(blockIdx.x * 4 + wv);
    const int rounds = unit < nunits ? ((nunits - unit + nwaves - 1) / nwaves + 2) / 3 : 0;
    load(unit, f0);
    load(unit + nwaves, f1);
    for (int it = 0; it < rounds; ++it) {
        load(unit + 2 * nwaves, f2);
        asm volatile("" ::: "memory");
        mma(f0);
        load(unit + 3 * nwaves, f0);
        asm volatile("" ::: "memory");
        mma(f1);
        load(unit + 4 * nwaves, f1);
        asm volatile("" ::: "memory");
        mma(f2);
        unit += 3 * nwaves;
    }
    // D[m = tap][n]: tile nt holds channel 2 col + nt; the 4 waves add up in LDS in wave order
    for (int w4 = 0; w4 < 4; ++w4) {
        if (wv == w4) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (t < ET) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int i = t * EC + 2 * col + nt;
                            red[i] = w4 == 0 ? acc[mt][nt][r] : red[i] + acc[mt][nt][r];
                        }
                    }
                }
        }
        __syncthreads();
    }
    float* o = ws + int64_t(blockIdx.x) * (ET * EC);
    for (int i = threadIdx.x; i < ET * EC; i += 256) o[i] = red[i];
}

// gw[c][t] (the Conv2d weight's own [1][64][7][7] layout) = sum over the partials: a block takes 16 outputs x 16 slices of the
// partial list (fixed order: slice sums first, then the 16 slices)
__global__ __launch_bounds__(256) void endconv_wgrad_finish_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ gw) {
    __shared__ float part[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;                  // t * 64 + c  (49 * 64 is a multiple of 16)
    float s = 0.f;
#pragma unroll 8
    for (int k = sl; k < nblk; k += 16) s += ws[int64_t(k) * (ET * EC) + i];
    part[sl][o] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
        const int j = blockIdx.x * 16 + threadIdx.x;
        gw[(j & 63) * ET + (j >> 6)] = t;
    }
}

constexpr int WGRAD_MAX_BLOCKS = 512;                              // 2 waves per SIMD (the kernel holds ~190 VGPRs)
int wgrad_blocks(int B, int OH) {
    const int64_t b = (int64_t(B) * (OH + EK - 1) + 3) / 4;
    return int(b < WGRAD_MAX_BLOCKS ? b : WGRAD_MAX_BLOCKS);
}

int fill_params(const nirgan_endconv_desc* d, EndP& p, const char* what) {
    NG_REQUIRE(d != nullptr, "%s: null descriptor", what);
    NG_REQUIRE(d->C == EC && d->k == EK, "%s: the direct kernels cover Conv2d(64, 1, 7) only (C=%d k=%d)", what, d->C, d->k);
    NG_REQUIRE(d->B > 0 && d->OH > 0 && d->OW > 0 && d->crop >= 0 && d->OH > 2 * d->crop && d->OW > 2 * d->crop, "%s: bad shape", what);
    NG_REQUIRE(d->x_hp == d->OH + EK - 1 && d->x_wp == d->OW + EK - 1, "%s: input halo geometry mismatch", what);
    NG_REQUIRE(d->w != nullptr, "%s: null weights", what);
    p.x = d->x; p.x_wp = d->x_wp; p.x_hp = d->x_hp; p.x_img = int64_t(d->x_hp) * d->x_wp * EC;
    p.B = d->B; p.OH = d->OH; p.OW = d->OW; p.crop = d->crop; p.H2 = d->OH - 2 * d->crop; p.W2 = d->OW - 2 * d->crop;
    p.w = d->w; p.bias = d->bias; p.act = d->act; p.out = d->out; p.dout = d->dout;
    p.dz = d->dz; p.dz_rows = d->x_hp + 6; p.dz_stride = (d->x_wp + 6 + 3) / 4 * 4 + 8; p.dz_img = int64_t(p.dz_rows) * p.dz_stride;
    p.gx = d->gx; p.gw = d->gw; p.gbias = d->gbias; p.ws = d->ws;
    NG_REQUIRE(int64_t(d->B) * p.dz_img < (1ll << 31) && int64_t(d->x_hp) * d->x_wp < (1ll << 24), "%s: problem too large", what);
    return NIRGAN_OK;
}

}  // namespace

extern "C" int64_t nirgan_endconv_dz_elems(int B, int OH, int OW) {
    if (B <= 0 || OH <= 0 || OW <= 0) return 0;
    return int64_t(B) * (OH + EK - 1 + 6) * ((OW + EK - 1 + 6 + 3) / 4 * 4 + 8);
}

extern "C" int64_t nirgan_endconv_ws_elems(int B, int OH, int OW) {
    (void)OW;
    if (B <= 0 || OH <= 0) return 0;
    return int64_t(wgrad_blocks(B, OH)) * ET * EC + 1024;         // + the dz pass's per-block partial sums of the bias gradient
}

extern "C" int nirgan_endconv_fwd(const nirgan_endconv_desc* d, void* stream) {
    EndP p;
    const int rc = fill_params(d, p, "endconv_fwd");
    if (rc != NIRGAN_OK) return rc;
    NG_REQUIRE(d->x && d->out, "endconv_fwd: null pointer");
    const int64_t waves = int64_t(p.B) * ((p.H2 + 3) / 4) * ((p.W2 + 63) / 64);
    hipLaunchKernelGGL(endconv_fwd_kernel, dim3(int((waves + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), p, p.x, p.w, p.out);
    return nirgan_check_launch("endconv_fwd");
}

extern "C" int nirgan_endconv_dz(const nirgan_endconv_desc* d, void* stream) {
    EndP p;
    const int rc = fill_params(d, p, "endconv_dz");
    if (rc != NIRGAN_OK) return rc;
    NG_REQUIRE(d->dout && d->dz && (d->act != NIRGAN_ACT_TANH || d->out), "endconv_dz: null pointer");
    NG_REQUIRE(d->dz_elems >= nirgan_endconv_dz_elems(d->B, d->OH, d->OW) && ng_aligned16(d->dz), "endconv_dz: workspace too small or unaligned");
    const int64_t total = int64_t(p.B) * p.dz_img;
    const int64_t g = (total + 255) / 256;
    const int nb = int(g < 1024 ? g : 1024);
    if (d->gbias) {
        NG_REQUIRE(d->ws && d->ws_elems >= nirgan_endconv_ws_elems(d->B, d->OH, d->OW), "endconv_dz: gbias needs the partials workspace (nirgan_endconv_ws_elems)");
        p.ws_bias0 = nirgan_endconv_ws_elems(d->B, d->OH, d->OW) - 1024;
    }
    hipLaunchKernelGGL(endconv_dz_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    if (d->gbias) return ng_partials_finish(p.ws + p.ws_bias0, nb, 1, d->gbias, static_cast<hipStream_t>(stream));
    return nirgan_check_launch("endconv_dz");
}

extern "C" int nirgan_endconv_dgrad(const nirgan_endconv_desc* d, void* stream) {
    EndP p;
    const int rc = fill_params(d, p, "endconv_dgrad");
    if (rc != NIRGAN_OK) return rc;
    NG_REQUIRE(d->dz && d->gx, "endconv_dgrad: null pointer");
    NG_REQUIRE(d->dz_elems >= nirgan_endconv_dz_elems(d->B, d->OH, d->OW) && ng_aligned16(d->dz), "endconv_dgrad: workspace too small or unaligned");
    const int64_t blocks = (int64_t(p.B) * ((p.x_hp * p.x_wp + 31) / 32) + 3) / 4;
    hipLaunchKernelGGL(endconv_dgrad_kernel, dim3(int(blocks < 2048 ? blocks : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), p, (const float*)p.dz, p.w, p.gx);
    return nirgan_check_launch("endconv_dgrad");
}

extern "C" int nirgan_endconv_wgrad(const nirgan_endconv_desc* d, void* stream) {
    EndP p;
    const int rc = fill_params(d, p, "endconv_wgrad");
    if (rc != NIRGAN_OK) return rc;
    NG_REQUIRE(d->x && d->dz && d->gw && d->ws, "endconv_wgrad: null pointer");
    NG_REQUIRE(d->dz_elems >= nirgan_endconv_dz_elems(d->B, d->OH, d->OW) && ng_aligned16(d->dz), "endconv_wgrad: dz workspace too small or unaligned");
    NG_REQUIRE(d->ws_elems >= nirgan_endconv_ws_elems(d->B, d->OH, d->OW), "endconv_wgrad: partials workspace too small");
    const int nblk = wgrad_blocks(d->B, d->OH);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(endconv_wgrad_kernel, dim3(nblk), dim3(256), 0, st, p, (const float*)p.dz, p.x, p.ws);
    hipLaunchKernelGGL(endconv_wgrad_finish_kernel, dim3(ET * EC / 16), dim3(256), 0, st, p.ws, nblk, p.gw);
    return nirgan_check_launch("endconv_wgrad");
}
