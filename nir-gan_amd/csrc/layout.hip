// NCHW boundary <-> halo'd NHWC, and the tap-plane gather/scatter of the single-output-channel
// convolutions (7x7 64->1 + tanh of the generator, 4x4 512->1 of the PatchGAN).  HBM-bound.
#include "common.h"
#include <cstdlib>

namespace {

__global__ void nchw_to_halo_kernel(const float* __restrict__ src, int B, int Cs, int H, int W,
                                    float* __restrict__ dst, int cs, int c0, int pad1, int pad2, int reflect) {
    const int P = pad1 + pad2;
    const int Hp = H + 2 * P, Wp = W + 2 * P;
    const int64_t total = int64_t(B) * Hp * Wp;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int ww = int(i % Wp);
        const int hh = int((i / Wp) % Hp);
        const int b = int(i / (int64_t(Wp) * Hp));
        int h = hh - P, w = ww - P;
        if (reflect) {
            // ReflectionPad2d(pad2) applied to F.pad(x, pad1, 'reflect'): undo pad2 first, then pad1
            h = ng_reflect(hh - pad2, H + 2 * pad1) - pad1;
            w = ng_reflect(ww - pad2, W + 2 * pad1) - pad1;
            h = ng_reflect(h, H);
            w = ng_reflect(w, W);
        } else if (h < 0 || h >= H || w < 0 || w >= W) {
            continue;
        }
        float* d = dst + i * cs + c0;
        for (int c = 0; c < Cs; ++c) d[c] = src[((int64_t(b) * Cs + c) * H + h) * W + w];
    }
}

struct GatherP {
    const float* q; int q_row, q_img, q_cs, q_hp, q_wp;
    int ntaps; int dh[64], dw[64]; int kh, kw;
    const float* bias; int act;
    int B, OH, OW, crop;
    float* dst;
};

// Block = 8 x 32 outputs.  The (8+kh-1) x (32+kw-1) window of tap-plane records is staged in LDS with coalesced
// 16-byte loads (records are q_cs contiguous floats) at an ODD float stride per record, so that the 49 (16) reads of
// a thread -- record (y+kh, x+kw), plane kh*k+kw -- are bank-conflict free across the 32 x-neighbours.
__global__ __launch_bounds__(256) void tap_gather_kernel(const GatherP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int H2 = p.OH - 2 * p.crop, W2 = p.OW - 2 * p.crop;
    const int tiles_x = (W2 + 31) / 32, tiles_y = (H2 + 7) / 8;
    const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * 8, tx0 = (tile % tiles_x) * 32;
    const int q4 = p.q_cs / 4, stride = p.q_cs | 1;
    const int rh = 8 + p.kh - 1, rw = 32 + p.kw - 1;
    // window origin in q coordinates: output (y, x) reads q rows y+crop+dh, cols x+crop+dw (dh, dw >= 0)
    const float* qb = p.q + int64_t(b) * p.q_img;
    for (int i = threadIdx.x; i < rh * rw * q4; i += 256) {
        const int c = i % q4, px = i / q4;
        const int wy = px / rw, wx = px - wy * rw;
        int gy = ty0 + p.crop + wy, gx = tx0 + p.crop + wx;
        gy = gy < p.q_hp ? gy : p.q_hp - 1;          // clamped rows/cols are only read by masked-off outputs
        gx = gx < p.q_wp ? gx : p.q_wp - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(qb + int64_t(gy) * p.q_row + int64_t(gx) * p.q_cs + c * 4);
        float* d = lds + px * stride + c * 4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W2 || y >= H2) return;
    float s = p.bias ? p.bias[0] : 0.f;
    for (int t = 0; t < p.ntaps; ++t) s += lds[((ly + p.dh[t]) * rw + lx + p.dw[t]) * stride + t];
    if (p.act == NIRGAN_ACT_TANH) s = tanhf(s);
    p.dst[(int64_t(b) * H2 + y) * W2 + x] = s;
}

// Full k x k tap sets in row-major order (dh = t / k, dw = t % k: what the 7x7 and 4x4 single-channel convolutions emit): a block takes
// 32 x 32 outputs and walks the kernel rows; per row it stages only the k planes of that row for the (32 x (32+k-1)) records it needs
// ([r][c][j], 7 or 4 floats per record: odd / small stride, conflict-free across the 32 x-neighbours), 4 outputs per thread.  34 KB of LDS
// instead of the 110 KB window of all planes (one block per CU), halo re-reads 1.4x instead of 2.1x.
__global__ __launch_bounds__(256) void tap_gather_rows_kernel(const GatherP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int H2 = p.OH - 2 * p.crop, W2 = p.OW - 2 * p.crop;
    const int tiles_x = (W2 + 31) / 32, tiles_y = (H2 + 31) / 32;
    const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * 32, tx0 = (tile % tiles_x) * 32;
    const int k = p.kw, rw = 32 + k - 1;
    const float* qb = p.q + int64_t(b) * p.q_img;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;      // outputs (ly + 8 i, lx), i = 0..3
    float s[4];
    const float b0 = p.bias ? p.bias[0] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = b0;
    const int per_row = rw * k;                                  // floats staged per window row
    for (int kh = 0; kh < p.kh; ++kh) {
        __syncthreads();                                         // the previous kernel row is consumed
        // a thread keeps its (column, plane) and walks the 32 rows: the index arithmetic is paid once, the loads of a row are
        // k-float runs at the record stride
        for (int cj = threadIdx.x; cj < per_row; cj += 256) {
            const int c = cj / k, j = cj - c * k;
            int gx = tx0 + p.crop + c;
            gx = gx < p.q_wp ? gx : p.q_wp - 1;                  // clamped rows / columns feed masked-off outputs only
            const float* col = qb + int64_t(gx) * p.q_cs + kh * k + j;
            const int gy0 = ty0 + p.crop + kh;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                int gy = gy0 + r;
                gy = gy < p.q_hp ? gy : p.q_hp - 1;
                lds[r * per_row + cj] = col[int64_t(gy) * p.q_row];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* row = lds + (ly + 8 * i) * per_row + lx * k;
            float a = 0.f;
            for (int kw = 0; kw < k; ++kw) a += row[kw * k + kw];
            s[i] += a;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = tx0 + lx, y = ty0 + ly + 8 * i;
        if (x < W2 && y < H2) p.dst[(int64_t(b) * H2 + y) * W2 + x] = p.act == NIRGAN_ACT_TANH ? tanhf(s[i]) : s[i];
    }
}

struct ScatterP {
    const float* dout; const float* out; int act;
    int B, OH, OW, crop;
    int ntaps; int dh[64], dw[64];
    float* dq; int q_hp, q_wp, q_cs;
    float* dbias;
};

// one thread per (pixel of dq, group of 4 taps): dq[b][hh][ww][t] = dz[hh-dh_t][ww-dw_t]
__global__ void tap_scatter_kernel(const ScatterP p) {
    // the tap tables are indexed per lane (t = 4 tq + k): from LDS, not from the kernel arguments (a divergent index into those is serialised)
    __shared__ int s_dh[64], s_dw[64];
    if (threadIdx.x < 64) { s_dh[threadIdx.x] = p.dh[threadIdx.x] + p.crop; s_dw[threadIdx.x] = p.dw[threadIdx.x] + p.crop; }
    __syncthreads();
    const int H2 = p.OH - 2 * p.crop, W2 = p.OW - 2 * p.crop;
    const int q4 = p.q_cs / 4;
    const int64_t total = int64_t(p.B) * p.q_hp * p.q_wp * q4;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int tq = int(i % q4);
        const int64_t pix = i / q4;
        const int ww = int(pix % p.q_wp), hh = int((pix / p.q_wp) % p.q_hp), b = int(pix / (int64_t(p.q_wp) * p.q_hp));
        f32x4 v = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = tq * 4 + k;
            if (t < p.ntaps) {
                const int h = hh - s_dh[t], w = ww - s_dw[t];
                if (h >= 0 && h < H2 && w >= 0 && w < W2) {
                    const int64_t o = (int64_t(b) * H2 + h) * W2 + w;
                    float g = p.dout[o];
                    if (p.act == NIRGAN_ACT_TANH) {
                        const float y = p.out[o];
                        g *= 1.f - y * y;
                    }
                    v[k] = g;
                }
            }
        }
        *reinterpret_cast<f32x4*>(p.dq + pix * p.q_cs + tq * 4) = v;
    }
}

// ONE block of 1024 threads (four elements in flight per thread and trip): the bias gradient of a single-output-channel layer is a sum
// over its (small) output map; no atomics between blocks, bitwise reproducible.  dbias += the sum.
__global__ __launch_bounds__(1024) void tap_dbias_kernel(const float* __restrict__ dout, const float* __restrict__ out, int act, int64_t n, float* dbias) {
    float s = 0.f;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += 4096) {
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + j * 1024;
            g[j] = i < n ? dout[i] : 0.f;
            if (act == NIRGAN_ACT_TANH && i < n) {
                const float y = out[i];
                g[j] *= 1.f - y * y;
            }
        }
        s += (g[0] + g[1]) + (g[2] + g[3]);
    }
    s = ng_wave_sum(s);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        dbias[0] += t;
    }
}

struct ChanDgradP {
    const float* dy; int dy_row, dy_img, dy_pad, C, OH, OW;
    const float* w; int cin, k, stride, pad, channel;
    int B, H, W;
    float* out;
};

// out[b][h][w] = sum_{kh,kw,co} dY[b][(h+p-kh)/s][(w+p-kw)/s][co] * W[co][c][kh][kw] over taps with integral indices.
// 16 lanes (float4 each) cover the channels of one pixel, a wave handles 4 pixels; the k*k*C weights of the
// selected input channel are staged once per block in LDS.
__global__ __launch_bounds__(256) void chan_dgrad_kernel(const ChanDgradP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wsel = reinterpret_cast<float*>(smem_raw);        // [k*k][C]
    const int kk = p.k * p.k;
    for (int i = threadIdx.x; i < kk * p.C; i += 256) {
        const int t = i / p.C, co = i - t * p.C;
        wsel[i] = p.w[(size_t(co) * p.cin + p.channel) * kk + t];
    }
    __syncthreads();
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;          // 16 pixels per block iteration
    const int64_t npix = int64_t(p.B) * p.H * p.W;
    const int q4 = p.C / 4;
    for (int64_t base = int64_t(blockIdx.x) * 16; base < npix; base += int64_t(gridDim.x) * 16) {
        const int64_t pix = base + grp;
        float acc = 0.f;
        if (pix < npix) {
            const int w = int(pix % p.W), h = int((pix / p.W) % p.H), b = int(pix / (int64_t(p.W) * p.H));
            for (int kh = 0; kh < p.k; ++kh) {
                const int nh = h + p.pad - kh;
                if (nh % p.stride) continue;
                const int oh = nh / p.stride;
                if (nh < 0 || oh >= p.OH) continue;
                for (int kw = 0; kw < p.k; ++kw) {
                    const int nw = w + p.pad - kw;
                    if (nw % p.stride) continue;
                    const int ow = nw / p.stride;
                    if (nw < 0 || ow >= p.OW) continue;
                    const float* d = p.dy + size_t(b) * p.dy_img + size_t(oh + p.dy_pad) * p.dy_row + size_t(ow + p.dy_pad) * p.C;
                    const float* ws = wsel + (kh * p.k + kw) * p.C;
                    for (int q = sub; q < q4; q += 16) {
                        const f32x4 dv = *reinterpret_cast<const f32x4*>(d + q * 4);
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + q * 4);
                        acc += dv[0] * wv[0] + dv[1] * wv[1] + dv[2] * wv[2] + dv[3] * wv[3];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0 && pix < npix) p.out[pix] = acc;
    }
}

// k = 4, stride 2, padding 1 (the PatchGAN's first convolution, model/networks.py:559): output-stationary on dY.  A block takes 8x8 dY
// positions plus a ring of one (10x10), forms the 16 tap products T[pos][kh][kw] = sum_co dY[pos][co] * W[co][c][kh][kw] once per
// position in LDS (each dY vector is read once per block instead of once per tap and pixel), then every one of its 16x16 input pixels
// adds its four taps: out(h, w) = sum over kh = (h+1)%2 + {0,2}, kw likewise, of T[(h+1-kh)/2][(w+1-kw)/2][kh][kw].
__global__ __launch_bounds__(256) void chan_dgrad_k4s2_kernel(const ChanDgradP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wsel = reinterpret_cast<float*>(smem_raw);        // [16][C]
    float* T = wsel + 16 * p.C;                              // [100][16]
    for (int i = threadIdx.x; i < 16 * p.C; i += 256) {
        const int t = i / p.C, co = i - t * p.C;
        wsel[i] = p.w[(size_t(co) * p.cin + p.channel) * 16 + t];
    }
    __syncthreads();
    const int tiles_x = (p.OW + 7) / 8, tiles_y = (p.OH + 7) / 8;
    const int ntiles = p.B * tiles_y * tiles_x;
    const int q4 = p.C / 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int oh0 = ty * 8 - 1, ow0 = tx * 8 - 1;        // first position of the 10x10 patch
        // phase 1: thread (pos, half) forms 8 of the 16 tap products of its position
        {
            const int pos = threadIdx.x & 127, th = threadIdx.x >> 7;
            if (pos < 100) {
                const int py = pos / 10, px = pos - py * 10;
                const int oh = oh0 + py, ow = ow0 + px;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (oh >= 0 && oh < p.OH && ow >= 0 && ow < p.OW) {
                    const float* d = p.dy + size_t(b) * p.dy_img + size_t(oh + p.dy_pad) * p.dy_row + size_t(ow + p.dy_pad) * p.C;
                    for (int q = 0; q < q4; ++q) {
                        const f32x4 dv = *reinterpret_cast<const f32x4*>(d + q * 4);
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(wsel + (th * 8 + t) * p.C + q * 4);
                            acc[t] += dv[0] * wv[0] + dv[1] * wv[1] + dv[2] * wv[2] + dv[3] * wv[3];
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) T[pos * 16 + th * 8 + t] = acc[t];
            }
        }
        __syncthreads();
        // phase 2: the 16x16 input pixels of the tile
        {
            const int y = threadIdx.x >> 4, x = threadIdx.x & 15;
            const int h = ty * 16 + y, w = tx * 16 + x;
            if (h < p.H && w < p.W) {
                const int kh0 = (h + 1) & 1, kw0 = (w + 1) & 1;
                float s = 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int kh = kh0 + 2 * a;
                    const int py = ((h + 1 - kh) >> 1) - oh0;            // 0 .. 9 by construction
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int kw = kw0 + 2 * c;
                        const int px = ((w + 1 - kw) >> 1) - ow0;
                        s += T[(py * 10 + px) * 16 + kh * 4 + kw];
                    }
                }
                p.out[(size_t(b) * p.H + h) * p.W + w] = s;
            }
        }
        __syncthreads();
    }
}

inline int grid_for(int64_t total) {
    const int64_t g = (total + 255) / 256;
    return int(g < 8192 ? (g < 1 ? 1 : g) : 8192);
}


// ------------------------------------------------------------------------------------------------ tiled inference
// A scene [B][C][H][W] is cut into tiles of `tile` pixels that overlap by 2 * margin: tile (b, ti, tj) covers scene rows
// ti * core - margin .. + tile - 1 (core = tile - 2 margin), reflected at the scene's borders (torch's 'reflect' padding: the edge
// pixel is not repeated) -- what F.pad(scene, (margin, margin + pw, margin, margin + ph), 'reflect') followed by a stack of slices
// produces, in one launch per batch of tiles.  Tiles are numbered b-major, then ti, then tj; a launch takes tiles first .. first + n - 1.
struct TileP { const float* scene; float* tiles; int B, C, H, W, tile, margin, core, nth, ntw, first, n; };

__global__ __launch_bounds__(256) void tile_gather_kernel(const TileP p) {
    const int64_t per = int64_t(p.C) * p.tile * p.tile, total = int64_t(p.n) * per;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int k = int(i / per);
        int r = int(i - int64_t(k) * per);
        const int c = r / (p.tile * p.tile);
        r -= c * p.tile * p.tile;
        const int y = r / p.tile, x = r - y * p.tile;
        const int id = p.first + k, b = id / (p.nth * p.ntw), t = id - b * p.nth * p.ntw, ti = t / p.ntw, tj = t - ti * p.ntw;
        const int h = ng_reflect(ti * p.core + y - p.margin, p.H), w = ng_reflect(tj * p.core + x - p.margin, p.W);
        p.tiles[i] = p.scene[((int64_t(b) * p.C + c) * p.H + h) * p.W + w];
    }
}

// the tiles' cores back into the scene [B][C][H][W] (C = 1 for the NIR prediction); pixels past the scene (its extent need not be a
// multiple of the core) are dropped
__global__ __launch_bounds__(256) void tile_scatter_kernel(const float* __restrict__ tiles, float* __restrict__ scene, const TileP p) {
    const int64_t per = int64_t(p.C) * p.core * p.core, total = int64_t(p.n) * per;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int k = int(i / per);
        int r = int(i - int64_t(k) * per);
        const int c = r / (p.core * p.core);
        r -= c * p.core * p.core;
        const int y = r / p.core, x = r - y * p.core;
        const int id = p.first + k, b = id / (p.nth * p.ntw), t = id - b * p.nth * p.ntw, ti = t / p.ntw, tj = t - ti * p.ntw;
        const int h = ti * p.core + y, w = tj * p.core + x;
        if (h < p.H && w < p.W)
            scene[((int64_t(b) * p.C + c) * p.H + h) * p.W + w] = tiles[((int64_t(k) * p.C + c) * p.tile + y + p.margin) * p.tile + x + p.margin];
    }
}

}  // namespace

extern "C" int nirgan_nchw_to_halo(const float* src, int B, int Cs, int H, int W, float* dst, int dst_cs, int c0,
                                   int pad1, int pad2, int pad_mode, void* stream) {
    NG_REQUIRE(src && dst && B > 0 && Cs > 0 && H > 0 && W > 0, "nchw_to_halo: bad arguments");
    NG_REQUIRE(c0 >= 0 && c0 + Cs <= dst_cs && pad1 >= 0 && pad2 >= 0, "nchw_to_halo: channel window / pads out of range");
    NG_REQUIRE(pad_mode != NIRGAN_BORDER_REFLECT || (pad1 < H && pad1 < W && pad2 < H + 2 * pad1 && pad2 < W + 2 * pad1), "nchw_to_halo: reflect pad wider than the image");
    const int P = pad1 + pad2;
    const int64_t total = int64_t(B) * (H + 2 * P) * (W + 2 * P);
    hipLaunchKernelGGL(nchw_to_halo_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, B, Cs, H, W, dst, dst_cs, c0, pad1, pad2, pad_mode == NIRGAN_BORDER_REFLECT ? 1 : 0);
    return nirgan_check_launch("nchw_to_halo");
}

extern "C" int nirgan_tap_gather(const nirgan_tap_gather_desc* d, void* stream) {
    NG_REQUIRE(d && d->q && d->dst, "tap_gather: null pointer");
    NG_REQUIRE(d->ntaps >= 1 && d->ntaps <= 64 && d->ntaps <= d->q_cs && d->q_cs % 4 == 0, "tap_gather: ntaps/q_cs out of range");
    NG_REQUIRE(d->B > 0 && d->crop >= 0 && d->OH > 2 * d->crop && d->OW > 2 * d->crop, "tap_gather: bad shape");
    NG_REQUIRE(ng_aligned16(d->q), "tap_gather: q must be 16-byte aligned");
    GatherP p;
    p.q = d->q; p.q_cs = d->q_cs; p.q_row = d->q_wp * d->q_cs; p.q_img = d->q_hp * p.q_row; p.q_hp = d->q_hp; p.q_wp = d->q_wp;
    p.ntaps = d->ntaps;
    int kh = 1, kw = 1;
    for (int t = 0; t < 64; ++t) {
        p.dh[t] = p.dw[t] = 0;
        if (t < d->ntaps) {
            NG_REQUIRE(d->tap_dh[t] >= 0 && d->OH - 1 + d->tap_dh[t] < d->q_hp && d->tap_dw[t] >= 0 && d->OW - 1 + d->tap_dw[t] < d->q_wp, "tap_gather: tap %d out of range", t);
            p.dh[t] = d->tap_dh[t]; p.dw[t] = d->tap_dw[t];
            kh = d->tap_dh[t] + 1 > kh ? d->tap_dh[t] + 1 : kh;
            kw = d->tap_dw[t] + 1 > kw ? d->tap_dw[t] + 1 : kw;
        }
    }
    p.kh = kh; p.kw = kw;
    p.bias = d->bias; p.act = d->act; p.B = d->B; p.OH = d->OH; p.OW = d->OW; p.crop = d->crop; p.dst = d->dst;
    const int H2 = d->OH - 2 * d->crop, W2 = d->OW - 2 * d->crop;
    bool rowmajor = kh == kw && d->ntaps == kh * kw;
    for (int t = 0; rowmajor && t < d->ntaps; ++t) rowmajor = d->tap_dh[t] == t / kw && d->tap_dw[t] == t % kw;
    if (rowmajor && int64_t(H2) * W2 >= 4096) {     // (small maps: the all-planes window kernel below has more blocks)
        const int grid = d->B * ((W2 + 31) / 32) * ((H2 + 31) / 32);
        hipLaunchKernelGGL(tap_gather_rows_kernel, dim3(grid), dim3(256), size_t(32) * (32 + kw - 1) * kw * 4, static_cast<hipStream_t>(stream), p);
        return nirgan_check_launch("tap_gather");
    }
    const size_t lds = size_t(8 + kh - 1) * (32 + kw - 1) * (d->q_cs | 1) * 4;
    NG_REQUIRE(lds <= 160 * 1024, "tap_gather: window does not fit LDS");
    static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(tap_gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }();
    (void)once;
    const int grid = d->B * ((W2 + 31) / 32) * ((H2 + 7) / 8);
    hipLaunchKernelGGL(tap_gather_kernel, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("tap_gather");
}

extern "C" int nirgan_tap_scatter(const nirgan_tap_scatter_desc* d, void* stream) {
    NG_REQUIRE(d && d->dout && d->dq, "tap_scatter: null pointer");
    NG_REQUIRE(d->act != NIRGAN_ACT_TANH || d->out, "tap_scatter: out required for tanh");
    NG_REQUIRE(d->ntaps >= 1 && d->ntaps <= 64 && d->ntaps <= d->q_cs && d->q_cs % 4 == 0, "tap_scatter: ntaps/q_cs out of range");
    NG_REQUIRE(d->B > 0 && d->crop >= 0 && d->OH > 2 * d->crop && d->OW > 2 * d->crop, "tap_scatter: bad shape");
    NG_REQUIRE(ng_aligned16(d->dq), "tap_scatter: dq must be 16-byte aligned");
    ScatterP p;
    p.dout = d->dout; p.out = d->out; p.act = d->act; p.B = d->B; p.OH = d->OH; p.OW = d->OW; p.crop = d->crop;
    p.ntaps = d->ntaps;
    for (int t = 0; t < 64; ++t) { p.dh[t] = t < d->ntaps ? d->tap_dh[t] : 0; p.dw[t] = t < d->ntaps ? d->tap_dw[t] : 0; }
    p.dq = d->dq; p.q_hp = d->q_hp; p.q_wp = d->q_wp; p.q_cs = d->q_cs; p.dbias = d->dbias;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = int64_t(d->B) * d->q_hp * d->q_wp * (d->q_cs / 4);
    hipLaunchKernelGGL(tap_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, st, p);
    if (d->dbias) {
        const int64_t n = int64_t(d->B) * (d->OH - 2 * d->crop) * (d->OW - 2 * d->crop);
        hipLaunchKernelGGL(tap_dbias_kernel, dim3(1), dim3(1024), 0, st, d->dout, d->out, d->act, n, d->dbias);
    }
    return nirgan_check_launch("tap_scatter");
}

extern "C" int nirgan_conv_channel_dgrad(const nirgan_chan_dgrad_desc* d, void* stream) {
    NG_REQUIRE(d && d->dy && d->w && d->out, "conv_channel_dgrad: null pointer");
    NG_REQUIRE(d->C >= 4 && d->C % 4 == 0 && d->k >= 1 && d->k <= 7 && d->stride >= 1 && d->pad >= 0, "conv_channel_dgrad: bad layer");
    NG_REQUIRE(d->channel >= 0 && d->channel < d->cin && d->B > 0 && d->H > 0 && d->W > 0 && d->dy_pad >= 0, "conv_channel_dgrad: bad shape");
    NG_REQUIRE(ng_aligned16(d->dy), "conv_channel_dgrad: dy must be 16-byte aligned");
    const int OH = (d->H + 2 * d->pad - d->k) / d->stride + 1, OW = (d->W + 2 * d->pad - d->k) / d->stride + 1;
    NG_REQUIRE(d->dy_hp == OH + 2 * d->dy_pad && d->dy_wp == OW + 2 * d->dy_pad, "conv_channel_dgrad: dY geometry mismatch");
    const size_t lds = size_t(d->k) * d->k * d->C * 4;
    NG_REQUIRE(lds <= 65536, "conv_channel_dgrad: weights do not fit LDS");
    ChanDgradP p;
    p.dy = d->dy; p.C = d->C; p.dy_row = d->dy_wp * d->C; p.dy_img = d->dy_hp * p.dy_row; p.dy_pad = d->dy_pad; p.OH = OH; p.OW = OW;
    p.w = d->w; p.cin = d->cin; p.k = d->k; p.stride = d->stride; p.pad = d->pad; p.channel = d->channel;
    p.B = d->B; p.H = d->H; p.W = d->W; p.out = d->out;
    if (d->k == 4 && d->stride == 2 && d->pad == 1 && d->H == 2 * OH && d->W == 2 * OW && 16 * d->C * 4 + 1600 * 4 <= 65536) {
        const int64_t ntiles = int64_t(d->B) * ((OH + 7) / 8) * ((OW + 7) / 8);
        hipLaunchKernelGGL(chan_dgrad_k4s2_kernel, dim3(int(ntiles < 8192 ? ntiles : 8192)), dim3(256), size_t(16) * d->C * 4 + 1600 * 4,
                           static_cast<hipStream_t>(stream), p);
        return nirgan_check_launch("conv_channel_dgrad");
    }
    const int64_t npix = int64_t(d->B) * d->H * d->W;
    int64_t g = (npix + 15) / 16;
    g = g < 4096 ? g : 4096;
    hipLaunchKernelGGL(chan_dgrad_kernel, dim3(int(g)), dim3(256), lds, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("conv_channel_dgrad");
}

static int tile_params(TileP& p, int B, int C, int H, int W, int tile, int margin, int first, int n, const char* who) {
    NG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && tile > 0 && margin >= 0 && 2 * margin < tile, "%s: bad shape", who);
    p.B = B; p.C = C; p.H = H; p.W = W; p.tile = tile; p.margin = margin; p.core = tile - 2 * margin;
    p.nth = (H + p.core - 1) / p.core; p.ntw = (W + p.core - 1) / p.core;
    // torch's reflect padding needs the pad narrower than the image: left/top = margin, right/bottom = margin + round-up of the extent
    NG_REQUIRE(margin < H && margin < W && p.nth * p.core - H + margin < H && p.ntw * p.core - W + margin < W, "%s: the reflected border is wider than the scene (%d x %d, tile %d, margin %d)", who, H, W, tile, margin);
    NG_REQUIRE(first >= 0 && n > 0 && int64_t(first) + n <= int64_t(B) * p.nth * p.ntw, "%s: tiles %d .. %d of %lld", who, first, first + n - 1, (long long)B * p.nth * p.ntw);
    p.first = first; p.n = n;
    return NIRGAN_OK;
}

extern "C" int64_t nirgan_tile_count(int B, int H, int W, int tile, int margin) {
    if (B <= 0 || H <= 0 || W <= 0 || tile <= 0 || margin < 0 || 2 * margin >= tile) return 0;
    const int core = tile - 2 * margin;
    return int64_t(B) * ((H + core - 1) / core) * ((W + core - 1) / core);
}

extern "C" int nirgan_tile_gather(const float* scene, int B, int C, int H, int W, int tile, int margin, int first, int n, float* tiles, void* stream) {
    NG_REQUIRE(scene && tiles, "tile_gather: null pointer");
    TileP p;
    const int rc = tile_params(p, B, C, H, W, tile, margin, first, n, "tile_gather");
    if (rc != NIRGAN_OK) return rc;
    p.scene = scene; p.tiles = tiles;
    hipLaunchKernelGGL(tile_gather_kernel, dim3(grid_for(int64_t(n) * C * tile * tile)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("tile_gather");
}

extern "C" int nirgan_tile_scatter(const float* tiles, int B, int C, int H, int W, int tile, int margin, int first, int n, float* scene, void* stream) {
    NG_REQUIRE(scene && tiles, "tile_scatter: null pointer");
    TileP p;
    const int rc = tile_params(p, B, C, H, W, tile, margin, first, n, "tile_scatter");
    if (rc != NIRGAN_OK) return rc;
    p.scene = nullptr; p.tiles = nullptr;
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(grid_for(int64_t(n) * C * p.core * p.core)), dim3(256), 0, static_cast<hipStream_t>(stream), tiles, scene, p);
    return nirgan_check_launch("tile_scatter");
}
