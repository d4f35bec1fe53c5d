/*
 * nirgan_hip.h -- C ABI of libnirgan_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the NIR-GAN Pix2Pix hot path.  The reference implements the path
 * with stock torch.nn layers (Python only, no FFI of its own); the entry points below are
 * what a binding of that path binds instead.  Each one cites the reference call site(s) it
 * replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library never
 *     allocates, frees or retains device memory;
 *   - `stream` is a hipStream_t passed as void*; every launch goes to that stream, no hidden
 *     synchronisation, safe for hipGraph capture;
 *   - activations are fp32 "NHWC with halo": [B][Hp][Wp][cs] floats, cs % 4 == 0, base 16-byte
 *     aligned.  Convolutions are *valid* convolutions over such buffers; the producer of a
 *     buffer writes the halo (reflect copies, or zeros that are set once at allocation);
 *   - weights are consumed in a packed layout [N][ntaps*run] (K contiguous) produced by
 *     nirgan_pack_rows from the reference layouts (Conv2d Cout,Cin,kh,kw; ConvTranspose2d
 *     Cin,Cout,kh,kw; Linear out,in) through an int32 index map;
 *   - every function returns 0 on success, a negative code on error (message from
 *     nirgan_last_error()); descriptors are validated (alignment, extents) before launch.
 */
#ifndef NIRGAN_HIP_H
#define NIRGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIRGAN_MAX_TAPS 16

#define NIRGAN_OK 0
#define NIRGAN_ERR_ARG (-1)      /* invalid descriptor / argument */
#define NIRGAN_ERR_LAUNCH (-2)   /* HIP launch error */

#define NIRGAN_ACT_NONE 0
#define NIRGAN_ACT_RELU 1
#define NIRGAN_ACT_LRELU 2
#define NIRGAN_ACT_TANH 3

#define NIRGAN_BORDER_KEEP 0     /* halo untouched (zero halo set once by the owner) */
#define NIRGAN_BORDER_REFLECT 1  /* halo = reflection of the interior (nn.ReflectionPad2d) */

int nirgan_version(void);
const char* nirgan_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the MFMA pipe (v_mfma_f32_32x32x2_f32, exact fp32).
 *   out[b][oh*out_stride+out_oh][ow*out_stride+out_ow][n] (+bias[n]) =
 *       sum_{t<ntaps} sum_{c<run} in[b][oh*in_stride+in_oh+tap_dh[t]][ow*in_stride+in_ow+tap_dw[t]][c] * w[n][t*run+c]
 * `run` floats are read contiguously from the tap's pixel (run may span several pixels:
 * run = kw*cs packs a kernel row).  One descriptor covers: forward of Conv2d (stride 1/2),
 * data-gradient of Conv2d (flipped weights; stride-2 as 4 sub-pixel phases), forward and
 * data-gradient of ConvTranspose2d (4 phases / strided), the 1x1 "tap-plane" products of the
 * single-output-channel 7x7 / 4x4 layers, and nn.Linear.
 * Replaces: nn.Conv2d / nn.ConvTranspose2d forward+backward at model/networks.py:342,349,
 * 360-363,367,405-427,559,566,574,579; nn.Linear at model/generator_inject.py:90,110.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* in;  int64_t in_elems;   /* input buffer and its size in floats */
    int in_hp, in_wp, in_cs;              /* halo'd height, width, floats per pixel */
    int run;                              /* contiguous floats per tap (multiple of 4) */
    int in_stride, in_oh, in_ow;          /* input step per output pixel, origin */
    int ntaps;
    int tap_dh[NIRGAN_MAX_TAPS], tap_dw[NIRGAN_MAX_TAPS];
    const float* w;   int64_t w_elems;    /* packed weights [N][ntaps*run] */
    const float* bias;                    /* [N] or NULL */
    float* out;       int64_t out_elems;
    int out_hp, out_wp, out_cs;
    int out_stride, out_oh, out_ow;
    int B, OH, OW, N;
    const float* zero_page;               /* >= 64 zero bytes, 16-byte aligned */
    /* optional split-K for problems with few output tiles (< 1 tile per CU slot): the K-steps are divided over
     * `ksplit` blocks per tile, partial tiles go to split_ws [ksplit][B*OH*OW][N] and are summed (fixed order,
     * + bias) into `out` by a second small launch.  ksplit <= 1: off. */
    int ksplit; float* split_ws; int64_t split_ws_elems;
    /* operand precision of the contraction: 0 = fp32 (default; the parity path), 1 = both operands rounded to bf16
     * (round-to-nearest-even) as they enter the matrix pipe, fp32 accumulate (BASELINE.json configs[4] "bf16 MFMA"),
     * 2 = fp32 operands split into two bf16 terms each, three bf16 products per fp32 product (error <= ~2^-16 relative
     * per product), fp32 accumulate.  Buffers stay fp32 in every mode. */
    int precision;
    /* 1: `w` points to bf16 values (same [N][ntaps*run] layout, written by nirgan_pack_rows_bf16; w_elems counts bf16
     * elements).  Only with precision == 1 and run % 8 == 0: the weights are rounded once when packed instead of at every
     * fragment read -- identical values, 25 % fewer operand bytes per K-step. */
    int w_bf16;
    /* 1: `in` points to bf16 values (a producer's out_bf16 / dy_bf16 twin: same halo'd geometry, in_elems counts bf16
     * elements).  Only together with w_bf16 (precision 1, run % 8 == 0, in_cs % 8 == 0): both operands are then read as
     * stored, half the bytes and LDS-DMA pieces per K-step, no conversion in the K loop. */
    int in_bf16;
    /* optional: partial sums for the instance norm that follows (nirgan_instnorm_fwd with stats_chunks / stats_shift).  Every 64 output
     * pixels x N channels held by a wave leave FOUR values per channel in stats_ws[b][stats_chunk0 + chunk][4][N]: a shift k (the
     * chunk's first pixel, WITHOUT the bias), sum (v - k), sum (v - k)^2, and the count 64 -- sums about a value of the data itself carry
     * no cancellation, nirgan_instnorm_fwd re-bases the chunks onto one shift.  chunk = (tile within the sample) * 2 + the wave's row
     * half.  The problem then has OH * OW % 128 == 0 (a tile does not cross samples) and no split-K; it contributes OH * OW / 64 chunks
     * per sample, a launch of several problems over one output (the sub-pixel phases of a transposed convolution) numbers them with
     * stats_chunk0, and stats_chunks is the total per sample (the row stride of stats_ws).  stats_ws >= B * stats_chunks * 4 * N floats. */
    float* stats_ws; int64_t stats_ws_elems; int stats_chunk0, stats_chunks;
    /* optional: the launch writes the gradient wrt the OUTPUT of a convolution + instance-norm (+ReLU / LeakyReLU) layer (it is the data
     * gradient of that layer's consumer); the first pass of that layer's backward (nirgan_instnorm_bwd: sums of g_z = g * act'(z) and of
     * g_z * z, z = (y - mean) * rstd) is then taken in the epilogue, next to the store of g: every 128-pixel tile leaves its sums in
     * fuse_part[b][fuse_chunk0 + tile within the sample][2][N] (fixed order, no atomics) and nirgan_instnorm_bwd runs with
     * sums_chunks = fuse_chunks (its first pass is not launched).  fuse_y: the layer's pre-normalisation output, dense
     * [B][fuse_h][fuse_w][N]; the launch's output pixel (oh, ow) is y pixel (oh * out_stride + fuse_oh, ow * out_stride + fuse_ow).
     * Needs OH * OW % 128 == 0, N % 4 == 0, no split-K, no bias.  fuse_part >= B * fuse_chunks * 2 * N floats. */
    const float* fuse_y; const float* fuse_mean; const float* fuse_rstd;
    int fuse_h, fuse_w, fuse_oh, fuse_ow, fuse_act; float fuse_slope;
    float* fuse_part; int64_t fuse_part_elems; int fuse_chunk0, fuse_chunks;
    /* bf16 operand mode: `out` points to bf16 elements (same geometry, out_elems counts them) -- the convolution's output in front of an
     * instance norm is stored rounded to nearest even while the statistics (stats_ws) come from the fp32 accumulators; the norm's launches
     * read it with y_bf16 = 1.  Needs N % 4 == 0, out_cs % 4 == 0, no split-K.  fuse_y_bf16: fuse_y points to such a tensor. */
    int out_bf16, fuse_y_bf16;
    /* kernel choice where more than one applies (A/B measurements, tests): 0 = default -- problems with both operands stored as bf16,
     * N % 256 == 0, run % 64 == 0 and rounds of 256 x 256 tiles that fill at least 62 % of the CUs run on the 256 x 256 x 64 eight-phase tile (one workgroup of
     * eight waves per CU, LDS-DMA in flight across the barriers), everything else on the 128-row tile; NIRGAN_CONV_TILE128 = always the
     * 128-row tile; NIRGAN_CONV_TILE256 = the 256-wide tile also for exact-fp32 problems.  Results differ by fp32 summation order only. */
    int algo;
    /* precision 3 -- fp32-EQUIVALENT on the bf16 matrix pipe (round 5): every fp32 operand is split into three bf16 terms h + m + l (exact)
     * and a product is contracted as the six bf16 products down to 2^-16 of the leading one (dropped: <= 2^-24 |a b|), fp32 accumulate;
     * buffers and results are those of precision 0; measured against float64 the convolutions sit at 0.7-1.0 x the exact tile's error, the weight
     * gradients at up to 1.65 x (max) / 2.4 x (rms) of it (twice the pixels per split in one fp32 chain at N = 128).  The activations are split inside the kernel (fp32 `in` as in every
     * mode); the packed weights come as three bf16 planes written by nirgan_split3 from `w`: w_x3 = plane h, planes m and l follow
     * w_x3_plane bf16 elements apart.  Problems the split tile does not cover (run % 32, N % 64, split-K, no w_x3) run as precision 0. */
    const void* w_x3; int64_t w_x3_plane;
    /* out_span = 2 (precision 3 only; 0 / 1 = one pixel per row): a GEMM row writes TWO horizontally adjacent output pixels -- N = 2 C
     * columns, column n = channel n % C of pixel (oh out_stride + out_oh, ow out_stride + out_ow + n / C); needs out_cs == C (dense pixels)
     * and out_stride >= 2.  Two sub-pixel phases of one output row of a stride-2 transposed convolution / data gradient with C = 64 become
     * ONE problem of 128-column tiles over the union of their taps (zero weight rows where a phase has no tap): the staged activation
     * rows serve both phases and the launch's two problems (output-row parities) balance over the persistent workgroups.  bias holds N
     * values (the layer's C twice).  The instance-norm partial sums and the fused backward sums keep their per-channel records: a chunk
     * of rows leaves two records (pixel parity 0, then 1) of C columns -- stats_chunk0 + 2 OH OW / 64 <= stats_chunks, stats_ws and
     * fuse_part sized with C; fuse_mean / fuse_rstd / fuse_y are those of the C-channel tensor.
     * Second form, out_cs == N: the caller already describes the output in pixel pairs (out_wp = W / 2, any out_stride; no fuse_y) --
     * the generator's first convolution with two adjacent outputs per GEMM row; out_span then only shapes the statistics' records. */
    int out_span;
} nirgan_conv_desc;
#define NIRGAN_CONV_TILE128 1
#define NIRGAN_CONV_X3_BN64 3   /* precision 3: the 256 x 64 block tile also where 256 x 128 applies (A/B) */
#define NIRGAN_CONV_X3_R4 4     /* precision 3, N % 128 == 0: the four-wave register-fed tile of igemm_x3r.h instead of the eight-wave tile (A/B; same output bits) */
#define NIRGAN_CONV_TILE256 2   /* exact-fp32 problems (N % 256 == 0, run % 32 == 0, >= 128 tiles) on the 256-wide tile too (A/B: within 1 % of the 128-row tile) */

int nirgan_conv_igemm(const nirgan_conv_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM weight gradient (MFMA, split over pixels into deterministic slabs).
 *   slab[s][n][t*run+c] = sum_{m in split s}
 *       p[b][oh+p_oh][ow+p_ow][n] * q[b][oh*q_stride+q_oh+tap_dh[t]][ow*q_stride+q_ow+tap_dw[t]][c]
 * with m = (b,oh,ow) over B*OH*OW.  Conv2d: p = dY (n = cout), q = X (c = cin).
 * ConvTranspose2d: p = X (n = cin), q = dY (c = cout).  nirgan_reduce_rows then sums the
 * slabs into the reference-layout gradient.  Replaces autograd's weight gradient of the same
 * layers as nirgan_conv_igemm.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* p;   int64_t p_elems;
    int p_hp, p_wp, p_cs, p_oh, p_ow;
    const float* q;   int64_t q_elems;
    int q_hp, q_wp, q_cs, q_stride, q_oh, q_ow;
    int run, ntaps;
    int tap_dh[NIRGAN_MAX_TAPS], tap_dw[NIRGAN_MAX_TAPS];
    int B, OH, OW, N;                     /* N: rows of the gradient; round_up(N,4) <= p_cs */
    float* slabs;     int64_t slab_elems; /* [nsplit][N][ntaps*run] */
    int nsplit, rows_per_split;           /* rows_per_split % 32 == 0, nsplit*rows_per_split >= B*OH*OW */
    const float* zero_page;
    int precision;                        /* as in nirgan_conv_desc */
    int nplanes; int64_t p_plane, q_plane;/* > 1: that many independent problems of this geometry in one grid; plane i reads
                                           * p + i*p_plane, q + i*q_plane (floats) and writes slabs [i][nsplit][N][K] (fp32 tile only) */
    int pq_bf16;                          /* 1: p and q point to bf16 twins (same geometry; the producers' out_bf16 / dy_bf16);
                                           * precision 1, N > 64, and N, run, p_cs, q_cs multiples of 8 */
    int algo;                             /* kernel choice where more than one applies (A/B measurements, tests): 0 = default (plane-matrix
                                           * problems walk their units as persistent workgroups); NIRGAN_WGRAD_ONE_UNIT = one unit per workgroup */
} nirgan_wgrad_desc;
#define NIRGAN_WGRAD_ONE_UNIT 1
#define NIRGAN_WGRAD_RING10 3    /* the 256-wide tile with an LDS ring of 10 half-tile slots (160 KB) instead of 8 (128 KB): A/B, no gain measured */
#define NIRGAN_WGRAD_TILE128 2   /* bf16 twins: never the 256 x 256 x 64 eight-phase tile (persistent workgroups, one per CU), which is the
                                  * default for N % 256 == 0, ntaps * run % 256 == 0, OW % 64 == 0 or 64 % OW == 0, rows_per_split % 64 == 0 */

int nirgan_wgrad_igemm(const nirgan_wgrad_desc* d, void* stream);

/* Up to 4 independent descriptors of the same tile width (all N <= 64 or all N > 64) in one launch: the four
 * sub-pixel phases of a stride-2 data gradient or of a ConvTranspose2d forward (model/networks.py:349,360-363). */
int nirgan_conv_igemm_group(const nirgan_conv_desc* const* descs, int n, void* stream);

/* Horizontally fused launch of a data-gradient convolution and the weight gradient of the same layer (both
 * read the same dY): one grid holds the tiles of both, so the partly filled last round of one problem is
 * filled by the other.  Semantics = nirgan_conv_igemm(c) followed by nirgan_wgrad_igemm(w). */
int nirgan_conv_wgrad_pair(const nirgan_conv_desc* c, const nirgan_wgrad_desc* w, void* stream);

/* names of the kernels the three launchers above pick for a descriptor (what a profile of the launch shows; NULL for an invalid
 * descriptor): the 128-row tiles or, in the bf16 operand mode, the 256 x 256 x 64 eight-phase tiles */
const char* nirgan_conv_kernel_name(const nirgan_conv_desc* d);
const char* nirgan_wgrad_kernel_name(const nirgan_wgrad_desc* d);
const char* nirgan_conv_wgrad_pair_kernel_name(const nirgan_conv_desc* c, const nirgan_wgrad_desc* w);

/* dst[n*dst_row_stride + map[k]] (= | +=) sum_s slabs[s][n][k]  for map[k] >= 0 */
int nirgan_reduce_rows(const float* slabs, int nsplit, int N, int K, const int32_t* map,
                       float* dst, int64_t dst_elems, int dst_row_stride, int accumulate, void* stream);
/* The same over a BAND of the slabs' rows: dst[n*dst_row_stride + map[k]] (= | +=) sum_s slabs[s][row0 + n][k] for n < rows (slabs of
 * N rows).  The weight gradient of the generator's first convolution, taken with two adjacent output pixels per GEMM row (its slab rows
 * are (pixel parity, output channel)), folds its two bands into the one weight tensor with two calls, the second accumulating
 * (model/networks.py:342 through autograd). */
int nirgan_reduce_rows_part(const float* slabs, int nsplit, int N, int row0, int rows, int K, const int32_t* map,
                            float* dst, int64_t dst_elems, int dst_row_stride, int accumulate, void* stream);

/* The slab sums of several weight gradients in one launch (a network's layers: launch latency for up to 31 MB each otherwise).
 * jobs_device: njobs x 10 int64 in DEVICE memory: {slabs, dst, map, nsplit, N, K, dst_elems, dst_row_stride | (accumulate ? 1 << 32 : 0),
 * first_block, taps}.  taps = 0: job j owns N_j * ceil(K_j / 256) blocks and scatters through map, the arithmetic and association of
 * nirgan_reduce_rows.  taps = T > 0: the caller asserts map[t * Cin + c] == c * T + t (K = T * Cin, Cin % 64 == 0, T <= 16: the Conv2d
 * weight's own layout [N][Cin][kh][kw], dst_row_stride >= Cin * T): job j owns N_j * Cin / 64 blocks which store 64 * T contiguous
 * floats each (same sums, no 4-byte scatter).  total_blocks = the sum; K % 4 == 0, slabs 16-byte aligned. */
int nirgan_reduce_rows_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream);

/* dst[n][k] = map[k] >= 0 ? src[n*src_row_stride + map[k]] : 0   (weight packing) */
int nirgan_pack_rows(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                     float* dst, int N, int K, void* stream);

/* nirgan_pack_rows with the destination in bf16 (round to nearest even), K % 8 == 0 */
int nirgan_pack_rows_bf16(const float* src, int64_t src_elems, int src_row_stride, const int32_t* map,
                          void* dst_bf16, int N, int K, void* stream);
/* All weight packs of a step in one launch.  jobs_device: njobs x 10 int64 in DEVICE memory:
 * {src, dst, map, src_elems, N, K, src_row_stride | (bf16 destination ? 1 << 32 : 0), first_block, w_x3, w_x3_plane}; job j owns blocks
 * [first_block_j, first_block_j + N_j * ceil(K_j / 1024)); total_blocks = their sum.  w_x3 != 0 (fp32 destination only): the job also
 * writes the three bf16 planes of nirgan_split3 for its packed values (plane stride w_x3_plane bf16 elements). */
int nirgan_pack_rows_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream);

/* precision 3 (nirgan_conv_desc.w_x3): n fp32 values -> three bf16 planes dst[0..n), dst[plane..), dst[2 plane..) with
 * src = h + m + l exactly, each term the round-to-nearest-even bf16 of what the previous ones left.  n % 8 == 0, plane >= n, plane % 8 == 0.
 * Runs once per optimizer step on the packed weights of the layers that take the split tile (model/networks.py:349,360-363,559-574). */
int nirgan_split3(const float* src, void* dst_bf16, int64_t n, int64_t plane, void* stream);

/* ---------------------------------------------------------------------------------------
 * InstanceNorm2d(affine=False, eps) + activation + residual + halo write, forward.
 *   z = norm ? (y - mean_bc) * rstd_bc : y ;  a = act(z) + residual ;  out interior = a,
 *   reflect halo of width o_pad written when border == NIRGAN_BORDER_REFLECT.
 * Biased variance over H*W per (b,c); mean/rstd saved for backward.
 * Replaces: nn.InstanceNorm2d + nn.ReLU / nn.LeakyReLU(0.2) / residual add / the
 * nn.ReflectionPad2d of the next layer, model/networks.py:30,343-344,350-351,364-365,
 * 405-433,559,567-568,575-576.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* y;                       /* [B][H][W][C] dense */
    int B, H, W, C;
    int norm; float eps;
    float* mean; float* rstd;             /* [B][C] */
    int act; float slope;
    const float* residual; int r_hp, r_wp, r_pad;   /* optional [B][r_hp][r_wp][C], interior at r_pad */
    float* out; int o_hp, o_wp, o_pad; int border;   /* out = NULL and out_bf16 = NULL (with norm): statistics only; out = NULL with
                                                      * out_bf16: only the twin is stored (every consumer reads bf16) */
    float* ws; int64_t ws_elems;          /* >= B * nchunk * 2 * C floats, see nirgan_instnorm_ws_elems */
    void* out_bf16;                       /* optional twin of `out` (same geometry, bf16 elements): every store is mirrored, rounded
                                           * to nearest even -- the operand the bf16 mode's convolutions read (in_bf16) */
    int stats_chunks;                     /* > 0 (with norm): the producer of y already left per-chunk partial sums in ws as
                                           * [B][stats_chunks][4][C] = {k, sum (v - k), sum (v - k)^2, count} with v = y - stats_shift and k a
                                           * value of the chunk itself (nirgan_wino6_output / nirgan_conv_igemm with stats_ws): the pass
                                           * over y that would form them is skipped, the chunks are re-based onto one shift in a fixed order */
    const float* stats_shift;             /* [C], what the producer left out of v (the convolution's bias); NULL = 0 */
    int y_bf16;                           /* 1: y points to bf16 elements (a convolution launch with out_bf16) */
} nirgan_in_fwd_desc;

int64_t nirgan_instnorm_ws_elems(int B, int H, int W, int C);
int nirgan_instnorm_fwd(const nirgan_in_fwd_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Backward of the same block.  Incoming gradient g_a = gather(g) (+ g2):
 *   g is a halo'd buffer [B][g_hp][g_wp][C]; g_fold = 1 folds a reflect halo of width g_pad
 *   back onto the interior (adjoint of ReflectionPad2d), g_fold = 0 reads the interior only;
 *   g2 is an optional dense [B][H][W][C] term (skip connection).
 *   g_z = g_a * act'(z), z = norm ? (y - mean)*rstd : y;  dy = norm ? rstd*(g_z - mean(g_z) - z*mean(g_z*z)) : g_z.
 * dy is written to the interior (d_pad) of a zero-halo buffer that feeds the data- and
 * weight-gradient GEMMs; gsum_out (optional, dense) receives g_a; dbias (optional, [C])
 * accumulates sum dy.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* g; int g_hp, g_wp, g_pad, g_fold;
    const float* g2;
    const float* a; int a_hp, a_wp, a_pad;   /* unused (kept for ABI stability): the mask is the sign of z recomputed from y */
    int act; float slope;
    const float* y; const float* mean; const float* rstd; int norm;
    int B, H, W, C;
    float* dy; int d_hp, d_wp, d_pad;     /* dy = NULL (with norm): the two reductions only, their means stay in ws for a consumer that
                                             evaluates dy on the fly (nirgan_wino6_input_dy_norm) */
    float* gsum_out;
    float* dbias;
    float* ws; int64_t ws_elems;
    void* dy_bf16;                        /* optional twin of `dy` (same geometry, bf16), as out_bf16; with norm and dy = NULL only the twin is
                                           * stored (dy = NULL and dy_bf16 = NULL: the two reductions only) */
    int sums_chunks;                      /* > 0 (with norm): the first pass is done -- the producer of the gradient left the partial sums of g_z
                                             and g_z * z in ws as [B][sums_chunks][2][C].  Either nirgan_wino6_output with fuse_gz (the folded
                                             gradient g_a then sits in gsum_out; g / g2 are not read), or a convolution launch with
                                             nirgan_conv_desc.fuse_* (gsum_out NULL: the second pass reads g itself, which then has no fold
                                             and no g2).  ws >= B * sums_chunks * 2 * C + B * 2 * C floats */
    int y_bf16;                           /* 1: y points to bf16 elements (as nirgan_in_fwd_desc.y_bf16) */
    int g_bf16;                           /* 1: g points to bf16 elements (same halo'd geometry): the data-gradient launch that produced it stored
                                           * it with nirgan_conv_desc.out_bf16 (bf16 operand mode; g2 and gsum_out stay fp32) */
} nirgan_in_bwd_desc;

int nirgan_instnorm_bwd(const nirgan_in_bwd_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Layout / halo helpers at the NCHW boundary.
 * ------------------------------------------------------------------------------------- */
/* dst[b][hh][ww][c0+c] = src[b][c][rh(hh)][rw(ww)], c < Cs.  pad_mode REFLECT: the halo of
 * width pad1+pad2 is the composition reflect(pad1) then reflect(pad2) (Px2Px_PL.forward's
 * F.pad(...,'reflect') model/pix2pix.py:91-93 followed by ReflectionPad2d(3) networks.py:341).
 * pad_mode KEEP: only the interior (offset pad1+pad2) is written (torch.cat + zero padding,
 * model/pix2pix.py:197,202,216 + networks.py:559). */
int nirgan_nchw_to_halo(const float* src, int B, int Cs, int H, int W,
                        float* dst, int dst_cs, int c0, int pad1, int pad2, int pad_mode, void* stream);

/* Data gradient of a Conv2d wrt ONE input channel (the generator step only needs dD/dpred, channel 3 of
 * cat(rgb, pred): model/pix2pix.py:216-221 through networks.py:559):
 *   out[b][h][w] = sum_{kh,kw,co} dY[b][(h+pad-kh)/stride][(w+pad-kw)/stride][co] * W[co][channel][kh][kw]
 * dY is a halo'd NHWC buffer [B][OH+2dy_pad][OW+2dy_pad][C]; W is in the reference layout [C][cin][k][k]. */
typedef struct {
    const float* dy; int dy_hp, dy_wp, dy_pad, C;
    const float* w; int cin, k, stride, pad, channel;
    int B, H, W;
    float* out;                           /* [B][H][W] */
} nirgan_chan_dgrad_desc;
int nirgan_conv_channel_dgrad(const nirgan_chan_dgrad_desc* d, void* stream);

/* Single-output-channel convolution tail: out[b][oh][ow] = act(bias + sum_t Q[b][oh+tap_dh[t]][ow+tap_dw[t]][t])
 * restricted to the crop window (crop pixels removed on every side); dst is NCHW [B][1][OH-2crop][OW-2crop].
 * With the 1x1 tap-plane product this is Conv2d(C,1,k) (+Tanh, + crop of pix2pix.py:107-108):
 * model/networks.py:367-368, :579. */
typedef struct {
    const float* q; int q_hp, q_wp, q_cs;
    int ntaps; int tap_dh[64], tap_dw[64];
    const float* bias;                    /* 1 float on device, or NULL */
    int act;
    int B, OH, OW, crop;
    float* dst;                           /* [B][OH-2crop][OW-2crop] */
} nirgan_tap_gather_desc;
int nirgan_tap_gather(const nirgan_tap_gather_desc* d, void* stream);

/* Adjoint: dq[b][hh][ww][t] = dz[b][hh-tap_dh[t]][ww-tap_dw[t]], dz = dout*act'(out) inside
 * the crop window and 0 elsewhere; channels t >= ntaps of dq are zeroed; dbias (optional)
 * accumulates sum dz. */
typedef struct {
    const float* dout; const float* out;  /* [B][OH-2crop][OW-2crop] each; out may be NULL when act == NONE */
    int act;
    int B, OH, OW, crop;
    int ntaps; int tap_dh[64], tap_dw[64];
    float* dq; int q_hp, q_wp, q_cs;
    float* dbias;
} nirgan_tap_scatter_desc;
int nirgan_tap_scatter(const nirgan_tap_scatter_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * The generator's last layer as direct kernels: Conv2d(64, 1, 7) + bias + tanh (+ crop) over the
 * reflect-padded halo'd NHWC input, and its backward (model/networks.py:366-368 -- ReflectionPad2d(3),
 * Conv2d(ngf, output_nc, kernel_size=7, padding=0), Tanh; the crop is model/pix2pix.py:101-110).
 * A wave's 64 lanes are the 64 input channels; C must be 64 and k 7 (other widths keep the
 * tap-plane route above).  Same results as nirgan_conv_igemm + nirgan_tap_gather /
 * nirgan_tap_scatter + nirgan_wgrad_igemm + nirgan_conv_igemm up to fp32 summation order.
 *   forward : out[b][y][x] = act(bias + sum_{ka,kb,c} x[b][y+crop+ka][x+crop+kb][c] * w[ka*7+kb][c])
 *   dz      : zero-bordered image of dout * act'(out) (workspace shared by the two gradients);
 *             gbias (optional) accumulates sum dz
 *   dgrad   : gx over the whole halo'd grid [B][x_hp][x_wp][64]
 *   wgrad   : gw[c*49 + t] (the Conv2d weight's own [1][64][7][7] layout), overwritten; fixed
 *             summation order (bitwise reproducible)
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* x; int x_hp, x_wp;       /* [B][x_hp][x_wp][C], x_hp = OH + k - 1 */
    int B, OH, OW, crop, C, k;
    const float* w;                       /* [k*k][C] tap-major (the tap-plane forward pack) */
    const float* bias; int act;
    float* out;                           /* [B][OH-2crop][OW-2crop] */
    const float* dout;                    /* gradient wrt out */
    float* dz; int64_t dz_elems;          /* >= nirgan_endconv_dz_elems(B, OH, OW) floats, 16-byte aligned */
    float* gx;
    float* gw; float* gbias;
    float* ws; int64_t ws_elems;          /* >= nirgan_endconv_ws_elems(B, OH, OW) floats */
} nirgan_endconv_desc;
int64_t nirgan_endconv_dz_elems(int B, int OH, int OW);
int64_t nirgan_endconv_ws_elems(int B, int OH, int OW);
int nirgan_endconv_fwd(const nirgan_endconv_desc* d, void* stream);
int nirgan_endconv_dz(const nirgan_endconv_desc* d, void* stream);
int nirgan_endconv_dgrad(const nirgan_endconv_desc* d, void* stream);
int nirgan_endconv_wgrad(const nirgan_endconv_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Losses (forward value + gradient in one pass).
 * ------------------------------------------------------------------------------------- */
/* LSGAN: loss_out[0] += weight * mean((pred - target)^2); grad = weight * 2 (pred-target)/n.
 * GANLoss('lsgan') with MSELoss, model/networks.py:233,258-276.  One workgroup (patch maps are a few 10^4 values): the sum is
 * bitwise reproducible. */
int nirgan_lsgan(const float* pred, int64_t n, float target, float weight,
                 float* loss_out, float* grad, void* stream);

/* L1 + spectral-index losses on NCHW tiles (rgb [B][3][H][W], nir/pred [B][1][H][W]).
 *   sums[0..6] += sum|pred-nir|, then the per-index sums of criterion(idx(nir), idx(pred))
 *   for ndvi, ndwi, gndvi, savi, msavi, evi (all divided by n by the caller);
 *   grad_pred = extra_scale*extra[...][extra_c] + w_l1*sign(pred-nir)/n + sum_i w_i * d crit_i / d pred
 * criterion 0 = l1, 1 = l2.  torch.nn.L1Loss model/pix2pix.py:60,222;
 * RemoteSensingIndices utils/remote_sensing_indices.py:23-71,84-319. */
typedef struct {
    const float* rgb;                     /* [B][3][H][W]; NULL = plain L1 on (pred, nir): every index weight 0, log_all 0 */
    const float* nir; const float* pred;
    int B, H, W;
    float w_l1, w_ndvi, w_ndwi, w_gndvi, w_savi, w_msavi, w_evi;  /* already multiplied by lambda_rs */
    int criterion;
    int log_all;                          /* 1: evaluate all six indices (logging_dict); 0: only those with weight != 0 */
    const float* extra; int extra_cs, extra_c; float extra_scale;  /* optional NHWC gradient term */
    float* sums;                          /* 7 floats, accumulated */
    float* grad_pred;                     /* [B][1][H][W] or NULL (forward only) */
    float* ws; int64_t ws_elems;          /* >= NIRGAN_PIX_LOSS_WS_ELEMS floats: per-block partial sums, added up in block order
                                             (no float atomics: the seven sums are bitwise reproducible); not shared between streams */
} nirgan_pix_loss_desc;
#define NIRGAN_PIX_LOSS_WS_ELEMS 8192
int nirgan_pix_loss(const nirgan_pix_loss_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam, amsgrad=False, weight_decay=0) on a flat fp32 range.
 * model/pix2pix.py:486-487.  `step` is the 1-based count after increment.
 * ------------------------------------------------------------------------------------- */
int nirgan_adam(float* p, const float* g, float* m, float* v, int64_t n,
                float lr, float beta1, float beta2, float eps, int step, void* stream);

/* -------------------------------------------------------------------------------------
 * Image-quality metrics of the train / validation loop, one pass on the device instead of the
 * reference's `.cpu()` round trip every 10th batch (model/pix2pix.py:183-186, :281):
 * utils/calculate_metrics.py:5-36 = F.l1_loss, F.mse_loss, kornia.metrics.psnr(pred, target, 1.0),
 * kornia.metrics.ssim(pred, target, window_size=5, max_val=1.).mean(); utils/losses.py:11-30 ssim_loss (window 11).
 * means[0] = mean |pred - target|, means[1] = mean (pred - target)^2, means[2] = mean of the SSIM map
 * (Gaussian window `window` x `window`, sigma, reflect border, output size = input size:
 *   ssim = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2) + eps), C1 = (0.01 max_val)^2, C2 = (0.03 max_val)^2);
 * PSNR = 10 log10(max_val^2 / means[1]) is left to the caller.  pred/target: `planes` = B*C dense H x W images.
 * ws: nirgan_image_metrics_ws_elems(planes, H, W) floats.  Deterministic (fixed-order partial sums).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* pred; const float* target;
    int planes, H, W;
    int window; float sigma, max_val, eps;   /* kornia: sigma 1.5, eps 1e-12; window odd, <= 11, H, W > window/2 */
    float* ws; int64_t ws_elems;
    float* means;                            /* 3 floats on device */
} nirgan_metrics_desc;

int64_t nirgan_image_metrics_ws_elems(int planes, int H, int W);
int nirgan_image_metrics(const nirgan_metrics_desc* d, void* stream);

/* -------------------------------------------------------------------------------------
 * SSIM term of the generator objective, value AND gradient: model/pix2pix.py:233-237 adds
 * lambda_ssim * ssim_loss(pred, nir); utils/losses.py:10-30: 1 - kornia.metrics.ssim(img1, img2, window_size=11).mean()
 * (Gaussian window sigma 1.5, reflect border, max_val 1, eps 1e-12 in the denominator).
 *   *loss      += weight * (1 - mean SSIM)                (atomic add; NULL = skip)
 *   *value      = 1 - mean SSIM                            (NULL = skip)
 *   grad_pred  += weight * d(1 - mean SSIM) / d pred       (dense [planes][H][W]; NULL = value only)
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* pred; const float* target;  /* dense [planes][H][W] */
    int planes, H, W;
    int window; float sigma, max_val, eps;
    float weight;
    float* ws; int64_t ws_elems;             /* nirgan_ssim_loss_ws_elems(planes, H, W, window) floats */
    float* loss; float* value; float* grad_pred;
} nirgan_ssim_loss_desc;

int64_t nirgan_ssim_loss_ws_elems(int planes, int H, int W, int window);
int nirgan_ssim_loss(const nirgan_ssim_loss_desc* d, void* stream);

/* emd_loss of utils/losses.py:64-78, value and gradient wrt pred: mean | cumsum(softmax(pred.reshape(B,-1),1),1) -
 * cumsum(softmax(target.reshape(B,-1),1),1) | over the B*N elements; scans in double (as torch's CPU cumsum accumulates).
 * *loss += weight * emd (atomic; NULL = skip), *value = emd (NULL = skip), grad_pred += weight * d emd / d pred (NULL = value only). */
typedef struct {
    const float* pred; const float* target;  /* dense [B][N] */
    int B; int64_t N;                        /* N = C*H*W < 2^24 */
    float weight;
    void* ws; int64_t ws_bytes;              /* nirgan_emd_loss_ws_bytes(B, N, grad_pred != NULL), 8-byte aligned */
    float* loss; float* value; float* grad_pred;
} nirgan_emd_loss_desc;

int64_t nirgan_emd_loss_ws_bytes(int B, int64_t N, int with_grad);
int nirgan_emd_loss(const nirgan_emd_loss_desc* d, void* stream);

/* -------------------------------------------------------------------------------------
 * SatCLIP location encoder, fp64 like the reference (model/satclip/load_lightweight.py:29,
 * satclip_wrapper.py:30-35; called from model/pix2pix.py:481-484 once per batch):
 * positional_encoding/spherical_harmonics.py:26-42 + spherical_harmonics_closed_form.py:8-40 (real spherical
 * harmonics of degree < L at phi = deg2rad(lon + 180), theta = deg2rad(lat + 90); feature (l, m) at index l*l+l+m),
 * then SirenNet (location_encoder.py:73-151): x <- sin(w0_i * (W_i x + b_i)) for every layer with w0_i != 0,
 * x <- W_i x + b_i where w0_i == 0 (the last layer's Identity).  Evaluation mode (dropout off), forward only.
 * sh_norm[l*l+l+m] = [sqrt 2 if m != 0] * sqrt((2l+1)(l-|m|)!/(4 pi (l+|m|)!)) is supplied by the host.
 * weights[i]: [dims[i+1]][dims[i]] row-major (nn.Linear layout), biases[i] may be NULL; all device pointers,
 * the pointer tables themselves live on the host.  features (optional): [B][L*L] copy of the harmonics.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const double* lonlat; int B;             /* [B][2] = (lon, lat) in degrees */
    int L; const double* sh_norm;            /* legendre_polys; [L*L] on device */
    int nlayers;                             /* linear layers incl. the last one, <= 8 */
    const double* const* weights; const double* const* biases;
    const int* dims;                         /* host: nlayers+1 widths, dims[0] = L*L, each <= 2048 */
    const double* w0;                        /* host: nlayers sine frequencies (30, 1, ..., 0) */
    double* out;                             /* [B][dims[nlayers]] */
    double* features;                        /* optional */
} nirgan_locenc_desc;

int nirgan_location_encoder(const nirgan_locenc_desc* d, void* stream);

/* -------------------------------------------------------------------------------------
 * Histogram matching of predicted tiles to a reference band: create_synthetic_dataset.py:34-47
 * `match_histograms(img_np, ref_np, channel_axis=None)` per tile (scikit-image, float path
 * `_match_cumulative_cdf`): out = interp(cumsum(src_counts)/n, cumsum(tmpl_counts)/n, tmpl_values)[src_lookup]
 * with (values, counts) = unique(...) of each plane.  image/reference/out: [B][N] dense planes of equal size
 * (the script resizes the reference to the tile first).  ws: nirgan_hist_match_ws_bytes(B, N) bytes, 8-byte aligned.
 * Finite inputs (no NaN); -0.0 and +0.0 are one value, as in numpy.  Deterministic.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* image; const float* reference;
    int B, N;
    void* ws; int64_t ws_bytes;
    float* out;
} nirgan_hist_match_desc;

int64_t nirgan_hist_match_ws_bytes(int B, int N);
int nirgan_hist_match(const nirgan_hist_match_desc* d, void* stream);

/* Output-gradient side of a Winograd layer's backward (used by nirgan_wino6_dy / nirgan_wino6_input_dy): Yt = A dY A^T per tile. */
typedef struct {
    const float* dy; int dy_hp, dy_wp, dy_pad;   /* halo'd [B][H+2pad][W+2pad][K] */
    int B, H, W, K;
    float* Yt; int64_t Yt_elems;                 /* [planes][tiles][K] */
    int r;                                       /* variant code of the layer (nirgan_wino6_desc.r) */
} nirgan_wino_dy_desc;

/* -------------------------------------------------------------------------------------
 * Winograd F(4x4, 3x3) for the same layers (nn.Conv2d(C, K, 3, stride 1) over a halo of 1, model/networks.py:405-427): 36 products
 * per 4x4 outputs (F(2x2,3x3): 64, direct: 144).  The 16 outputs of a tile do not fit the register file next to the product, so
 * the stages are separate launches and the transform-domain product M goes through HBM once:
 *   nirgan_wino6_input   V[f][t][c] = (B^T d B)[f]   6x6 patches at stride 4 of x; T = B*ceil(H/4)*ceil(W/4) tiles, f = 6 f1 + f2
 *   nirgan_wino6_gemm    M[f][t][k] = sum_c V[f][t][c] U[f][k][c]   36 GEMMs in one grid on the direct 128x128 MFMA tile
 *   nirgan_wino6_output  y[b][4ty+i][4tx+j][k] = (A^T M A)[i][j] + bias[k]
 * Cook-Toom points 0, 1, -1, 2, -1/2, inf (matrices in csrc/wino6.hip); exact fp32 products, the result differs from the direct
 * contraction by fp32 rounding only (measured 4e-6 of the output's maximum).  U = nirgan_wino6_weights(W) is [36][K][C].
 * C % 4 == 0, K % 4 == 0, K > 64; extents that are no multiple of 4 cost one partly used tile row / column.
 * Data gradient: the same calls with x = dY (zero halo 2), transpose_flip weights, H x W = the padded input extent.
 * Weight gradient: dU[f][k][c] = sum_t Yt[f][t][k] V[f][t][c] with Yt = A dY A^T (nirgan_wino6_dy / nirgan_wino6_input_dy) and V of
 * the forward input -- ONE nirgan_wgrad_igemm launch with nplanes = 36 -- then nirgan_wino6_wgrad_finish: dW = G^T (sum of the
 * split slabs, in order) G in the reference layout [K][C][3][3].
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const float* x; int x_hp, x_wp;       /* [B][H+2][W+2][C] */
    int B, H, W, C, K;
    const float* U; const float* bias;    /* [36][K][C]; [K] or NULL */
    float* V; int64_t V_elems;            /* [36][T][C] */
    float* M; int64_t M_elems;            /* [36][T][K] */
    float* y;                             /* dense [B][H][W][K] */
    const float* zero_page;
    int r;                                /* filter size: 0 or 3 = F(4x4,3x3) above; 4 = F(4x4,4x4): nn.Conv2d(C, K, 4, stride 1) of the PatchGAN
                                             (model/networks.py:573-579): x is [B][H+3][W+3][C] for H x W outputs, 7x7 patches, 49 planes in
                                             U / V / M / Yt; Cook-Toom over 0, 1, -1, 2, -2, 1/2, inf; fp32 error 1e-5 of the output's maximum;
                                             6 = F(6x6,3x3): 3x3 filters as above with 6x6 outputs per tile, 8x8 patches at stride 6, 64 planes,
                                             T = B * ceil(H/6) * ceil(W/6) (nirgan_wino6_tiles_r); Cook-Toom over 0, 1, -1, 2, -2, 1/2, -1/2, inf;
                                             fp32 error 1.7e-5 of the output's maximum.  The same code goes into nirgan_wino_dy_desc.r,
                                             nirgan_wino6_weights_r and nirgan_wino6_wgrad_finish_r for the layer */
    float* stats_ws; int64_t stats_ws_elems;   /* optional (nirgan_wino6_output): per-tile partial sums of the output for the instance norm that
                                             follows, [B][tiles per image][4][K] = T * 4 * K floats: {k, sum (o - k), sum (o - k)^2, count} over the
                                             tile's stored outputs o WITHOUT the bias, k = the tile's first output; feed nirgan_instnorm_fwd with
                                             ws = stats_ws, stats_chunks = tiles per image, stats_shift = bias */
    /* optional (nirgan_wino6_output of a DATA GRADIENT over the padded extent, 3x3 filters): the first pass of the instance-norm backward
       of the layer that consumes this gradient, in the same kernel.  H x W here is the padded extent (interior (H-2) x (W-2), reflect halo
       of 1): the tile's outputs are folded onto the interior in registers (adjoint of ReflectionPad2d(1); the far halo line and its
       partner must fall in one tile: (H-3) / m == (H-1) / m for tile size m, same for W), fuse_g2 (optional, dense) is added, the
       result g_a goes to fuse_gz (dense [B][H-2][W-2][K]) INSTEAD of y (which is not written and may be NULL), and the tile's partial sums
       of g_z = g_a * act'(z) and g_z * z, z = (fuse_y - mean) * rstd, go to fuse_part as [B][tiles per image][2][K].  Feed
       nirgan_instnorm_bwd with gsum_out = fuse_gz, ws = fuse_part, sums_chunks = tiles per image. */
    const float* fuse_y; const float* fuse_mean; const float* fuse_rstd; const float* fuse_g2;
    float* fuse_gz; float* fuse_part; int64_t fuse_part_elems; int fuse_act; float fuse_slope;
    int algo;                             /* plane-GEMM kernel choice (nirgan_wino6_gemm, nirgan_wino6_gemm_wgrad_pair) where more than one
                                             applies: 0 = default (persistent workgroups on 32-k stages for C = 256 / 512);
                                             NIRGAN_W6_ONE_TILE = one tile per workgroup (16-k stages); NIRGAN_W6_PERSIST16 = persistent workgroups
                                             on 16-k stages; NIRGAN_W6_DIRECT_TILE = the direct convolution tile (32-k stages, one tile per
                                             workgroup).  All compute the same products; kept for A/B measurements and the kernel tests. */
    const void* U3;                       /* optional (precision 3, as nirgan_conv_desc.w_x3): U as three bf16 planes h, m, l, each [planes][K][C],
                                             written by nirgan_wino6_weights_x3 next to U.  With it (and C % 32 == 0, K % 64 == 0) the plane
                                             GEMMs run on the bf16 matrix pipe as six bf16 products per fp32 product, V split inside the
                                             kernel: fp32-equivalent results.  U itself may then be NULL (nirgan_wino6_weights_x3 with U = NULL
                                             writes the planes only: 6 instead of 10 bytes per transformed weight); a launch the split tile
                                             does not take fails without U.  nirgan_wino6_gemm_wgrad_pair then runs its two halves as two
                                             launches (the weight-gradient planes take nirgan_wgrad_desc.precision = 3 on their own). */
} nirgan_wino6_desc;
#define NIRGAN_W6_ONE_TILE 1
#define NIRGAN_W6_PERSIST16 2
#define NIRGAN_W6_DIRECT_TILE 3
#define NIRGAN_W6_TILE256 4                 /* nirgan_wino6_gemm: the exact-fp32 256 x 256 eight-phase tile as persistent workgroups (A/B; K % 256 == 0, C % 32 == 0) */
#define NIRGAN_W6_X3_R4 5                   /* nirgan_wino6_gemm with U3, K % 128 == 0: the plane GEMMs on the four-wave register-fed split tile (A/B; same bits) */
#define NIRGAN_W6_PATCH_PER_THREAD 16      /* nirgan_wino6_input*: F(6x6,3x3) patches one per thread (A/B; default for the plain / dY transforms: a wave per patch x 32 channels) */
#define NIRGAN_W6_PATCH_PER_LANES 17       /* ... and the lane-spread form also for the normalising variant (default there: one per thread) */
/* names of the kernels the two launchers above pick for a descriptor (what a profile of the launch shows) */
const char* nirgan_wino6_gemm_kernel_name(const nirgan_wino6_desc* d);
const char* nirgan_wino6_pair_kernel_name(const nirgan_wino6_desc* d, const nirgan_wgrad_desc* w);

int64_t nirgan_wino6_tiles(int B, int H, int W);                   /* T of the 4x4-output variants */
int64_t nirgan_wino6_tiles_r(int B, int H, int W, int r);          /* T of variant r (0 / 3, 4, 6); 0 for an unknown variant */
int nirgan_wino6_weights(const float* w, int K, int C, int transpose_flip, float* U, void* stream);   /* transpose_flip = 0: U of the forward filter, w = [K][C][3][3]; 1: U of the DATA GRADIENT's filter g'[k][c][i][j] = w[c][k][2-i][2-j] with w = [C][K][3][3] the forward weight */
int nirgan_wino6_weights_r(const float* w, int K, int C, int r, int transpose_flip, float* U, void* stream);   /* w = [K][C][r][r], U = [(r+3)^2][K][C] */
/* nirgan_wino6_weights_r that also leaves U as three bf16 planes (nirgan_split3's rule; U3 = 3 x (r+3)^2 x K x C bf16 elements); U may be
 * NULL: the planes only */
int nirgan_wino6_weights_x3(const float* w, int K, int C, int r, int transpose_flip, float* U, void* U3_bf16, void* stream);
/* all weight transforms of a step in one launch: njobs x 8 int64 {w, U, K, C, transpose_flip, first_block, r, U3 (or 0)} in device memory,
 * first_block = running sum of ceil(K*C/256) */
int nirgan_wino6_weights_batch(const int64_t* jobs_device, int njobs, int total_blocks, void* stream);
int nirgan_wino6_input(const nirgan_wino6_desc* d, void* stream);
/* V straight from a convolution's raw output y (dense [B][H][W][C], d->x unused): x = act((y - mean) * rstd) under a REFLECT halo of 1,
 * evaluated on the fly: nirgan_instnorm_fwd with out = NULL (statistics only) then this call replace the apply pass + the transform */
int nirgan_wino6_input_norm(const nirgan_wino6_desc* d, const float* y, const float* mean, const float* rstd, int act, float slope, void* stream);
int nirgan_wino6_gemm(const nirgan_wino6_desc* d, void* stream);
/* the data gradient's plane GEMMs (c: x = dY, transpose_flip weights; its V written by nirgan_wino6_input_dy) and the layer's 36
 * transform-domain weight-gradient problems (w: nplanes = 36) in ONE grid, like nirgan_conv_wgrad_pair: the long weight-gradient
 * blocks first, the GEMM blocks pack behind them.  Semantics = nirgan_wino6_gemm(c) followed by nirgan_wgrad_igemm(w). */
int nirgan_wino6_gemm_wgrad_pair(const nirgan_wino6_desc* c, const nirgan_wgrad_desc* w, void* stream);
int nirgan_wino6_output(const nirgan_wino6_desc* d, void* stream);
int nirgan_wino6_conv3x3(const nirgan_wino6_desc* d, void* stream);   /* input + gemm + output */
/* Yt [(r+3)^2][B*ceil(H/4)*ceil(W/4)][K]; the descriptor's r field selects the filter size (0 / 3 or 4) */
int nirgan_wino6_dy(const nirgan_wino_dy_desc* d, void* stream);
/* nirgan_wino6_input(c) and nirgan_wino6_dy(y) of the SAME output-gradient buffer (c->x == y->dy, zero halo 2) in one pass: the 4x4
 * block of tile (ty, tx) is the lower-right corner of data-gradient patch (ty, tx) */
int nirgan_wino6_input_dy(const nirgan_wino6_desc* c, const nirgan_wino_dy_desc* y, void* stream);
/* the same pass with dY NOT read from memory but evaluated on the fly as the instance-norm backward's result (F(6x6,3x3), C % 32 == 0): n
 * describes the block (g / g2 / gsum_out, y, mean, rstd, act, ws as given to nirgan_instnorm_bwd with dy = NULL, which leaves the two
 * reduction passes' means in ws); every patch element is rstd * (g_z - mean(g_z) - z * mean(g_z * z)), bitwise what the second pass would
 * have stored into c->x -- that buffer is neither written nor read (c->x / y->dy only describe its geometry: zero halo 2) */
int nirgan_wino6_input_dy_norm(const nirgan_wino6_desc* c, const nirgan_wino_dy_desc* y, const nirgan_in_bwd_desc* n, void* stream);
int nirgan_wino6_wgrad_finish(const float* slabs, int nsplit, int K, int C, float* grad, int accumulate, void* stream);
int nirgan_wino6_wgrad_finish_r(const float* slabs, int nsplit, int K, int C, int r, float* grad, int accumulate, void* stream);   /* [K][C][r][r] */
/* the same for n <= 16 layers of one geometry in ONE grid (host arrays of device pointers; every layer then keeps its own slabs until the
 * call): a single layer's finish is launch latency for 33 MB, twelve of them in one launch run at the memory rate */
int nirgan_wino6_wgrad_finish_batch(const float* const* slabs, float* const* grads, int n, int nsplit, int K, int C, int r, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * SatCLIP injection (model/generator_inject.py:110-127).
 * ------------------------------------------------------------------------------------- */
/* F.interpolate(e.view(B,1,S,S), size=(OH,OW), mode='bilinear', align_corners=False) */
int nirgan_bilinear_fwd(const float* src, int B, int SH, int SW, float* dst, int OH, int OW, void* stream);
int nirgan_bilinear_bwd(const float* ddst, int B, int OH, int OW, float* dsrc, int SH, int SW, void* stream);

/* a = relu(z * (1 + s*e)) ('multiply') or relu(z + s*e) ('add'); z dense [B][H][W][C], e [B][H][W],
 * a written to the interior of a halo'd buffer. */
typedef struct {
    const float* z; const float* e; const float* scale;   /* scale: 1 float on device */
    int style;                                            /* 0 multiply, 1 add */
    int B, H, W, C;
    float* out; int o_hp, o_wp, o_pad;
} nirgan_inject_fwd_desc;
int nirgan_inject_fwd(const nirgan_inject_fwd_desc* d, void* stream);

/* backward: g (dense, wrt a) -> dz (dense), de [B][H][W], dscale (1 float, accumulated) */
typedef struct {
    const float* g; const float* a; int a_hp, a_wp, a_pad;
    const float* z; const float* e; const float* scale;
    int style;
    int B, H, W, C;
    float* dz; float* de; float* dscale;
    float* ws; int64_t ws_elems;          /* >= 2048 floats when dscale is set: per-block partial sums of dscale (fixed-order finish) */
} nirgan_inject_bwd_desc;
int nirgan_inject_bwd(const nirgan_inject_bwd_desc* d, void* stream);

/* column sums: out[c] (+)= sum_r x[r][c]  (bias gradients of Linear) */
int nirgan_colsum(const float* x, int64_t rows, int cols, float* out, int accumulate, void* stream);

/* misc stream-ordered helpers */
/* ResnetGenerator_inject's optional post-correction (model/generator_inject.py:97-100,133-134: `x = x * self.post_correction_param`, a
 * learnable 0-dim parameter): out[i] = x[i] * (*param); the parameter is read from device memory (it changes with every optimizer step).
 * Backward: gx[i] = gout[i] * (*param) and *dparam += sum_i gout[i] * x[i] (x = the UNcorrected output; per-block partial sums in ws --
 * at least min(1024, ceil(n / 256)) floats -- added in block order: bitwise reproducible; the caller zeroes dparam first). */
int nirgan_param_scale_fwd(const float* x, const float* param, float* out, int64_t n, void* stream);
int nirgan_param_scale_bwd(const float* gout, const float* x, const float* param, float* gx, float* dparam, float* ws, int64_t ws_elems,
                           int64_t n, void* stream);
int nirgan_fill(float* dst, int64_t n, float value, void* stream);
int nirgan_axpy(float* y, const float* x, int64_t n, float alpha, void* stream);   /* y += alpha*x */

/* run a pre-built list of descriptors back to back (one host call per phase) */
#define NIRGAN_OP_CONV 1
#define NIRGAN_OP_WGRAD 2
#define NIRGAN_OP_IN_FWD 3
#define NIRGAN_OP_IN_BWD 4
typedef struct { int op; const void* desc; } nirgan_plan_entry;
int nirgan_run_plan(const nirgan_plan_entry* entries, int n, void* stream);

/* ---------------------------------------------------------------------------------------
 * Tiled inference on large scenes (create_synthetic_dataset.py:100-118 runs model(hr) per tile; model/pix2pix.py:91-93,107-108 hides
 * tile-edge artefacts with reflect-pad / crop): scene [B][C][H][W] -> n overlapping tiles [n][C][tile][tile] (tile (b, ti, tj), b-major,
 * covers rows ti*core - margin ..., core = tile - 2*margin, reflected at the scene's borders like F.pad(mode='reflect')), and the
 * tiles' cores back into a scene (pixels past H x W dropped).  One launch per batch of tiles [first, first + n).
 * ------------------------------------------------------------------------------------- */
int64_t nirgan_tile_count(int B, int H, int W, int tile, int margin);      /* B * ceil(H/core) * ceil(W/core) */
int nirgan_tile_gather(const float* scene, int B, int C, int H, int W, int tile, int margin, int first, int n, float* tiles, void* stream);
int nirgan_tile_scatter(const float* tiles, int B, int C, int H, int W, int tile, int margin, int first, int n, float* scene, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NIRGAN_HIP_H */
