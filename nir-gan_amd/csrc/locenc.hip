// SatCLIP location encoder: lon/lat -> real spherical harmonics (L*L features, closed form) ->
// SirenNet -> embedding, all in fp64 like the reference (model/satclip/load_lightweight.py:29 `.double()`,
// satclip_wrapper.py:33).  B is the tile batch (8-32 coordinates): a latency kernel, one workgroup per
// coordinate (16 waves), activations in LDS, two output rows of a linear layer per wave and pass, wave-shuffle reductions.
//
// Harmonics follow positional_encoding/spherical_harmonics.py:26-42 and spherical_harmonics_closed_form.py:8-40:
//   phi = deg2rad(lon + 180), theta = deg2rad(lat + 90), feature (l, m), m = -l..l, l = 0..L-1 at index l*l + l + m:
//   m = 0: K * P_l^0(cos theta);  m > 0: K * cos(m phi) * P_l^m;  m < 0: K * sin(-m phi) * P_l^-m
//   with K = [sqrt 2] * sqrt((2l+1)(l-|m|)! / (4 pi (l+|m|)!)) precomputed on the host (exact integer factorials),
//   and the associated Legendre recurrence evaluated in the reference's operation order.
#include "common.h"

namespace {

constexpr int MAX_LAYERS = 8;
constexpr int MAX_WIDTH = 2048;

struct LocEncP {
    const double* lonlat;
    const double* K;            // [L*L]
    const double* w[MAX_LAYERS];
    const double* b[MAX_LAYERS];
    int din[MAX_LAYERS], dout[MAX_LAYERS];
    double w0[MAX_LAYERS];      // sine frequency of the layer; 0 = identity (last layer)
    int nlayers, L;
    double* out;
    double* feat;               // optional [B][L*L] copy of the harmonics (tests), or nullptr
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ double assoc_legendre(int l, int m, double x) {
    double pmm = 1.0;
    if (m > 0) {
        const double somx2 = sqrt((1.0 - x) * (1.0 + x));
        double fact = 1.0;
        for (int i = 1; i <= m; ++i) {
            pmm = pmm * (-fact) * somx2;
            fact += 2.0;
        }
    }
    if (l == m) return pmm;
    double pmmp1 = x * (2.0 * m + 1.0) * pmm;
    if (l == m + 1) return pmmp1;
    double pll = 0.0;
    for (int ll = m + 2; ll <= l; ++ll) {
        pll = ((2.0 * ll - 1.0) * x * pmmp1 - (ll + m - 1.0) * pmm) / (ll - m);
        pmm = pmmp1;
        pmmp1 = pll;
    }
    return pll;
}

__global__ __launch_bounds__(1024) void locenc_kernel(const LocEncP p) {
    __shared__ double buf[2][MAX_WIDTH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.x;
    const double lon = p.lonlat[2 * bi], lat = p.lonlat[2 * bi + 1];
    const double d2r = 3.14159265358979323846 / 180.0;          // torch.deg2rad: x * (pi / 180)
    const double phi = (lon + 180.0) * d2r, theta = (lat + 90.0) * d2r;
    const double ct = cos(theta);
    const int nf = p.L * p.L;
    for (int f = tid; f < nf; f += 1024) {
        int l = int(sqrt(double(f)));
        while (l * l > f) --l;
        while ((l + 1) * (l + 1) <= f) ++l;
        const int m = f - l * l - l;
        const int am = m < 0 ? -m : m;
        const double P = assoc_legendre(l, am, ct);
        double y;
        if (m == 0) y = p.K[f] * P;
        else if (m > 0) y = p.K[f] * cos(double(m) * phi) * P;
        else y = p.K[f] * sin(double(am) * phi) * P;
        buf[0][f] = y;
        if (p.feat) p.feat[size_t(bi) * nf + f] = y;
    }
    __syncthreads();
    int cur = 0;
    for (int li = 0; li < p.nlayers; ++li) {
        const int din = p.din[li], dout = p.dout[li];
        const double* W = p.w[li];
        const bool last = li == p.nlayers - 1;
        // 16 waves, two output rows per wave and pass: the weight rows stream from L2, so the loop is latency-bound and
        // wants many independent loads in flight (4 waves x 1 row measured 412 us for 32 coordinates)
        for (int j = wave * 2; j < dout; j += 32) {
            const bool two = j + 1 < dout;
            const double* row0 = W + size_t(j) * din;
            const double* row1 = W + size_t(two ? j + 1 : j) * din;
            double s0 = 0.0, s1 = 0.0;
            for (int k = lane; k < din; k += 64) {
                const double xv = buf[cur][k];
                s0 += row0[k] * xv;
                s1 += row1[k] * xv;
            }
            s0 = wave_sum_f64(s0);
            s1 = wave_sum_f64(s1);
            if (lane < 2 && (lane == 0 || two)) {
                const int jj = j + lane;
                double s = lane == 0 ? s0 : s1;
                if (p.b[li]) s += p.b[li][jj];
                if (p.w0[li] != 0.0) s = sin(p.w0[li] * s);
                if (last) p.out[size_t(bi) * dout + jj] = s;
                else buf[cur ^ 1][jj] = s;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace

extern "C" int nirgan_location_encoder(const nirgan_locenc_desc* d, void* stream) {
    NG_REQUIRE(d != nullptr && d->lonlat && d->sh_norm && d->weights && d->biases && d->dims && d->w0 && d->out, "location_encoder: null pointer");
    NG_REQUIRE(d->B > 0, "location_encoder: empty batch");
    NG_REQUIRE(d->L >= 1 && d->L * d->L <= MAX_WIDTH, "location_encoder: legendre_polys=%d out of range (L*L <= %d)", d->L, MAX_WIDTH);
    NG_REQUIRE(d->nlayers >= 1 && d->nlayers <= MAX_LAYERS, "location_encoder: %d linear layers (1..%d)", d->nlayers, MAX_LAYERS);
    NG_REQUIRE(d->dims[0] == d->L * d->L, "location_encoder: first layer expects %d inputs, the harmonics give %d", d->dims[0], d->L * d->L);
    LocEncP p;
    p.lonlat = d->lonlat; p.K = d->sh_norm; p.L = d->L; p.nlayers = d->nlayers; p.out = d->out; p.feat = d->features;
    for (int i = 0; i < MAX_LAYERS; ++i) { p.w[i] = nullptr; p.b[i] = nullptr; p.din[i] = p.dout[i] = 0; p.w0[i] = 0.0; }
    for (int i = 0; i < d->nlayers; ++i) {
        NG_REQUIRE(d->weights[i] != nullptr, "location_encoder: layer %d has no weight", i);
        NG_REQUIRE(d->dims[i] >= 1 && d->dims[i] <= MAX_WIDTH && d->dims[i + 1] >= 1 && d->dims[i + 1] <= MAX_WIDTH,
                   "location_encoder: layer %d is %d -> %d (width <= %d)", i, d->dims[i], d->dims[i + 1], MAX_WIDTH);
        p.w[i] = d->weights[i]; p.b[i] = d->biases[i]; p.din[i] = d->dims[i]; p.dout[i] = d->dims[i + 1]; p.w0[i] = d->w0[i];
    }
    hipLaunchKernelGGL(locenc_kernel, dim3(d->B), dim3(1024), 0, static_cast<hipStream_t>(stream), p);
    return nirgan_check_launch("location_encoder");
}
