// Shared host/device helpers for libnirgan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/nirgan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NG_GLOBAL __attribute__((address_space(1)))
#define NG_LDS __attribute__((address_space(3)))
#define NG_CONST __attribute__((address_space(4)))      // the kernarg segment: uniform reads are scalar loads, also with a runtime index

void nirgan_set_error(const char* fmt, ...);

#define NG_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            nirgan_set_error(__VA_ARGS__);         \
            return NIRGAN_ERR_ARG;                 \
        }                                          \
    } while (0)

static inline int nirgan_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        nirgan_set_error("%s: %s", what, hipGetErrorString(e));
        return NIRGAN_ERR_LAUNCH;
    }
    return NIRGAN_OK;
}

static inline bool ng_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// bijective XCD-aware remap of a 1-D block id: consecutive logical ids land on one XCD
// (blocks b and b+8 share an XCD under round-robin dispatch; speed only, never correctness).
__device__ __forceinline__ int ng_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// 16-byte async global -> LDS copy: LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void ng_glds16(const float* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const NG_GLOBAL void*)src, (NG_LDS void*)lds_wave_base, 16, 0, 0);
}

// The same with the address as SGPR base + 32-bit VGPR byte offset (global_load_lds_dwordx4 v_off, s[base:base+1]): no vector
// arithmetic per piece.  `base` must be wave-uniform (it is pinned to SGPRs here); the empty asm keeps the zero-extension of the
// offset inside the basic block of the load, where instruction selection can fold it (hoisted, it becomes a 64-bit VALU add).
__device__ __forceinline__ const char* ng_uniform_ptr(const char* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(a)), hi = __builtin_amdgcn_readfirstlane(unsigned(a >> 32));
    return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ void ng_glds16_so(const char* base, unsigned off, void* lds_wave_base) {
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((const NG_GLOBAL void*)(base + off), (NG_LDS void*)lds_wave_base, 16, 0, 0);
}

// reflect index into [0, n) (nn.ReflectionPad2d semantics, pad < n)
__host__ __device__ __forceinline__ int ng_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ float ng_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Fixed-order sum of per-block partial sums (no float atomics between blocks: results are bitwise reproducible).  ws[rows][nv], nv <= 8:
// dst[i] += sum over rows of ws[row][i], one block, every launch the same association.  The final add is ONE atomic add per value, so
// that two streams accumulating into the same scalar (micro-batches) commute.  Defined in elementwise.hip.
int ng_partials_finish(const float* ws, int rows, int nv, float* dst, hipStream_t st);

