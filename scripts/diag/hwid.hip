// Diagnostic: which hardware wave slots do the waves of co-resident 256-thread blocks (64 KB LDS each) get?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, unsigned long long* t) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = 1;
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // keep the block resident for a while so that two blocks overlap on a CU
    for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(10);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
        t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t0;
    }
}
int main() {
    const int nb = 1024;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, nb * 4 * 2 * 4); hipMalloc(&dt, nb * 4 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 65536, 0, d, dt);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8); std::vector<unsigned long long> ht(nb * 4);
    hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, nb * 32, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;   // key: xcc, se, sh, cu, simd -> list of (block, slot)
    for (int b = 0; b < 40; ++b) {
        printf("block %4d:", b);
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            printf("  [xcc %u se %u sh %u cu %2u simd %u slot %2u]", x & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    // parity statistics of slot ids among waves sharing a SIMD in the first 512 blocks
    std::map<unsigned long long, std::vector<unsigned>> simd;
    for (int b = 0; b < 512; ++b) for (int w = 0; w < 4; ++w) {
        unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
        unsigned long long key = ((unsigned long long)(x & 15) << 32) | (hw & 0xFFF0);
        simd[key].push_back(hw & 15);
    }
    int same = 0, diff = 0, n1 = 0, n2 = 0, nmore = 0;
    for (auto& kv : simd) {
        if (kv.second.size() == 1) n1++; else if (kv.second.size() == 2) { n2++; ((kv.second[0] ^ kv.second[1]) & 1) ? diff++ : same++; } else nmore++;
    }
    printf("SIMDs hosting 1/2/more waves of first 512 blocks: %d %d %d ; pairs with different slot parity %d, same %d\n", n1, n2, nmore, diff, same);
    return 0;
}
