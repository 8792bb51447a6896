// hwid_probe.hip -- where do the waves of a grid land?  Reads HW_REG_HW_ID / HW_REG_XCC_ID in every wave of a grid of
// 256-thread blocks and prints how many distinct (xcc, se, sh, cu, simd) positions were used and how many waves each got.
// Background for the persistent fill kernel's per-SIMD balancing (sdtw_persist.hpp): it keys its "tasks running on this
// SIMD" counters by this index.   build: hipcc --offload-arch=gfx950 -O3 -o tools/hwid_probe tools/hwid_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__device__ __forceinline__ unsigned simd_index() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, bits 3:0
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return ((((xcc & 7) * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd;
}

__global__ void __launch_bounds__(256, 6) probe(unsigned *out, unsigned *raw, int spin) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    // hold the slot for a while so that the whole grid is resident together (80 VGPRs like the fill: occupancy 6)
    float v[64];
    for (int i = 0; i < 64; ++i) v[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < spin; ++it)
        for (int i = 0; i < 64; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) & 63];
    float s = 0;
    for (int i = 0; i < 64; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) {
        out[wave] = simd_index() | (s == 12345.f ? 0x80000000u : 0u);
        raw[wave] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    }
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1536;
    const int spin = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *d, *r;
    hipMalloc(&d, 4 * blocks * 4);
    hipMalloc(&r, 4 * blocks * 4);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, r, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 4), hr(blocks * 4);
    hipMemcpy(h.data(), d, 4 * blocks * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), r, 4 * blocks * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    int same_block_same_simd = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 4; ++w) cnt[h[b * 4 + w] & 0x7fffffff]++;
        for (int w = 1; w < 4; ++w)
            if ((h[b * 4 + w] & 3) == (h[b * 4] & 3)) same_block_same_simd++;
    }
    std::map<int, int> hist;
    for (auto &kv : cnt) hist[kv.second]++;
    printf("blocks %d: %zu distinct SIMD positions;", blocks, cnt.size());
    for (auto &kv : hist) printf(" %d SIMDs x %d waves,", kv.second, kv.first);
    printf(" blocks with two waves on one SIMD: %d\n", same_block_same_simd);
    std::map<unsigned, int> xcc, se, cu;
    for (auto &kv : cnt) {
        xcc[kv.first >> 9]++;
        se[(kv.first >> 7) & 3]++;
    }
    printf("SIMDs per xcc:");
    for (auto &kv : xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\nfirst raw HW_ID words: %08x %08x %08x %08x %08x\n", hr[0], hr[1], hr[2], hr[3], hr[4]);
    return 0;
}
