// valu_mix.hip -- how do half-rate (min/max family: 4.3 cycles per wave-instruction alone) and full-rate (add/sub/mov:
// ~2.3 cycles) VALU opcodes share a gfx950 SIMD?  K full-rate instructions per v_min3_f32, independent chains, several
// interleavings and occupancies.  serial model: 4.3 + 2.3*K cycles per group; concurrent model: max(4.3, 2.3*K).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// PATTERN 0: m a a .. (per chain group, interleaved finely)   PATTERN 1: all 16 min3 first, then all 16*K adds
template <int K, int PATTERN, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k(const float *in, float *out, int iters) {
    float a[16], b[16], c[16], d[16];
    for (int r = 0; r < 16; ++r) {
        a[r] = in[threadIdx.x + r * 256];
        b[r] = in[threadIdx.x + r * 256 + 4096];
        c[r] = in[threadIdx.x + r * 256 + 8192];
        d[r] = in[threadIdx.x + r * 256 + 1024];
    }
    for (int t = 0; t < iters; ++t) {
        if (PATTERN == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
#pragma unroll
                for (int j = 0; j < K; ++j) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(d[(r + j) & 15]) : "v"(b[r]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < K; ++j) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(d[(r + j) & 15]) : "v"(b[r]));
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a[r] + d[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// cell mixes: MODE 0 = 16x(sub, min3, add|d|)   MODE 1 = 8x v_pk_add_f32 (two subtractions each) + 16x(min3, add|d|)
// MODE 2 = pk_add alone
template <int MODE>
__global__ void __launch_bounds__(256, 6) kc(const float *in, float *out, int iters) {
    float a[16], b[16], c[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x2[8], d2[8], y2;
    for (int r = 0; r < 16; ++r) {
        a[r] = in[threadIdx.x + r * 256];
        b[r] = in[threadIdx.x + r * 256 + 4096];
        c[r] = in[threadIdx.x + r * 256 + 8192];
    }
    for (int r = 0; r < 8; ++r) {
        x2[r][0] = in[threadIdx.x + r * 256 + 1024];
        x2[r][1] = in[threadIdx.x + r * 256 + 1025];
        d2[r] = x2[r];
    }
    y2[0] = in[threadIdx.x + 77];
    y2[1] = y2[0];
    for (int t = 0; t < iters; ++t) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("v_sub_f32_e32 %1, %2, %1\n\tv_min3_f32 %0, %0, %2, %1\n\tv_add_f32_e64 %0, |%1|, %0" : "+v"(a[r]), "+v"(c[r]) : "v"(b[r]));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d2[r]) : "v"(x2[r]), "v"(y2));
                asm volatile("v_min3_f32 %0, %0, %1, %2\n\tv_add_f32_e64 %0, |%3|, %0" : "+v"(a[2 * r]) : "v"(b[2 * r]), "v"(c[2 * r]), "v"(d2[r][0]));
                asm volatile("v_min3_f32 %0, %0, %1, %2\n\tv_add_f32_e64 %0, |%3|, %0" : "+v"(a[2 * r + 1]) : "v"(b[2 * r + 1]), "v"(c[2 * r + 1]), "v"(d2[r][1]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d2[r]) : "v"(y2));
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a[r] + c[r];
    for (int r = 0; r < 8; ++r) s += d2[r][0] + d2[r][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void runc(const char *name, int instr_per_iter, const float *din, float *dout) {
    const int iters = 20000, blocks = 256 * 6 * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kc<MODE>), dim3(blocks), dim3(256), 0, 0, din, dout, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kc<MODE>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double it_per_simd = double(blocks) * 4 * iters / 1024.0;
    const double ns = ms * 1e6 / it_per_simd;
    printf("%-52s %.2f ms  %.1f ns per iteration per SIMD (%d instr: %.2f ns each)\n", name, ms, ns, instr_per_iter, ns / instr_per_iter);
}

template <int K, int PATTERN, int WAVES>
void run(const float *din, float *dout) {
    const int iters = 20000, blocks = 256 * WAVES * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K, PATTERN, WAVES>), dim3(blocks), dim3(256), 0, 0, din, dout, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, PATTERN, WAVES>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_simd = double(blocks) * 4 * iters * 16 / 1024.0;
    const double ns = ms * 1e6 / groups_per_simd;
    printf("K=%d adds per min3, pattern %d, %d waves/SIMD: %.2f ms, %.2f ns per group per SIMD = %.2f cycles @2.1GHz  (serial %.1f, concurrent %.1f)\n",
           K, PATTERN, WAVES, ms, ns, ns * 2.1, 4.3 + 2.3 * K, (2.3 * K > 4.3 ? 2.3 * K : 4.3));
}

int main() {
    float *din, *dout;
    std::vector<float> h(16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) % 1000) / 500.0f + 0.5f;
    (void)hipMalloc(&din, h.size() * 4);
    (void)hipMalloc(&dout, 256 * 8 * 2 * 256 * 4);
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    runc<0>("16 cells: 16x(sub,min3,add)", 48, din, dout);
    runc<1>("16 cells: 8x pk_add + 16x(min3,add)", 40, din, dout);
    runc<2>("8x v_pk_add_f32 alone", 8, din, dout);
    run<0, 0, 8>(din, dout);
    run<1, 0, 8>(din, dout);
    run<2, 0, 8>(din, dout);
    run<3, 0, 8>(din, dout);
    run<4, 0, 8>(din, dout);
    run<6, 0, 8>(din, dout);
    run<2, 1, 8>(din, dout);
    run<4, 1, 8>(din, dout);
    run<2, 0, 4>(din, dout);
    run<2, 0, 2>(din, dout);
    run<2, 0, 1>(din, dout);
    run<2, 1, 4>(din, dout);
    run<2, 1, 2>(din, dout);
    return 0;
}
