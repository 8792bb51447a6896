// valu_opcost.hip -- issue cost of individual VALU opcodes on gfx950 (wave64 on SIMD-32), 8 waves/SIMD,
// 16 independent chains per lane so that dependencies never stall.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256, 8) k(const float *in, float *out, int iters) {
    float a[16], b[16], c[16];
    for (int r = 0; r < 16; ++r) {
        a[r] = in[threadIdx.x + r * 256];
        b[r] = in[threadIdx.x + r * 256 + 4096];
        c[r] = in[threadIdx.x + r * 256 + 8192];
    }
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
            if (MODE == 2) asm volatile("v_add_f32_e64 %0, |%1|, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
            if (MODE == 4) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 5) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
            if (MODE == 6) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 7) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[r]) : "v"(b[r]));  // 2 distinct VGPRs
            if (MODE == 8) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 9) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 10) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
            if (MODE == 11) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b[r]) : "vcc");
            if (MODE == 12) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b[r]) : "vcc");
            if (MODE == 13) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 14) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 15) asm volatile("v_sub_f32_e32 %1, %2, %1\n\tv_min3_f32 %0, %0, %2, %1\n\tv_add_f32_e64 %0, |%1|, %0" : "+v"(a[r]), "+v"(c[r]) : "v"(b[r]));
            if (MODE == 16) asm volatile("v_min_i32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 17) asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 18) asm volatile("v_min_f64 %0, %0, %1" : "+v"(*(double*)&a[r & ~1]) : "v"(*(double*)&b[r & ~1]));
            if (MODE == 19) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[r]) : "v"(b[r]));
            if (MODE == 21) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));  // |b - c| + a, 32-bit unsigned
            // a fixed-point cell in TWO instructions: m = min3(up, diag, left); cell = |x - y| + m
            if (MODE == 22) asm volatile("v_min3_u32 %1, %0, %2, %1\n\tv_sad_u32 %0, %2, %1, %1" : "+v"(a[r]), "+v"(c[r]) : "v"(b[r]));
            if (MODE == 23) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));  // gfx950: IEEE-754-2019 minimum
            if (MODE == 24) asm volatile("v_sub_f32_e32 %1, %2, %1\n\tv_minimum3_f32 %0, %0, %2, %1\n\tv_add_f32_e64 %0, |%1|, %0" : "+v"(a[r]), "+v"(c[r]) : "v"(b[r]));
            if (MODE == 25) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b[r]), "v"(c[r]));
            if (MODE == 20) asm volatile("v_sub_f32_e32 %1, %2, %1\n\tv_add_f32_e64 %0, |%1|, %0\n\tv_add_f32_e64 %0, |%1|, %0" : "+v"(a[r]), "+v"(c[r]) : "v"(b[r]));
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, const float *din, float *dout) {
    const int iters = 40000, blocks = 256 * 8 * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, din, dout, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = double(blocks) * 4 * iters * 16 * ((MODE == 11 || MODE == 22) ? 2 : (MODE == 15 || MODE == 20 || MODE == 24) ? 3 : 1);
    printf("%-34s %.2f ms  %.3e wave-instr/s  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, instr / (ms * 1e-3), 2.4e9 * 1024 / (instr / (ms * 1e-3)));
}
int main() {
    float *din, *dout;
    std::vector<float> h(16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) % 1000) / 500.0f + 0.5f;
    (void)hipMalloc(&din, h.size() * 4);
    (void)hipMalloc(&dout, 256 * 8 * 2 * 256 * 4);
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("v_add_f32_e32 (VOP2)", din, dout);
    run<6>("v_sub_f32_e32 (VOP2)", din, dout);
    run<4>("v_min_f32_e32 (VOP2)", din, dout);
    run<2>("v_add_f32_e64 |abs| (VOP3, 2 src)", din, dout);
    run<1>("v_min3_f32 (VOP3, 3 src)", din, dout);
    run<7>("v_min3_f32 (VOP3, 2 distinct src)", din, dout);
    run<5>("v_min3_u32 (VOP3, 3 src)", din, dout);
    run<3>("v_fma_f32 (VOP3, 3 src)", din, dout);
    run<8>("v_max_f32_e32", din, dout);
    run<9>("v_min_u32_e32", din, dout);
    run<16>("v_min_i32_e32", din, dout);
    run<10>("v_med3_f32", din, dout);
    run<11>("v_cmp_lt + v_cndmask (2 instr)", din, dout);
    run<12>("v_cndmask_b32", din, dout);
    run<13>("v_mov_b32", din, dout);
    run<14>("v_and_b32", din, dout);
    run<19>("v_add_u32", din, dout);
    run<17>("v_pk_min_f16", din, dout);
    run<15>("sub+min3+add (3 instr, cell mix)", din, dout);
    run<20>("sub+add+add (3 instr, no min)", din, dout);
    run<21>("v_sad_u32 (|a-b|+c, VOP3)", din, dout);
    run<22>("min3_u32 + sad_u32 (2 instr, fixed-point cell)", din, dout);
    run<23>("v_minimum3_f32 (gfx950)", din, dout);
    run<24>("sub+minimum3+add (3 instr)", din, dout);
    run<25>("v_pk_minimum3_f16 (gfx950)", din, dout);
    run<15>("sub+min3+add (3 instr, cell mix) again", din, dout);
    return 0;
}
