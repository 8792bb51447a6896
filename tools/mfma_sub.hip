// mfma_sub.hip -- can the matrix pipe take the subtraction of the sDTW cell off the VALU, bit-exactly?
//
// Idea: v_mfma_f32_32x32x2_f32 computes D = A(32x2) * B(2x32) + C(32x32).  Lane l supplies ONE B element (row l/32,
// column l%32) and receives 16 D elements of column l%32 (rows 8*(v/4) + 4*(l/32) + v%4).  With A a 0/-1 selector
// matrix (row i picks k = (i>>2)&1) every lane gets  D[v] = -x_lane + C[v]  for its own x: sixteen `y - x` per lane
// per instruction, on a pipe that otherwise idles.
//
// Part 1: exactness of that against v_sub_f32 (normal, denormal, equal, huge operands).
// Part 2: throughput of {1 MFMA + N VALU} per loop iteration against {N+16 VALU}, DP-like op mix, 4 and 6 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float selector() {
    const int l = threadIdx.x & 63;
    const int i = l & 31, k = l >> 5;
    return (((i >> 2) & 1) == k) ? -1.0f : 0.0f;
}

__global__ void exact_kernel(const float *x, const float *y, float *d_mfma, float *d_valu) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const float a = selector();
    const float b = x[tid];
    f16v c;
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = y[tid * 16 + v];
    f16v d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        d_mfma[tid * 16 + v] = d[v];
        d_valu[tid * 16 + v] = c[v] - b;
    }
}

// MODE 0: VALU only, NV ops per iteration (cell mix: sub, min3, add|d|)
// MODE 1: 1 MFMA + NV VALU ops per iteration (min3, add|d| consuming the previous MFMA's results)
template <int MODE, int NV, int WAVES>
__global__ void __launch_bounds__(256, WAVES) thr_kernel(const float *in, float *out, int iters) {
    float c[16], xr[16];
    f16v yv, d0, d1;
    for (int r = 0; r < 16; ++r) {
        c[r] = in[threadIdx.x + r * 256];
        xr[r] = in[threadIdx.x + r * 256 + 4096];
        yv[r] = in[threadIdx.x + r * 256 + 8192];
        d0[r] = 0.f;
        d1[r] = 0.f;
    }
    const float a = selector();
    float up = in[threadIdx.x], diag = in[threadIdx.x + 1];
    auto cells = [&](const f16v &dd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // cell: m = min3(up, diag, left); cn = |d| + m   (+ d = x - y on the VALU in MODE 0)
            float d;
            if (MODE == 0) {
                asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d) : "v"(xr[r]), "v"(yv[r]));
            } else {
                d = dd[r];
            }
            float m;
            asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(up), "v"(diag), "v"(c[r]));
            diag = c[r];
            asm volatile("v_add_f32_e64 %0, |%1|, %2" : "=v"(c[r]) : "v"(d), "v"(m));
            up = c[r];
            if (NV > 32 && r == 0) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(diag) : "v"(up));
        }
    };
    for (int t = 0; t < iters; t += 2) {
        // one row's sixteen columns of y - x for the NEXT half-iteration, double-buffered
        if (MODE == 1) {
            __builtin_amdgcn_sched_barrier(0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[0], yv, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        cells(d0);
        if (MODE == 1) {
            __builtin_amdgcn_sched_barrier(0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[1], yv, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        cells(d1);
    }
    float s = up + diag;
    for (int r = 0; r < 16; ++r) s += c[r] + d0[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV, int WAVES>
void run(const char *name, const float *din, float *dout) {
    const int iters = 20000, blocks = 256 * WAVES * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((thr_kernel<MODE, NV, WAVES>), dim3(blocks), dim3(256), 0, 0, din, dout, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((thr_kernel<MODE, NV, WAVES>), dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-iterations per SIMD: blocks*4 waves over 1024 SIMDs
    const double wave_iters_per_simd = double(blocks) * 4 * iters / 1024.0;
    const double ns_per_iter = ms * 1e6 / wave_iters_per_simd;
    printf("%-44s waves/SIMD %d  %.2f ms  %.1f ns per 16-cell row-step per SIMD  (%.1f cycles @2.4GHz, %.1f @2.0GHz)\n", name, WAVES, ms,
           ns_per_iter, ns_per_iter * 2.4, ns_per_iter * 2.0);
}

int main() {
    // ---- part 1 ----
    const int n = 64 * 1024;
    std::vector<float> hx(n), hy(n * 16);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < n; ++i) {
        const unsigned cls = rnd() % 8;
        float v;
        unsigned bits = rnd();
        if (cls == 0) { bits &= 0x807fffffu; memcpy(&v, &bits, 4); }               // denormal
        else if (cls == 1) { v = float(int(rnd() % 2001) - 1000) / 256.0f; }         // quantised
        else if (cls == 2) { bits = (bits & 0x807fffffu) | (unsigned(1 + rnd() % 253) << 23); memcpy(&v, &bits, 4); }  // any exponent
        else { v = (float(rnd() % 2000001) - 1000000.0f) / 250000.0f; }              // z-score range
        hx[i] = v;
        for (int k = 0; k < 16; ++k) {
            const unsigned c2 = rnd() % 8;
            float w;
            unsigned b2 = rnd();
            if (c2 == 0) { b2 &= 0x807fffffu; memcpy(&w, &b2, 4); }
            else if (c2 == 1) { w = v; }
            else if (c2 == 2) { w = float(int(rnd() % 2001) - 1000) / 256.0f; }
            else if (c2 == 3) { b2 = (b2 & 0x807fffffu) | (unsigned(1 + rnd() % 253) << 23); memcpy(&w, &b2, 4); }
            else if (c2 == 4) { unsigned vb; memcpy(&vb, &v, 4); vb += int(rnd() % 5) - 2; memcpy(&w, &vb, 4); }  // neighbours of x
            else { w = (float(rnd() % 2000001) - 1000000.0f) / 250000.0f; }
            hy[i * 16 + k] = w;
        }
    }
    float *dx, *dy, *dm, *dv;
    (void)hipMalloc(&dx, n * 4);
    (void)hipMalloc(&dy, n * 64);
    (void)hipMalloc(&dm, n * 64);
    (void)hipMalloc(&dv, n * 64);
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dy, hy.data(), n * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dy, dm, dv);
    std::vector<float> hm(n * 16), hv(n * 16);
    (void)hipMemcpy(hm.data(), dm, n * 64, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hv.data(), dv, n * 64, hipMemcpyDeviceToHost);
    long bad = 0, bad_abs = 0, bad_host = 0, shown = 0;
    for (int i = 0; i < n * 16; ++i) {
        unsigned a, b;
        memcpy(&a, &hm[i], 4);
        memcpy(&b, &hv[i], 4);
        const float host = hy[i] - hx[i / 16];
        if (memcmp(&host, &hv[i], 4) != 0 && !(host == 0.0f && hv[i] == 0.0f)) ++bad_host;
        if (a != b) {
            ++bad;
            if ((a & 0x7fffffffu) != (b & 0x7fffffffu)) {
                ++bad_abs;
                if (shown++ < 8) printf("  mismatch x=%a y=%a mfma=%a valu=%a\n", hx[i / 16], hy[i], hm[i], hv[i]);
            }
        }
    }
    printf("exactness: %d values, %ld differ bitwise, %ld differ in |d| (what the cell uses); VALU vs host: %ld\n", n * 16, bad, bad_abs, bad_host);

    // ---- part 2 ----
    float *din, *dout;
    std::vector<float> h(16384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) % 1000) / 500.0f + 0.5f;
    (void)hipMalloc(&din, h.size() * 4);
    (void)hipMalloc(&dout, 256 * 8 * 2 * 256 * 4);
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 49, 6>("VALU only: 16x(sub,min3,add)+1", din, dout);
    run<0, 49, 4>("VALU only: 16x(sub,min3,add)+1", din, dout);
    run<1, 33, 6>("MFMA sub + 16x(min3,add)+1", din, dout);
    run<1, 33, 4>("MFMA sub + 16x(min3,add)+1", din, dout);
    run<1, 33, 3>("MFMA sub + 16x(min3,add)+1", din, dout);
    run<1, 33, 2>("MFMA sub + 16x(min3,add)+1", din, dout);
    return 0;
}
